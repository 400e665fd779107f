#!/usr/bin/env python
"""Fold rocprofv3 --pmc counter CSVs of tools/gemm_bench into tracked summaries under profiles/.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [rows]
        HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: both counters in
        KiB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced read).  `rows` = the M the
        shapes were run at (tools/gemm_bench's second argument, default 8064); bench.py quotes the figure
        only for a run with the same M.
    python tools/pmc_traffic.py --mfma <sq_counter_collection.csv> <out.json> [rows]
        MFMA utilisation inside each GEMM kernel from one SQ pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
        SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY, SQ_INSTS_VALU_MFMA_MOPS_F16 when present).
"""
import csv
import json
import re
import sys
from collections import defaultdict

EPI = ["f16", "f16_swish", "f16_relu", "glu", "resid", "f32", "qkv"]


def short_name(kernel: str):
    """'void k_gemm<1, 128, 0, 2, 1>(GemmArgs)' -> 'k_gemm<f16_swish,128>' (+ loader / stage variant);
    'void k_gemm256<1, 0>(GemmArgs)' -> 'k_gemm256<f16_swish>'."""
    w = re.search(r"k_gemm256<(\d+)(?:, (\d+))?(?:, (\d+))?>", kernel)
    if w:
        wq, mi = int(w.group(2) or 0), int(w.group(3) or 4)
        return (f"k_gemm256<{EPI[int(w.group(1))]}{',192' if mi == 3 else ''}>",
                f"{64 * mi} x 256 tiles, one block per CU, every wave stages and computes" + (f", int{wq} weights" if wq else ""))
    m = re.search(r"k_gemm(?:_pk)?<(\d+), (\d+), (\w+), (\d+)(?:, (\d+))?>", kernel)
    if not m:
        return None, None
    epi, bn, wq, nst, ld = int(m.group(1)), int(m.group(2)), m.group(3), int(m.group(4)), int(m.group(5) or 0)
    wq = {"false": 0, "true": 4}.get(wq, None) if not wq.isdigit() else int(wq)
    name = f"k_gemm<{EPI[epi]},{bn}>"
    variant = ("mubuf-register-staged loaders" if ld == 1 else f"direct-to-LDS loaders, {nst} stages") + (f", int{wq} weights" if wq else "")
    return name, variant


LAUNCHES = {}   # kernel name -> launches seen in the last CSV read (tools/gemm_bench also runs every kernel once, cold,
                # for its bit-for-bit comparison of the two tile shapes: a count of 1-2 marks such a row)


def mean_by_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        LAUNCHES[k] = len(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def traffic(argv):
    fetch = mean_by_kernel(argv[0], "FETCH_SIZE")
    write = mean_by_kernel(argv[1], "WRITE_SIZE")
    rows = int(argv[3]) if len(argv) > 3 else 8064
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gemm_bench (model GEMM "
                     f"shapes at M = {rows}, same kernels and launch geometry as the engine); bytes = counter * 1024, "
                     "FETCH_SIZE x 2 (gfx950 correction, calibrated on k_layernorm).  A kernel name shared by two "
                     "shapes (FFN-down and the out-projection) reports their mean.",
           "rows": rows, "kernels": {}}
    for k, f in fetch.items():
        short, variant = short_name(k)
        if short is None:
            continue
        w = write.get(k, 0.0)
        out["kernels"][short] = {"hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kib_raw": round(f, 1),
                                 "write_size_kib": round(w, 1), "variant": variant, "rocprof_name": k, "launches": LAUNCHES.get(k)}
    json.dump(out, open(argv[2], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


def mfma(argv):
    path, dst = argv[0], argv[1]
    rows = int(argv[2]) if len(argv) > 2 else 8064
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
             "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU_MFMA_MOPS_F16", "GRBM_GUI_ACTIVE"]
    per = {n: mean_by_kernel(path, n) for n in names}
    out = {"method": "rocprofv3 --pmc (one SQ pass, no tracing besides --kernel-trace) on tools/gemm_bench at "
                     f"M = {rows}; counters are chip sums per launch, averaged over launches.  "
                     "SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles per SIMD (= 32 x the number of "
                     "v_mfma_f32_32x32x16_f16 issued); SQ_BUSY_CYCLES is summed over the 32 shader engines, so the "
                     "kernel lasted SQ_BUSY_CYCLES / 32 shader cycles and the chip's 1024 SIMDs offered "
                     "SQ_BUSY_CYCLES * 32 matrix-pipe cycles: mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * 32) "
                     "(clock-independent; the TFLOP/s-over-2.5-PF figure of bench.py is the same quantity times actual "
                     "clock / 2.4 GHz).  The wave-cycle split (parked at s_waitcnt / s_barrier, issue-stalled, issuing) "
                     "is the one MI355X_MICROARCH.md describes (quad-cycles).",
           "rows": rows, "kernels": {}}
    kernels = set()
    for d in per.values():
        kernels |= set(d)
    for k in sorted(kernels):
        short, variant = short_name(k)
        if short is None:
            continue
        e = {n: per[n].get(k) for n in names if per[n].get(k) is not None}
        row = {"variant": variant, "rocprof_name": k, "launches": LAUNCHES.get(k), **{n: round(v, 1) for n, v in e.items()}}
        if e.get("SQ_WAVE_CYCLES"):
            wc = e["SQ_WAVE_CYCLES"]
            for n, key in (("SQ_WAIT_ANY", "frac_wave_cycles_parked"), ("SQ_WAIT_INST_ANY", "frac_wave_cycles_issue_stalled"),
                           ("SQ_ACTIVE_INST_ANY", "frac_wave_cycles_issuing")):
                if n in e:
                    row[key] = round(e[n] / wc, 4)
        if e.get("SQ_VALU_MFMA_BUSY_CYCLES") and e.get("SQ_BUSY_CYCLES"):
            row["kernel_shader_cycles"] = round(e["SQ_BUSY_CYCLES"] / 32.0, 1)
            row["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["SQ_BUSY_CYCLES"] * 32.0), 4)
        out["kernels"].setdefault(short, row)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--mfma":
        mfma(sys.argv[2:])
    else:
        traffic(sys.argv[1:])
