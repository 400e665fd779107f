#!/usr/bin/env python
"""Fold two rocprofv3 --pmc counter CSVs (FETCH_SIZE pass, WRITE_SIZE pass) of tools/gemm_bench into
profiles/<round>_pmc_traffic.json: HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024
(MI355X_MICROARCH.md: both counters in KiB; on gfx950 FETCH_SIZE reports half the bytes).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv
import json
import sys
from collections import defaultdict

NAMES = {"k_gemm<1, 128, 0, 2>": "k_gemm<f16_swish,128>", "k_gemm<4, 128, 0, 4>": "k_gemm<resid,128,4-stage>",
         "k_gemm<4, 128, 0, 3>": "k_gemm<resid,128,3-stage>", "k_gemm<6, 128, 0, 2>": "k_gemm<qkv,128>",
         "k_gemm<3, 128, 0, 2>": "k_gemm<glu,128>"}
NAMES.update({k.replace(", 0, ", ", false, "): v for k, v in list(NAMES.items())})   # binaries built before the WQ parameter


def mean_by_kernel(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = mean_by_kernel(sys.argv[1], "FETCH_SIZE")
    write = mean_by_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on tools/gemm_bench (model GEMM "
                     "shapes at M = 8064, same kernels and launch geometry as the engine); bytes = counter * 1024, "
                     "FETCH_SIZE x 2 (gfx950 correction, calibrated on k_layernorm)", "kernels": {}}
    for k, f in fetch.items():
        short = next((v for n, v in NAMES.items() if n in k), None)
        if short is None:
            continue
        w = write.get(k, 0.0)
        out["kernels"][short] = {"hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kib_raw": round(f, 1),
                                 "write_size_kib": round(w, 1), "rocprof_name": k}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()
