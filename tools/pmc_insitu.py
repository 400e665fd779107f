#!/usr/bin/env python
"""Matrix-pipe utilisation of the GEMM family IN SITU: fold the counter CSV of

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o p -- \
        python bench.py --contexts 1 [--batch 256] --steps 3 --warmup 1 --no-cpu-baseline --no-post-logits --no-extra

(the bench command itself, one batch at a time: counter collection serialises dispatches anyway) into one number per GEMM
kernel and one for ALL GEMM launches together, weighted by what each launch offered:

    mfma_util = sum SQ_VALU_MFMA_BUSY_CYCLES / (32 * sum SQ_BUSY_CYCLES)      (tools/pmc_traffic.py explains the 32)

    python tools/pmc_insitu.py <counter_collection.csv> <out.json> <rows> <precision>
bench.py quotes `all_gemm.mfma_util` as roofline.mfma_util_in_situ for a run with the same row count and weights.
"""
import csv
import json
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import short_name  # noqa: E402


def main():
    path, dst, rows, prec = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for r in csv.DictReader(open(path)):
        name, _ = short_name(r["Kernel_Name"])
        if name is None:
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[name].add(r["Dispatch_Id"])
    out = {"method": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace over bench.py --contexts 1 (the "
                     "forward's own launches, warm-up and roofline replays included); per kernel and over all GEMM launches: "
                     "mfma_util = sum MFMA_BUSY / (32 * sum SQ_BUSY)", "rows": rows, "weights": prec, "kernels": {}}
    tot_m = tot_b = 0.0
    for k, v in sorted(acc.items()):
        m, b = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), v.get("SQ_BUSY_CYCLES", 0.0)
        if not b:
            continue
        out["kernels"][k] = {"launches": len(launches[k]), "mfma_util": round(m / (32.0 * b), 4),
                             "share_of_gemm_busy_cycles": 0.0, "SQ_BUSY_CYCLES_sum": round(b, 1)}
        tot_m += m
        tot_b += b
    for k in out["kernels"]:
        out["kernels"][k]["share_of_gemm_busy_cycles"] = round(out["kernels"][k]["SQ_BUSY_CYCLES_sum"] / tot_b, 4)
    out["all_gemm"] = {"mfma_util": round(tot_m / (32.0 * tot_b), 4) if tot_b else None, "launches": sum(len(v) for v in launches.values())}
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["all_gemm"]), json.dumps({k: v["mfma_util"] for k, v in out["kernels"].items()}))


if __name__ == "__main__":
    main()
