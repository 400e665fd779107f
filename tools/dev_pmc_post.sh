#!/bin/bash
# dev: SQ counters of selected kernels in a short one-batch-at-a-time bench run (separate --pmc passes, kernel trace only)
# usage: tools/dev_pmc_post.sh <kernel-name-regex>   (the post-logits replay of tools/post_bench.py instead of the bench)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; PREC=${2:-fp16}
O=$R/gpurun_out/pmcp
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, counters...
  tag=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$O/$tag" -o p -- python "$R/tools/post_bench.py" --steps 3 > /dev/null 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
python - "$O" "$PAT" <<'PY'
import csv, glob, re, sys
from collections import defaultdict
O, pat = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for tag in ("a", "b"):
    for f in glob.glob(f"{O}/{tag}/**/*counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            if not re.search(pat, r["Kernel_Name"]): continue
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r["Kernel_Name"], r["Dispatch_Id"])
            if tag == "a" and key not in seen: seen.add(key); cnt[r["Kernel_Name"][:48]] += 1
for k, v in acc.items():
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c in sorted(v): print(f"   {c:26s} {v[c] / n:16.0f}")
    w = v.get("SQ_WAVE_CYCLES", 0)
    if w:
        print("   wave cycles: parked %.2f  issue-stalled %.2f  issuing %.2f;  VALU-active share of wave cycles %.2f" % (
            v["SQ_WAIT_ANY"] / w, v["SQ_WAIT_INST_ANY"] / w, v["SQ_ACTIVE_INST_ANY"] / w, v["SQ_ACTIVE_INST_VALU"] / w))
PY
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -size +8M -delete
