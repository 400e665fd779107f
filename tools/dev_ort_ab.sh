#!/bin/bash
# dev: same-box A/B of two library builds in precision 2: ORT tests once, then per-kernel averages (one batch at a time) at
# B = 64 / 256 and the bench lines.   usage: tools/dev_ort_ab.sh <kernel-regex> <libA.so> <libB.so>   (paths under the repo)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PAT=$1; shift
cd "$R"; timeout 600 python -m pytest tests/test_gpu_ort_mixed.py -q -x 2>&1 | tail -n 1
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename "$lib" .so)
  for bt in 64 256; do
    st=12; [ $bt = 256 ] && st=5
    QVERSE_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/ortab/$n$bt" -o p -- python "$R/bench.py" --precision ort --batch $bt --steps $st --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
    f=$(find "$R/gpurun_out/ortab/$n$bt" -name "*kernel_stats.csv" | head -1)
    python - "$f" "$n B=$bt" "$PAT" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[3], r["Name"]):
        print(sys.argv[2], r["Name"][:60], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
    QVERSE_LIB=$R/$lib timeout 300 python "$R/bench.py" --precision ort --batch $bt --steps $((st * 3)) --no-cpu-baseline --no-post-logits --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   $n B=$bt bench', d['value'], d['ms_per_step'])"
  done
done
find "$R/gpurun_out/ortab" -name "*_kernel_trace.csv" -delete
