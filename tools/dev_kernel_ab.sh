#!/bin/bash
# dev: bitwise test of the forward + rocprofv3 per-kernel averages (one batch at a time) at B = 64 and B = 256.
# usage: tools/dev_kernel_ab.sh <tag> <kernel-name-regex> [precision]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}; PAT=${2:-.}; PREC=${3:-fp16}
O=$R/gpurun_out/ab_$TAG
mkdir -p "$O"
cd "$R"
timeout 600 python -m pytest tests/test_gpu_forward.py -q -x 2>&1 | tail -n 2
cd /tmp && export TMPDIR=/tmp
for bt in 64 256; do
  st=12; [ $bt = 256 ] && st=5
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/p_$bt" -o p -- python "$R/bench.py" --precision $PREC --batch $bt --steps $st --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_$bt.json" 2>/dev/null
  f=$(find "$O/p_$bt" -name "*kernel_stats.csv" | head -1)
  echo "B=$bt $(python -c "import json,sys; d=json.loads(open('$O/bench_$bt.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"
  grep -E "$PAT" "$f" | cut -d, -f1,2,4 | cut -c1-160
done
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
