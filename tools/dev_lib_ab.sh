#!/bin/bash
# dev: same-box A/B of two builds of the library (QVERSE_LIB): per-kernel averages one batch at a time + bench lines.
# usage: tools/dev_lib_ab.sh <tag> <kernel-name-regex> <libA> <libB>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}; PAT=${2:-.}; LA=$R/${3}; LB=$R/${4}
O=$R/gpurun_out/lab_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$LA" "$LB"; do
  n=$(basename "$lib" .so)
  QVERSE_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/p_${n}_$rep" -o p -- python "$R/bench.py" --steps 16 --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
  f=$(find "$O/p_${n}_$rep" -name "*kernel_stats.csv" | head -1)
  echo "$n rep $rep: $(grep -E "$PAT" "$f" | cut -d, -f2,4 | tr '\n' ' ')"
  for c in "" "--contexts 1"; do
    QVERSE_LIB=$lib timeout 200 python "$R/bench.py" --steps 60 --no-cpu-baseline --no-post-logits --no-extra $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   $n $c', d['value'], d['ms_per_step'])"
  done
done
done | tee "$O/ab.txt"
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*agent_info.csv" -delete
