#!/bin/bash
# (historical: the prefetching variant QVERSE_ATT_OLD=3 this script measured was removed after it; see DESIGN.md, kernel table)
# attention variants again: the one-wave-per-tile kernel bounded to 2 waves/SIMD (variant 2) and with prefetched fragments (3)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -5
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-post-logits $EXTRA 2>/dev/null | pr "$tag $EXTRA"; }
EXTRA="--steps 80"; run default X=1; run att_old2 QVERSE_ATT_OLD=1; run att_pf3 QVERSE_ATT_OLD=3; run default_again X=1; run att_pf3_again QVERSE_ATT_OLD=3
EXTRA="--batch 256 --steps 24"; run default X=1; run att_old2 QVERSE_ATT_OLD=1; run att_pf3 QVERSE_ATT_OLD=3
EXTRA="--contexts 1 --steps 40"; run default X=1; run att_old2 QVERSE_ATT_OLD=1; run att_pf3 QVERSE_ATT_OLD=3
cd /tmp && export TMPDIR=/tmp
for v in 0 1 3; do
  e=QVERSE_ATT_OLD=$v; [ $v = 0 ] && e=X=1
  env $e rocprofv3 --kernel-trace --stats -d $R/gpurun_out/att_v$v -o t -- python $R/bench.py --contexts 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-post-logits >/dev/null 2>&1
  f=$(find $R/gpurun_out/att_v$v -name '*kernel_stats.csv' | head -1)
  echo "variant env $e"; grep -i attention "$f" | cut -c1-60,170-400 | head -3
done
