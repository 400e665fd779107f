#!/bin/bash
# short-utterance attention kernel: forward tests, then A/B against the key-tiled kernel (QVERSE_ATT_TILED=1)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_forward.py -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -15
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-post-logits $EXTRA 2>/dev/null | pr "$tag $EXTRA"; }
EXTRA="--steps 80"; run default X=1; run tiled QVERSE_ATT_TILED=1; run default_again X=1; run tiled_again QVERSE_ATT_TILED=1
EXTRA="--batch 256 --steps 24"; run default X=1; run tiled QVERSE_ATT_TILED=1
EXTRA="--contexts 1 --steps 40"; run default X=1; run tiled QVERSE_ATT_TILED=1
EXTRA="--seconds 5 --steps 80"; run default X=1; run tiled QVERSE_ATT_TILED=1
EXTRA="--precision ort --batch 256 --steps 16"; run default X=1; run tiled QVERSE_ATT_TILED=1
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/att13_v$v -o t -- env QVERSE_ATT_TILED=$v python $R/bench.py --contexts 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-post-logits >/dev/null 2>&1
  f=$(find $R/gpurun_out/att13_v$v -name '*kernel_stats.csv' | head -1)
  echo "QVERSE_ATT_TILED=$v"; grep -i attention "$f" | awk -F, '{print substr($1,1,60), $2, $4}' | head -3
done
