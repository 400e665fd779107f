#!/bin/bash
# dev (needs `python offline-tarteel_amd/build.py --dev-hooks`): marginal cost of a kernel class under the current overlap --
# QVERSE_SKIP drops the class (results garbage), QVERSE_DUP launches it twice (results unchanged)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/skip
mkdir -p "$O"
cd "$R"
run() {  # label, env assignment, extra flags
  for rep in 1 2; do
    env $2 timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-post-logits --no-extra $3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', '$3', d['value'], d['ms_per_step'])"
  done
}
for ctx in "" "--contexts 1"; do
  run base X=0 "$ctx"
  run skip_ln QVERSE_SKIP=1 "$ctx"
  run skip_ln_ln2 QVERSE_SKIP=3 "$ctx"
  run skip_att QVERSE_SKIP=4 "$ctx"
  run skip_dw QVERSE_SKIP=8 "$ctx"
  run skip_front QVERSE_SKIP=32 "$ctx"
  run dup_ln QVERSE_DUP=1 "$ctx"
  run dup_att QVERSE_DUP=4 "$ctx"
  run dup_dw QVERSE_DUP=8 "$ctx"
done | tee "$O/skip_sweep.txt"
