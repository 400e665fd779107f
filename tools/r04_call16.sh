#!/bin/bash
# precision-2 soak mismatches: which knob makes them go away
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
run() { tag=$1; shift; echo "== $tag"; env "$@" timeout 200 python tools/soak.py --batches 500 --seed 13 --precision 2 2>&1 | grep -v amdgpu.ids | tail -4; }
run default X=1
run att_tiled QVERSE_ATT_TILED=1
run tiles128 QVERSE_GEMM_T256=0
run tiles256 QVERSE_GEMM_T256=1
