#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run() { env "$@" timeout 200 python bench.py --steps 80 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "$*"; }
run X=default
run QVERSE_GEMM_MINTILES_LONGK=100000
run QVERSE_GEMM_MINTILES=100
run QVERSE_GEMM_MINTILES=160
run QVERSE_GEMM_MINTILES=200
run QVERSE_GEMM_T256=0
run X=default_again
