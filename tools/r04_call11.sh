#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-post-logits $EXTRA 2>/dev/null | pr "$tag $EXTRA"; }
EXTRA="--steps 80"; run default X=1; run att_old QVERSE_ATT_OLD=1; run att_hpb1 QVERSE_ATT_HPB=1; run default_again X=1
EXTRA="--batch 256 --steps 24"; run default X=1; run att_old QVERSE_ATT_OLD=1; run att_hpb1 QVERSE_ATT_HPB=1
EXTRA="--contexts 1 --steps 40"; run default X=1; run att_old QVERSE_ATT_OLD=1
