#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04h; mkdir -p "$O"; cd "$R"
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d['config']['batches_in_flight'])"; }
for c in 3 4 5 6; do timeout 200 python bench.py --steps 60 --contexts $c --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "contexts=$c"; done
timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "again contexts=4"
timeout 200 python bench.py --batch 256 --steps 20 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "B=256 fp16"
timeout 200 python bench.py --contexts 1 --steps 40 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "contexts=1"
