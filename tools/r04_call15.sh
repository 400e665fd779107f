#!/bin/bash
# conv.0 of the f16 path on the f32 matrix pipe: forward tests, bench legs, k_sub01's time in situ
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -4
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr headline
timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr headline_again
timeout 300 python bench.py --contexts 1 --steps 40 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr contexts1
timeout 300 python bench.py --batch 256 --steps 24 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr b256
timeout 300 python bench.py --workload tta30 --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | pr tta30
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof15" -o p -- python "$R/bench.py" --steps 20 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof15/p_kernel_stats.csv")))
for r in rows:
    if any(k in r['Name'] for k in ('k_sub01','k_logmel','k_dwconv2d','attention')): print(r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1))
PY
find "$R/gpurun_out/prof15" -name "*_kernel_trace.csv" -delete
