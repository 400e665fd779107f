#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04p; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_ort_mixed.py -m gpu -x -q -rP > "$O/tests_ort.log" 2>&1; tail -n 3 "$O/tests_ort.log"; grep -h "conv.0+conv.2\|\[ort-e2e\] random" "$O/tests_ort.log" | head -6 | cut -c1-200
timeout 300 python bench.py --batch 256 --precision ort --steps 12 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=256 ort', d['value'], d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b256_ort" -o p -- python "$R/bench.py" --precision ort --batch 256 --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
cd "$R"; find "$O" -name "*_kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04p/prof_b256_ort/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "sub01" in r["Name"] or "dwconv2d_ort" in r["Name"]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg")
PY
