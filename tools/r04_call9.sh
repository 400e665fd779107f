#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], 'utt/s', d['ms_per_step'], 'ms/step')"; }
for b in 1 4 16; do timeout 200 python bench.py --batch $b --contexts 1 --steps 100 --warmup 10 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "B=$b contexts=1"; done
timeout 200 python bench.py --batch 1 --contexts 4 --steps 200 --warmup 10 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | pr "B=1 contexts=4"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r04s/prof_b1" -o p -- python "$R/bench.py" --batch 1 --contexts 1 --steps 50 --warmup 5 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
cd "$R"; find gpurun_out/r04s -name "*_kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04s/prof_b1/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
steps=[int(r["Calls"]) for r in rows if "k_decode" in r["Name"]][0]
print("launches per step", sum(int(r["Calls"]) for r in rows)/steps, "kernel ms per step", sum(float(r["TotalDurationNs"]) for r in rows)/steps/1e6)
for r in rows[:8]: print("  ", r["Name"][:60], int(r["Calls"])//steps, round(float(r["AverageNs"])/1e3,1))
PY
