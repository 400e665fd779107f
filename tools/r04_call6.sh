#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04g; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tracker.py tests/test_gpu_tta.py -m gpu -x -q 2>&1 | tail -n 3
timeout 200 python tools/post_bench.py > "$O/post_bench.jsonl" 2>/dev/null; cut -c1-130 "$O/post_bench.jsonl"
timeout 200 python tools/tracker_bench.py --cpu-texts 2 > "$O/tracker_bench.jsonl" 2>/dev/null; tail -n 3 "$O/tracker_bench.jsonl" | cut -c1-220
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profpost" -o p -- python "$R/tools/post_bench.py" --steps 5 > /dev/null 2>&1
cd "$R"; find "$O" -name "*_kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r04g/profpost/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:56].ljust(56), r["Calls"], round(float(r["AverageNs"])/1e3,1), "us avg", round(float(r["MaxNs"])/1e3,1), "max")
PY
