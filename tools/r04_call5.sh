#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04f; mkdir -p "$O"; cd "$R"
timeout 900 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tracker.py tests/test_gpu_tta.py -m gpu -x -q 2>&1 | tail -n 3
timeout 200 python tools/post_bench.py > "$O/post_bench.jsonl" 2>/dev/null; cut -c1-160 "$O/post_bench.jsonl"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profpost" -o p -- python "$R/tools/post_bench.py" --steps 5 > /dev/null 2>&1
cd "$R"; find "$O" -name "*_kernel_trace.csv" -delete
f=$(find "$O/profpost" -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 "$f" | head -14
