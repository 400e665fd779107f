#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c5
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -x -s > "$O/tests.log" 2>&1; tail -n 15 "$O/tests.log" | cut -c1-300
timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.err"; cut -c1-200 "$O/bench.json"; tail -n 3 "$O/bench.err"
timeout 300 python bench.py --contexts 1 --no-cpu-baseline --no-extra > "$O/bench_contexts1.json" 2>/dev/null; cut -c1-140 "$O/bench_contexts1.json"
timeout 200 python tools/post_bench.py > "$O/post_bench.jsonl" 2>/dev/null; cut -c1-160 "$O/post_bench.jsonl"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof1" -o p -- python "$R/bench.py" --steps 20 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profpost" -o p -- python "$R/tools/post_bench.py" --steps 5 > /dev/null 2>&1
cd "$R"
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
