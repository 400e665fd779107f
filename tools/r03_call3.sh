#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c3
mkdir -p "$O"
cd "$R"
timeout 500 python -m pytest tests/test_gpu_ort_mixed.py -q -s --maxfail=20 -k "not batch256" > "$O/ort_tests.log" 2>&1; tail -n 4 "$O/ort_tests.log"
timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-post-logits > "$O/bench_fp16_b64.json" 2> "$O/bench_fp16_b64.err"; cut -c1-140 "$O/bench_fp16_b64.json"
for prec in ort; do
  timeout 200 python bench.py --precision $prec --steps 30 --no-cpu-baseline --no-post-logits > "$O/bench_${prec}_b64.json" 2> "$O/bench_${prec}_b64.err"; cut -c1-140 "$O/bench_${prec}_b64.json"; tail -n 2 "$O/bench_${prec}_b64.err"
  timeout 300 python bench.py --precision $prec --batch 256 --steps 12 --no-cpu-baseline --no-post-logits > "$O/bench_${prec}_b256.json" 2> "$O/bench_${prec}_b256.err"; cut -c1-140 "$O/bench_${prec}_b256.json"
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b64_ort" -o p -- python "$R/bench.py" --precision ort --steps 16 --contexts 1 --no-cpu-baseline --no-post-logits > "$O/bench_ort_b64_contexts1_under_rocprof.json" 2>/dev/null
cd "$R"
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
