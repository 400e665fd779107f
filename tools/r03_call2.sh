#!/bin/bash
# round-3 evidence call: ORT-mixed bench lines + B = 256 kernel tables (usage: tools/r03_call2.sh)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c2
mkdir -p "$O"
cd "$R"
timeout 400 python -m pytest tests/test_gpu_ort_mixed.py -q -s -k "batch256" > "$O/ort_b256_test.log" 2>&1; tail -n 6 "$O/ort_b256_test.log"
timeout 200 python bench.py --steps 30 --no-cpu-baseline --no-post-logits > "$O/bench_fp16_b64.json" 2> "$O/bench_fp16_b64.err"; cut -c1-140 "$O/bench_fp16_b64.json"
for prec in ort mixed; do
  timeout 200 python bench.py --precision $prec --steps 30 --no-cpu-baseline --no-post-logits > "$O/bench_${prec}_b64.json" 2> "$O/bench_${prec}_b64.err"; cut -c1-140 "$O/bench_${prec}_b64.json"; tail -n 2 "$O/bench_${prec}_b64.err"
done
for prec in fp16 mixed ort; do
  timeout 300 python bench.py --precision $prec --batch 256 --steps 12 --no-cpu-baseline --no-post-logits > "$O/bench_${prec}_b256.json" 2> "$O/bench_${prec}_b256.err"; cut -c1-140 "$O/bench_${prec}_b256.json"
done
timeout 300 python tools/ort_delta.py --seconds 3 --out "$O/ort_semantics_delta.json" > /dev/null 2> "$O/ort_delta.err"; grep -A3 hip_ort "$O/ort_semantics_delta.json"
cd /tmp && export TMPDIR=/tmp
for prec in fp16 mixed ort; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b256_$prec" -o p -- python "$R/bench.py" --precision $prec --batch 256 --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits > "$O/bench_${prec}_b256_contexts1_under_rocprof.json" 2>/dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b64_ort" -o p -- python "$R/bench.py" --precision ort --steps 16 --contexts 1 --no-cpu-baseline --no-post-logits > "$O/bench_ort_b64_contexts1_under_rocprof.json" 2>/dev/null
cd "$R"
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
ls -R "$O" | head -40
