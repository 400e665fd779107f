#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c9
mkdir -p "$O"
cd "$R"
for c in 3 4 5 6; do
  for rep in 1 2; do
    timeout 200 python bench.py --contexts $c --steps 60 --no-cpu-baseline --no-post-logits --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ctx', $c, d['value'], d['ms_per_step'], d['config']['batches_in_flight'])"
  done
done | tee "$O/contexts_sweep.txt"
QVERSE_POST_GRAPH=1 timeout 200 python bench.py --steps 60 --no-cpu-baseline --no-post-logits --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('postgraph', d['value'], d['ms_per_step'])" | tee -a "$O/contexts_sweep.txt"
for prec in ort mixed; do
  timeout 200 python bench.py --precision $prec --steps 30 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_${prec}_b64.json" 2>/dev/null; cut -c1-140 "$O/bench_${prec}_b64.json"
  timeout 300 python bench.py --precision $prec --batch 256 --steps 12 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_${prec}_b256.json" 2>/dev/null; cut -c1-140 "$O/bench_${prec}_b256.json"
done
timeout 300 python bench.py --batch 256 --steps 12 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_fp16_b256.json" 2>/dev/null; cut -c1-140 "$O/bench_fp16_b256.json"
cd /tmp && export TMPDIR=/tmp
for prec in ort; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b256_$prec" -o p -- python "$R/bench.py" --precision $prec --batch 256 --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b64_$prec" -o p -- python "$R/bench.py" --precision $prec --steps 16 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
done
cd "$R"
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
