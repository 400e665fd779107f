#!/bin/bash
# round 4, GPU call 3: fused-FFN ablations; does an engine sized for 30 s clips pay for it on 10 s batches?
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04d; mkdir -p "$O"; cd "$R"
for m in 8064 32256; do timeout 120 tools/ffn_fused_bench 30 $m 1 2>&1 | tee -a "$O/ffn_fused_bench.log"; done
for capf in "" "--capacity-seconds 30"; do
  timeout 300 python bench.py --steps 40 --no-cpu-baseline --no-extra --no-post-logits $capf 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 10s capacity', d['config']['engine_capacity_seconds'], d['value'], d['ms_per_step'])"
done
timeout 300 python tools/sweep.py --warmup 8 --out "$O/sweep.json" > "$O/sweep.log" 2>&1; python -c "
import json; d=json.load(open('$O/sweep.json')); print([(r['case'], r['ms_per_batch']) for r in d['rows']])"
timeout 600 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tta.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -n 2
