#!/bin/bash
# (historical: the split this script measured lost -- tta30 1,505 -> 1,410 utt/s -- and was reverted)
# long CTC targets in their own kernel: post-logits / TTA / forward tests, then the TTA 30 s workload and the headline
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tta.py tests/test_gpu_tracker.py tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -4
pr() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
timeout 300 python bench.py --workload tta30 --steps 6 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | pr tta30
timeout 300 python bench.py --workload tta30 --precision ort --steps 4 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | pr tta30_ort
timeout 300 python bench.py --steps 80 --no-cpu-baseline --no-extra 2>/dev/null | pr headline
timeout 300 python bench.py --seconds 40 --capacity-seconds 40 --steps 10 --no-cpu-baseline --no-extra 2>/dev/null | pr clips_40s
timeout 120 tools/att_bench 64 126 200 > gpurun_out/att_bench.log 2>&1; timeout 120 tools/att_bench 256 126 100 >> gpurun_out/att_bench.log 2>&1; timeout 120 tools/att_bench 64 376 50 >> gpurun_out/att_bench.log 2>&1
grep variant gpurun_out/att_bench.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_tta30" -o p -- python "$R/bench.py" --workload tta30 --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
grep -i "k_ctc" "$R/gpurun_out/prof_tta30/p_kernel_stats.csv" | awk -F, '{print substr($1,1,70), $2, $4}'
find "$R/gpurun_out/prof_tta30" -name "*_kernel_trace.csv" -delete
