set -u
O=gpurun_out/r06_h; mkdir -p $O
timeout 120 tools/att_bench 64 376 50 > $O/att_bench_376.log 2>&1; head -3 $O/att_bench_376.log | cut -c1-200; sed -n 5,6p $O/att_bench_376.log | cut -c1-600
timeout 120 tools/att_bench 64 251 50 > $O/att_bench_251.log 2>&1; head -3 $O/att_bench_251.log | cut -c1-200
timeout 120 tools/att_bench 64 200 50 2>&1 | head -3 | cut -c1-200
timeout 2400 python -m pytest tests/test_gpu_forward.py -x -q > $O/tests_forward.log 2>&1; tail -4 $O/tests_forward.log
A="--no-cpu-baseline --no-extra --no-post-logits"
timeout 300 python bench.py --workload tta30 --steps 5 --warmup 2 $A > $O/bench_tta30.json 2>/dev/null
timeout 300 python bench.py --workload tta30 --tta-mix --steps 8 --warmup 3 $A > $O/bench_tta30_mix.json 2>/dev/null
timeout 300 python bench.py --seconds 30 --steps 10 --warmup 3 $A > $O/bench_30s.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_h/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
