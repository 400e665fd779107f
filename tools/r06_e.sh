set -u
O=gpurun_out/r06_e; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_audio.py tests/test_gpu_bench.py tests/test_gpu_tta.py -x -q > $O/tests.log 2>&1; tail -15 $O/tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_e/bench_default.json").read().strip().splitlines()[-1])
print("default", d["value"], d["ms_per_step"], "ingest", d.get("ingest"), "mix", (d.get("realistic_mix") or {}).get("value"))
PY
timeout 600 python bench.py --workload strong2048 --steps 3 --warmup 1 --no-cpu-baseline --no-post-logits --no-extra > $O/bench_strong.json 2> $O/bench_strong.err; cut -c1-400 $O/bench_strong.json; tail -3 $O/bench_strong.err
timeout 600 python bench.py --workload strong2048 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-post-logits --no-extra > $O/bench_strong_b128.json 2> $O/bench_strong_b128.err; cut -c1-200 $O/bench_strong_b128.json
