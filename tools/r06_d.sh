set -u
O=gpurun_out/r06_d; mkdir -p $O
R=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tracker.py tests/test_gpu_tta.py tests/test_gpu_audio.py -x -q > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout 300 python tools/post_bench.py > $O/post_bench.jsonl 2>&1; cut -c1-150 $O/post_bench.jsonl
timeout 300 python tools/post_bench.py --frames 376 > $O/post_bench_376.jsonl 2>&1; cut -c1-150 $O/post_bench_376.jsonl
A="--no-cpu-baseline --no-extra"
timeout 300 python bench.py --steps 20 --warmup 5 $A > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload tta30 --steps 5 --warmup 2 $A --no-post-logits > $O/bench_tta30.json 2> $O/bench_tta30.err
timeout 300 python bench.py --workload tta30 --tta-mix --steps 8 --warmup 3 $A --no-post-logits > $O/bench_tta30_mix.json 2> $O/bench_tta30_mix.err
python - <<'PY'
import json
for n in ("default","tta30","tta30_mix"):
    try:
        d=json.loads(open(f"gpurun_out/r06_d/bench_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], (d.get("post_logits") or {}).get("gate_fail"), (d.get("realistic_mix") or {}).get("value"))
    except Exception as e: print(n, "FAILED", e); print(open(f"gpurun_out/r06_d/bench_{n}.err").read()[-1500:])
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o p -- python $R/tools/post_bench.py --steps 5 > /dev/null 2>&1
f=$(find $R/$O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/$O/post_kernel_stats.csv && head -12 "$f" | cut -d, -f1-6 | cut -c1-140
rm -rf $R/$O/prof
