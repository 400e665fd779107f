set -u
O=gpurun_out/r06_i; mkdir -p $O; R=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_audio.py tests/test_gpu_tta.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
A="--no-cpu-baseline --no-extra --no-post-logits"
timeout 300 python bench.py --workload tta30 --steps 6 --warmup 2 $A > $O/bench_tta30.json 2>/dev/null
timeout 300 python bench.py --workload tta30 --tta-mix --steps 8 --warmup 3 $A > $O/bench_tta30_mix.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_i/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --workload tta30 --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-extra --no-post-logits > /dev/null 2>&1
grep -i "upfirdn" $R/$O/prof/p_kernel_stats.csv | cut -c1-200
cp $R/$O/prof/p_kernel_stats.csv $R/$O/tta30_kernel_stats.csv; rm -rf $R/$O/prof
