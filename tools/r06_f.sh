set -u
O=gpurun_out/r06_f; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_interference.py tests/test_gpu_gemm256.py tests/test_gpu_ort_mixed.py -x -q > $O/tests.log 2>&1; tail -6 $O/tests.log
grep "ort-e2e" $O/tests.log | cut -c1-300
for b in gemm_bench gemm_bench_nopk; do for M in 8064 32256; do echo "== $b M=$M"; timeout 300 tools/$b 50 $M 2 2>&1 | grep -v amdgpu.ids | cut -c1-100; done; done > $O/gemm_packed_vs_not.log 2>&1; cat $O/gemm_packed_vs_not.log
A="--no-cpu-baseline --no-extra --no-post-logits"
for r in 1 2; do
timeout 300 python bench.py --steps 40 --warmup 5 $A > $O/bench_default_$r.json 2>/dev/null
timeout 300 python bench.py --batch 256 --steps 12 --warmup 3 $A > $O/bench_b256_$r.json 2>/dev/null
timeout 300 python bench.py --batch 256 --precision mixed --steps 12 --warmup 3 $A > $O/bench_b256_mixed_$r.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_f/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "FAILED", e)
PY
timeout 900 python tools/ort_floor_table.py --out $O/ort_floor_table.json > $O/ort_floor_table.log 2>&1; tail -4 $O/ort_floor_table.log | cut -c1-400
