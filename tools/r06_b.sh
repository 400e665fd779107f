set -u
O=gpurun_out/r06_b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_gemm256.py tests/test_gpu_postlogits.py -x -q > $O/tests.log 2>&1
tail -5 $O/tests.log
for M in 24064 21632 26432 8064 32256; do for BM in 0 3; do
  echo "== M=$M QVERSE_GEMM_BM=$BM" >> $O/gemm_bench.log
  QVERSE_GEMM_BM=$BM timeout 300 tools/gemm_bench 50 $M 2 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O/gemm_bench.log
done; done
cat $O/gemm_bench.log
b() { name=$1; shift; env "$@" > /dev/null 2>&1; }
run() { name=$1; shift; ( "$@" > $O/bench_$name.json 2> $O/bench_$name.err ); python - $O/bench_$name.json $name <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"])
except Exception as e: print(sys.argv[2], "FAILED", e)
PY
}
A="--no-cpu-baseline --no-post-logits --no-extra"
for BM in 0 1 2; do
run tta30_bm$BM env QVERSE_GEMM_BM=$BM timeout 300 python bench.py --workload tta30 --steps 5 --warmup 2 $A
run tta30_c1_bm$BM env QVERSE_GEMM_BM=$BM timeout 300 python bench.py --workload tta30 --contexts 1 --steps 5 --warmup 2 $A
run tta30mix_bm$BM env QVERSE_GEMM_BM=$BM timeout 300 python bench.py --workload tta30 --tta-mix --steps 8 --warmup 3 $A
run default_bm$BM env QVERSE_GEMM_BM=$BM timeout 300 python bench.py --steps 20 --warmup 5 $A
run c1_bm$BM env QVERSE_GEMM_BM=$BM timeout 300 python bench.py --contexts 1 --steps 20 --warmup 5 $A
done
run tta30_oldctc env QVERSE_CTC=0 timeout 300 python bench.py --workload tta30 --steps 5 --warmup 2 $A
timeout 300 python tools/post_bench.py > $O/post_bench.jsonl 2>&1
QVERSE_CTC=0 timeout 300 python tools/post_bench.py > $O/post_bench_oldctc.jsonl 2>&1
cat $O/post_bench.jsonl $O/post_bench_oldctc.jsonl | cut -c1-200
