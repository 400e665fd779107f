set -u
O=gpurun_out/r06_p; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_forward.py tests/test_gpu_interference.py tests/test_gpu_fullsize.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
A="--no-cpu-baseline --no-extra --no-post-logits"
for v in 1 0 1 0; do
echo "QVERSE_ATT_WS=$v"
QVERSE_ATT_WS=$v timeout 300 python bench.py --workload tta30 --steps 8 --warmup 3 $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' tta30', d['value'], d['ms_per_step'])"
QVERSE_ATT_WS=$v timeout 300 python bench.py --workload tta30 --tta-mix --steps 10 --warmup 3 $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' tta30-mix', d['value'], d['ms_per_step'])"
QVERSE_ATT_WS=$v timeout 300 python bench.py --seconds 30 --steps 12 --warmup 4 $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' 30s', d['value'], d['ms_per_step'])"
QVERSE_ATT_WS=$v timeout 300 python bench.py --steps 40 --warmup 6 $A 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' 10s', d['value'], d['ms_per_step'])"
done
