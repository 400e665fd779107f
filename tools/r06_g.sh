set -u
O=gpurun_out/r06_g; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_interference.py -x -q > $O/tests_interference.log 2>&1; tail -4 $O/tests_interference.log
timeout 120 tools/att_bench 64 376 50 > $O/att_bench_376.log 2>&1; cut -c1-1500 $O/att_bench_376.log
timeout 120 tools/att_bench 64 251 50 > $O/att_bench_251.log 2>&1; head -4 $O/att_bench_251.log | cut -c1-200
