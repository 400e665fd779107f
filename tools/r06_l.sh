set -u
O=gpurun_out/r06_l; mkdir -p $O; R=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tracker.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python tools/post_bench.py 2>/dev/null | cut -c1-120
timeout 300 python tools/post_bench.py --frames 376 2>/dev/null | cut -c1-120
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/tools/post_bench.py --steps 5 > /dev/null 2>&1
head -8 $R/$O/prof/p_kernel_stats.csv | cut -d, -f1-4,7,8 | cut -c1-160
rm -rf $R/$O/prof
