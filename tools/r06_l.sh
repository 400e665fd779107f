set -u
O=gpurun_out/r06_l; mkdir -p $O; R=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tta.py -x -q > $O/tests.log 2>&1; tail -2 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o p -- python $R/bench.py --workload tta30 --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-extra --no-post-logits > /dev/null 2>&1
grep -i "k_ctc\|k_candidates" $R/$O/prof/p_kernel_stats.csv | cut -c1-230
rm -rf $R/$O/prof
cd $R
for i in 1 2; do timeout 300 python bench.py --workload tta30 --steps 8 --warmup 3 --no-cpu-baseline --no-extra --no-post-logits 2>/dev/null | cut -c1-150; done
timeout 300 python tools/post_bench.py 2>/dev/null | cut -c1-120
