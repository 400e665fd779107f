set -u
O=gpurun_out/r06_r; mkdir -p $O; R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for c in 0 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$c -o p -- python $R/tools/post_bench.py --steps 10 --case $c > /dev/null 2>&1
cp $R/$O/prof$c/p_kernel_stats.csv $R/$O/post_case${c}_kernel_stats.csv; rm -rf $R/$O/prof$c
done
