set -u
O=gpurun_out/r06_c; mkdir -p $O
R=$(pwd)
timeout 1500 python -m pytest tests/test_gpu_postlogits.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
for V in 0 1; do
  QVERSE_FRAG=$V timeout 300 python tools/post_bench.py > $O/post_bench_frag$V.jsonl 2>&1
  cut -c1-150 $O/post_bench_frag$V.jsonl
done
cd /tmp && export TMPDIR=/tmp
for V in 0 1; do
  QVERSE_FRAG=$V timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_frag$V -o p -- python $R/tools/post_bench.py --steps 5 > /dev/null 2>&1
  f=$(find $R/$O/prof_frag$V -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/post_kernel_stats_frag$V.csv; head -12 "$f" | cut -d, -f1-6 | cut -c1-140
  rm -rf $R/$O/prof_frag$V
done
cd $R
bash tools/dev_pmc_post.sh "k_frag|k_lcs_full|k_spans2|k_trigram" > $O/pmc_post.txt 2>&1; cat $O/pmc_post.txt
