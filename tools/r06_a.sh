set -u
O=gpurun_out/r06_a; mkdir -p $O
rocm-smi --showuniqueid 2>&1 | grep -i unique | head -2 > $O/device.log
timeout 900 python tools/diag_tta_work.py > $O/diag_tta.log 2>&1
timeout 300 python bench.py --workload tta30 --steps 5 --warmup 2 --no-cpu-baseline --no-post-logits --no-extra > $O/bench_tta30.json 2> $O/bench_tta30.err
timeout 300 python bench.py --workload tta30 --tta-mix --steps 8 --warmup 3 --no-cpu-baseline --no-post-logits --no-extra > $O/bench_tta30_mix.json 2> $O/bench_tta30_mix.err
timeout 300 python tools/post_bench.py > $O/post_bench.jsonl 2>&1
timeout 300 python tools/post_bench.py --frames 376 > $O/post_bench_376.jsonl 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $O/bench_default.json 2> $O/bench_default.err
tail -30 $O/diag_tta.log; cut -c1-300 $O/bench_tta30.json $O/bench_tta30_mix.json; cat $O/post_bench.jsonl $O/post_bench_376.jsonl; cut -c1-300 $O/bench_default.json
