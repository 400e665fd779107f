set -u
O=gpurun_out/r06_j; mkdir -p $O
A="--no-cpu-baseline --no-extra --no-post-logits"
for c in 3 4 3 4; do
timeout 300 python bench.py --workload tta30 --contexts $c --steps 8 --warmup 3 $A > $O/t.json 2>$O/t.err; python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('tta30 contexts $c', d['value'], d['ms_per_step'])" || tail -3 $O/t.err
timeout 300 python bench.py --workload tta30 --tta-mix --contexts $c --steps 10 --warmup 3 $A > $O/t.json 2>$O/t.err; python -c "
import json; d=json.loads(open('$O/t.json').read().strip().splitlines()[-1]); print('tta30-mix contexts $c', d['value'], d['ms_per_step'], d['config']['tta_gated_fraction'])" || tail -3 $O/t.err
done
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -k "tta30" 2>&1 | tail -3
