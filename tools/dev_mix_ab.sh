#!/bin/bash
# dev: realistic_mix / post_logits legs of bench.py for two library builds on the same box
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for rep in 1 2; do
for lib in "$@"; do
  QVERSE_LIB=$R/$lib timeout 300 python bench.py --steps 30 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], 'mix', d['realistic_mix']['value'], d['realistic_mix']['ms_per_step'], 'gate_fail', d['post_logits']['gate_fail']['ms_per_batch'])"
done
done
