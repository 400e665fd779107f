#!/bin/bash
# round 4, GPU call 1: suite at this commit + bench + short-clip / sweep discrepancy measurements
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04b; mkdir -p "$O"; cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1; tail -n 3 "$O/tests.log"
grep -h "\[ort-" "$O/tests.log" | cut -c1-260 | head -60
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -n 1 "$O/smoke.log" | cut -c1-400
s0=$SECONDS; timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "default bench.py wall $((SECONDS - s0)) s"; cut -c1-200 "$O/bench.json"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04b/bench.json"))
print({k:(v.get("value"),v.get("ms_per_step")) if isinstance(v,dict) else v for k,v in (d.get("extra") or {}).items()})
print("mix", d.get("realistic_mix",{}).get("runs_utt_per_s"), "post", {k:v.get("ms_per_batch") for k,v in d.get("post_logits",{}).items() if isinstance(v,dict)})
PY
timeout 300 python tools/sweep.py --warmup 8 --out "$O/sweep_w8.json" > "$O/sweep.log" 2>&1; python -c "
import json; d=json.load(open('$O/sweep_w8.json')); print([(r['case'], r['ms_per_batch']) for r in d['rows']])"
