#!/bin/bash
# round 4, GPU call 2: fused-FFN prototype (correctness + timing), suite at this commit, default bench
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04c; mkdir -p "$O"; cd "$R"
for m in 8064 32256; do timeout 120 tools/ffn_fused_bench 50 $m 1 2>&1 | tee -a "$O/ffn_fused_bench.log"; done
timeout 120 tools/ffn_fused_bench 50 8064 4 2>&1 | tee -a "$O/ffn_fused_bench.log"
timeout 1500 python -m pytest tests -m gpu -x -q -rP > "$O/tests.log" 2>&1; tail -n 3 "$O/tests.log"
grep -h "^\[ort-" "$O/tests.log" | cut -c1-300 | head -70
s0=$SECONDS; timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "default bench.py wall $((SECONDS - s0)) s"; cut -c1-200 "$O/bench.json"; tail -n 3 "$O/bench.err"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04c/bench.json"))
print({k:(v.get("value"),v.get("ms_per_step")) if isinstance(v,dict) else v for k,v in (d.get("extra") or {}).items()})
print("mix", d.get("realistic_mix",{}).get("runs_utt_per_s"), "post", {k:v.get("ms_per_batch") for k,v in d.get("post_logits",{}).items() if isinstance(v,dict)})
PY
