#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c7
mkdir -p "$O"
cd "$R"
timeout 900 python -m pytest tests -m gpu -q -x > "$O/tests.log" 2>&1; tail -n 12 "$O/tests.log" | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline --no-extra > "$O/bench.json" 2> "$O/bench.err"; cut -c1-200 "$O/bench.json"
timeout 300 python bench.py --contexts 1 --no-cpu-baseline --no-extra --no-post-logits > "$O/bench_contexts1.json" 2>/dev/null; cut -c1-140 "$O/bench_contexts1.json"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof1" -o p -- python "$R/bench.py" --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
cd "$R"
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/c7"
f = glob.glob(O + "/prof1/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
out = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "anonymous" in n and ("k_lcs_full" in n or "k_frag" in n or "k_ctc" in n or "k_topk" in n or "k_trigram" in n or "k_decode" in n):
        out[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(O + "/post_kernel_durations.txt", "w") as w:
    for k, v in out.items():
        w.write(k + " " + " ".join(f"{x:.0f}" for x in v[-24:]) + "\n")
print(open(O + "/post_kernel_durations.txt").read())
PY
find "$O" -name "*_kernel_trace.csv" -delete
find "$O" -name "*agent_info.csv" -delete
