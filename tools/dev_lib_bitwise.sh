#!/bin/bash
# dev: the log-probs of a ragged seeded batch from two library builds, compared bitwise (tools/ab_forward.py)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
QVERSE_LIB=$R/$1 timeout 120 python tools/ab_forward.py /tmp/lp_a.pt > /dev/null 2>&1
QVERSE_LIB=$R/$2 timeout 120 python tools/ab_forward.py /tmp/lp_b.pt > /dev/null 2>&1
python - <<'PY'
import torch
a, b = torch.load("/tmp/lp_a.pt"), torch.load("/tmp/lp_b.pt")
d = float((a["lp"] - b["lp"]).abs().max())
print("bitwise:", "identical" if torch.equal(a["lp"], b["lp"]) else f"DIFFERS, max |d| {d:g}")
PY
