#!/bin/bash
# last GPU call of round 4: soak of batches in flight in the three precisions, then the evidence round (tests first)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
for p in 0 1 2; do timeout 300 python tools/soak.py --batches 3000 --seed $((11 + p)) --precision $p --third 2>&1 | grep -v amdgpu.ids | cut -c1-300; done > gpurun_out/soak_r04t.log 2>&1
timeout 300 python tools/dev_ort_race.py --batches 400 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -3 >> gpurun_out/soak_r04t.log
cat gpurun_out/soak_r04t.log
bash tools/final_round.sh r04t
