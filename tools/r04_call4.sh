#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04e; mkdir -p "$O"; cd "$R"
for m in 8064 32256; do timeout 120 tools/ffn_fused_bench 30 $m 1 2>&1 | tee -a "$O/ffn_fused_bench.log"; done
