#!/bin/bash
# PMC evidence for the GEMM family in one GPU-box call (separate rocprofv3 --pmc passes, kernel trace only):
#   FETCH_SIZE, WRITE_SIZE (HBM traffic) and one SQ pass (MFMA busy, wave-cycle split) over tools/gemm_bench.
# usage: tools/pmc_round.sh <tag> [rows]     -> gpurun_out/pmc_<tag>/{traffic,mfma}.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-x}; ROWS=${2:-8064}
O=$R/gpurun_out/pmc_$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# tile height fixed to 256 rows: what the engine runs with three or more batches in flight (the bench line the numbers are quoted in);
# left to the policy, a lone gemm_bench launch at M = 8,064 picks 192-row tiles for the N = 512 shapes and the kernel names no longer match
export QVERSE_GEMM_BM=0
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -o p -- "$R/tools/gemm_bench" 10 $ROWS > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -o p -- "$R/tools/gemm_bench" 10 $ROWS > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/sq" -o p -- "$R/tools/gemm_bench" 10 $ROWS > "$O/sq.log" 2>&1
cd "$R"
F=$(find "$O/fetch" -name "*counter_collection.csv" | head -1); W=$(find "$O/write" -name "*counter_collection.csv" | head -1); S=$(find "$O/sq" -name "*counter_collection.csv" | head -1)
python tools/pmc_traffic.py "$F" "$W" "$O/traffic.json" $ROWS > /dev/null
python tools/pmc_traffic.py --mfma "$S" "$O/mfma.json" $ROWS > /dev/null
find "$O" -name "*_kernel_trace.csv" -delete; find "$O" -name "*counter_collection.csv" -size +4M -delete
ls -la "$O"; head -c 1500 "$O/mfma.json"
