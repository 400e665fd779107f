#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/c6
mkdir -p "$O"
cd "$R"
python tools/hwq_probe.py early > "$O/probe_early.json" 2>/dev/null; cat "$O/probe_early.json"
python tools/hwq_probe.py late > "$O/probe_late.json" 2>/dev/null; cat "$O/probe_late.json"
HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29741 tools/hwq_probe.py dist > "$O/probe_dist.json" 2>"$O/probe_dist.err"; grep rounds "$O/probe_dist.json"
for v in 17 19; do
  QVERSE_LIB=$R/offline-tarteel_amd/build/alt/libqverse_pms$v.so timeout 200 python tools/post_bench.py > "$O/post_bench_pms$v.jsonl" 2>/dev/null; echo pms$v; cut -c1-110 "$O/post_bench_pms$v.jsonl"
done
timeout 200 python tools/post_bench.py > "$O/post_bench.jsonl" 2>/dev/null; echo base; cut -c1-110 "$O/post_bench.jsonl"
