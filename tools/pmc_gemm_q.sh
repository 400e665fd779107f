#!/bin/bash
# dev: SQ counters of k_gemm256 (gemm_bench mode 2) and k_gemm256q (mode 3) side by side, M = $1 (default 32256)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
M=${1:-32256}
O=$R/gpurun_out/pmc_q; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for mode in 2 3; do
  timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d "$O/a$mode" -o p -- "$R/tools/gemm_bench" 6 $M $mode > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d "$O/b$mode" -o p -- "$R/tools/gemm_bench" 6 $M $mode > "$O/b$mode.log" 2>&1
done
python - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for mode in (2, 3):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("a", "b"):
        for f in glob.glob(f"{O}/{sub}{mode}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "k_gemm256" in k:
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in sorted(acc.items()):
        m = {n: sum(v[1:]) / max(1, len(v) - 1) if len(v) > 1 else v[0] for n, v in c.items()}
        busy, wave = m.get("SQ_BUSY_CYCLES", 0), m.get("SQ_WAVE_CYCLES", 1)
        print(f"mode {mode} {k[:40]:40s} mfma_util {m.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(busy*32+1e-9):.3f}  wave cycles: parked {m.get('SQ_WAIT_ANY',0)/wave:.2f} issue-wait {m.get('SQ_WAIT_INST_ANY',0)/wave:.2f} issuing {m.get('SQ_ACTIVE_INST_ANY',0)/wave:.2f} | "
              f"lds: active {m.get('SQ_ACTIVE_INST_LDS',0)/wave:.3f} wait {m.get('SQ_WAIT_INST_LDS',0)/wave:.3f} bank-conflict cycles/idx-active {m.get('SQ_LDS_BANK_CONFLICT',0)/(m.get('SQ_LDS_IDX_ACTIVE',1) or 1):.3f} insts {m.get('SQ_INSTS_LDS',0):.0f} | valu active {m.get('SQ_ACTIVE_INST_VALU',0)/wave:.3f} insts {m.get('SQ_INSTS_VALU',0):.0f}")
PY
