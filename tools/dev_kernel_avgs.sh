#!/bin/bash
# dev: rocprofv3 kernel averages (one batch at a time) of a few kernels for several library builds:  LIBS="a.so b.so" tools/dev_kernel_avgs.sh
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in ${LIBS:?}; do
  rm -rf /tmp/ka; QVERSE_LIB=$R/offline-tarteel_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka -o p -- python "$R/bench.py" --steps 30 --contexts 1 ${BENCH_ARGS:-} --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
  python - "$lib" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ka/**/p_kernel_stats.csv", recursive=True)[0]
want = ("k_quant_rows", "k_ln_ort", "k_rows_minmax", "k_dwconv1d_ort", "k_layernormE", "k_layernorm2", "k_attention_short", "k_dwconv1d", "k_gemm256<1", "k_gemm256<6", "k_gemm<4", "k_gemm<3", "k_gemm256<4", "k_gemm256<3")
row = {}
for r in csv.DictReader(open(f)):
    for w in want:
        if w in r["Name"]: row[w] = round(float(r["AverageNs"]) / 1e3, 2)
print(sys.argv[1], row)
PY
done
