#!/bin/bash
# One parametrised script for the GPU-box calls of a round (replaces the per-call r0N_call*.sh one-offs):
#   tools/gpu_round.sh <tag> <stage> [<stage> ...]        everything lands in gpurun_out/<tag>/
# stages:
#   interference   the k_logmel disturbance protocol (device identity, probe matrix of victims x aggressors, a run with the
#                  shader clock capped, the attribution dump) -> interference_*.log
#   newtests       the GPU tests added this round (variant equality, the interference regression test)
#   ab_frontend    headline / precision-2 bench lines with the log-mel and conv.0 variants switched (QVERSE_LOGMEL, QVERSE_ORT_SUB)
#   soak           tools/soak.py in the three precisions + tools/dev_ort_race.py
#   suite          the full -m gpu suite + smoke
#   final          tools/final_round.sh (tests, smoke, bench lines per configuration, rocprofv3 tables, PMC passes)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:?tag}; shift
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
filt() { grep -v amdgpu.ids | cut -c1-400; }
bench1() { # name, env assignments..., -- bench args
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-cpu-baseline --no-extra --no-post-logits "$@" > "$O/bench_$name.json" 2> "$O/bench_$name.err"
  python - "$O/bench_$name.json" "$name" <<'EOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]}: {d['value']:.0f} {d['unit']} {d['ms_per_step']:.3f} ms/step")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
EOF
}
for stage in "$@"; do
case $stage in
interference)
  L=$O/interference
  { rocm-smi --showuniqueid --showserial --showclocks --showpower --showperflevel --showvoltage 2>&1 | grep -v "^$" | head -60; } > "${L}_device.log" 2>&1
  P=tools/interference_probe
  ${P}_withdrawn 1 0 99 2>&1 | filt > "${L}_lane_primitives.log"
  {
    echo "== control: no aggressor"
    ${P}_withdrawn 200 0 0 2>&1 | filt
    echo "== the shipped kernel(s) next to the withdrawn kernel: range pass (1), real pass (2), both (3)"
    for v in 0 1; do for a in 1 2 3; do echo "-- QVERSE_LOGMEL=$v aggressor $a"; QVERSE_LOGMEL=$v ${P}_withdrawn 600 $a 0 2>&1 | filt; done; done
    echo "== victims next to the withdrawn range pass: 6 plain copy, 7 register FFT, 10 one frame per block + __syncthreads, 11 split re/im, 12 copy with attribution dump"
    for v in 6 7 10 11 12; do ${P}_withdrawn 600 1 $v 2>&1 | filt; done
    echo "== second attribution run (both passes)"
    ${P}_withdrawn 600 3 12 2>&1 | filt
    echo "== the precision-2 front end that ships (conv.0 on the f32 matrix pipe) as the aggressor, both log-mel kernels"
    for v in 0 1; do QVERSE_LOGMEL=$v QVERSE_ORT_SUB=1 ${P}_current 600 3 0 2>&1 | filt; done
    echo "== the other kernels that share the chip with it: k_sub01 (4), k_attention_short (8), k_attention_ws (16)"
    for a in 4 8 16; do QVERSE_LOGMEL=1 ${P}_current 300 $a 0 2>&1 | filt; done
  } > "${L}_probe_matrix.log" 2>&1
  # (round 5 also tried a shader-clock cap here; the pool no longer allows changing machine settings, and the attempt was inconclusive:
  # profiles/archive/r05_a_interference_clock_cap.log)
  cat "${L}_device.log" | head -30; cat "${L}_probe_matrix.log"
  ;;
interference2)
  P=tools/interference_probe
  {
    echo "== five-stage attribution (victim 13), withdrawn range pass"
    ${P}_withdrawn 600 1 13 2>&1 | filt
    echo "== the unpack's magnitude: 14 no square root, 15 raw v_sqrt_f32 + 32 idle cycles, 16 raw v_sqrt_f32"
    for v in 14 15 16; do ${P}_withdrawn 600 1 $v 2>&1 | filt; done
  } > "$O/interference_stage_attribution.log" 2>&1
  cat "$O/interference_stage_attribution.log"
  ;;
interference3)   # (r05_a was taken with flavour names withdrawn = packed FP32 on, withdrawn_nopk = off; since then the probe is
  # built like the product: withdrawn = off, withdrawn_pk = on)
  P=tools/interference_probe
  {
    echo "== victims compiled like the product (no packed-FP32 instructions) next to the withdrawn kernel: 0 shipped, 6 copy, 13 five-stage dump, 7 register FFT"
    for v in 0 6 13 7; do ${P}_withdrawn 600 1 $v 2>&1 | filt; done
    echo "== both passes, both log-mel kernels"
    for v in 0 1; do QVERSE_LOGMEL=$v ${P}_withdrawn 600 3 0 2>&1 | filt; done
    echo "== positive control: the same victims with packed FP32 left on"
    for v in 0 13 7; do ${P}_withdrawn_pk 600 1 $v 2>&1 | filt; done
    echo "== the shipped precision-2 front end and the other co-running kernels as aggressors, victims with packed FP32 on and off"
    for a in 3 4 8 16; do ${P}_current 300 $a 0 2>&1 | filt; done
  } > "$O/interference_no_packed_f32.log" 2>&1
  cat "$O/interference_no_packed_f32.log"
  ;;
newtests)
  timeout 900 python -m pytest tests/test_gpu_interference.py tests/test_gpu_forward.py tests/test_gpu_ort_mixed.py -m gpu -x -q -s \
    -k "interference or undisturbed or variants_reproduce or register_fft or matrix_pipe or frontend_integer or logmel_frontend or fused_subsampling" > "$O/newtests.log" 2>&1
  tail -n 15 "$O/newtests.log" | filt
  ;;
ab_frontend)
  for v in 0 1; do bench1 headline_logmel$v QVERSE_LOGMEL=$v -- --steps 40; done
  for v in 0 1; do bench1 contexts1_logmel$v QVERSE_LOGMEL=$v -- --steps 40 --contexts 1; done
  for v in 0 1; do bench1 b256_ort_sub$v QVERSE_ORT_SUB=$v -- --batch 256 --precision ort --steps 12; done
  ;;
ab_nopk)   # the library built without packed-FP32 instructions (offline-tarteel_amd/libqverse_nopk.so) against the default build
  for lib in libqverse.so libqverse_nopk.so; do
    n=${lib%.so}
    bench1 headline_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --steps 60
    bench1 contexts1_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --steps 60 --contexts 1
    bench1 b256_fp16_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --batch 256 --steps 16
    bench1 b256_ort_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --batch 256 --precision ort --steps 12
    bench1 tta30_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --workload tta30 --steps 6 --warmup 2
  done
  ;;
skip_ln)   # dev-hook library (offline-tarteel_amd/libqverse_dev.so, -DQV_DEV_HOOKS): what the stand-alone LayerNorm launches cost, i.e. the most a fusion could win
  for rep in 1 2; do
    for cfg in "b64_ctx4:--steps 60" "b64_ctx1:--steps 60 --contexts 1" "b256_ctx4:--batch 256 --steps 16" "b256_ctx1:--batch 256 --steps 16 --contexts 1"; do
      n=${cfg%%:*}; a=${cfg#*:}
      bench1 ${n}_base_$rep QVERSE_LIB=$R/offline-tarteel_amd/libqverse_dev.so -- $a
      bench1 ${n}_skipln_$rep QVERSE_LIB=$R/offline-tarteel_amd/libqverse_dev.so QVERSE_SKIP=1 -- $a
    done
  done
  ;;
spans)   # the prefix-shared span pass: tests that exercise match_verse, then the post-logits replay with both kernels
  timeout 900 python -m pytest tests/test_gpu_postlogits.py tests/test_gpu_tracker.py tests/test_gpu_tta.py -m gpu -x -q > "$O/spans_tests.log" 2>&1; tail -n 4 "$O/spans_tests.log"
  for v in 0 1; do QVERSE_SPANS=$v timeout 200 python tools/post_bench.py > "$O/post_bench_spans$v.jsonl" 2>/dev/null; echo "QVERSE_SPANS=$v"; cut -c1-110 "$O/post_bench_spans$v.jsonl"; done
  cd /tmp && export TMPDIR=/tmp
  for v in 0 1; do QVERSE_SPANS=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profpost_spans$v" -o p -- python "$R/tools/post_bench.py" --steps 5 > /dev/null 2>&1; grep -i "k_spans" "$O/profpost_spans$v/p_kernel_stats.csv" | cut -c1-160; done
  find "$O" -name "*_kernel_trace.csv" -delete; cd "$R"
  ;;
ab_lib)   # A/B of a second library build: AB_LIB=<file under offline-tarteel_amd/> tools/gpu_round.sh <tag> ab_lib
  for rep in 1 2 3; do for lib in libqverse.so ${AB_LIB:?}; do
    n=${lib%.so}_$rep
    bench1 contexts1_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --steps 60 --contexts 1
    bench1 headline_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --steps 60
    bench1 b256_fp16_$n QVERSE_LIB=$R/offline-tarteel_amd/$lib -- --batch 256 --steps 16
  done; done
  ;;
fwd_graph)
  timeout 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "graph_replay" 2>&1 | tail -15
  for rep in 1 2 3; do for g in 0 1; do bench1 headline_fwdgraph${g}_$rep QVERSE_FWD_GRAPH=$g -- --steps 60; done; done
  bench1 ctx2_fwdgraph0 QVERSE_FWD_GRAPH=0 -- --steps 60 --contexts 2
  bench1 ctx2_fwdgraph1 QVERSE_FWD_GRAPH=1 -- --steps 60 --contexts 2
  bench1 ort_fwdgraph0 QVERSE_FWD_GRAPH=0 -- --steps 40 --precision ort
  bench1 ort_fwdgraph1 QVERSE_FWD_GRAPH=1 -- --steps 40 --precision ort
  ;;
ctx_graph)
  timeout 600 python -m pytest tests/test_gpu_forward.py -x -q -m gpu -k "graph_replay" 2>&1 | tail -3
  for rep in 1 2 3; do for g in 0 1; do bench1 driver_cmd_fwdgraph${g}_$rep QVERSE_FWD_GRAPH=$g -- --gpus 1 --steps 20 --warmup 5; done; done
  grep -o '"forward_graph": {[^}]*}' "$O"/bench_driver_cmd_fwdgraph*_1.json
  ;;
post_graph)
  for rep in 1 2 3; do for g in 0 1; do bench1 headline_postgraph${g}_$rep QVERSE_POST_GRAPH=$g -- --steps 60; done; done
  ;;
ctx_sweep)
  for c in 2 3 4 5 6 8; do bench1 b64_contexts$c X=0 -- --steps 60 --contexts $c; done
  for c in 2 4 6 8; do bench1 b256_contexts$c X=0 -- --batch 256 --steps 16 --contexts $c; done
  ;;
hwq_sweep)
  for q in 8 12 16; do for c in 4 6 8; do bench1 b64_hwq${q}_contexts$c GPU_MAX_HW_QUEUES=$q -- --steps 60 --contexts $c; done; done
  ;;
soak)
  for p in 0 1 2; do timeout 400 python tools/soak.py --batches ${SOAK_BATCHES:-3000} --seed $((11 + p)) --precision $p --third 2>&1 | filt; done > "$O/soak_three_precisions.log" 2>&1
  timeout 300 python tools/dev_ort_race.py --batches 400 2>&1 | filt | tail -3 >> "$O/soak_three_precisions.log"
  cat "$O/soak_three_precisions.log"
  ;;
soak_graph)   # keys repeat (7 recipes, 4 graph slots per context): the forward-graph replay / eviction path against a one-context engine
  for p in ${SOAK_PRECISIONS:-0 2}; do timeout 200 python tools/soak.py --batches ${SOAK_BATCHES:-400} --seed $((21 + p + ${SOAK_BATCHES:-400})) --precision $p --recipes 7 2>&1 | filt; done > "$O/soak_graph_replay_${SOAK_BATCHES:-400}.log" 2>&1
  cat "$O/soak_graph_replay_${SOAK_BATCHES:-400}.log"
  ;;
suite)
  timeout 1500 python -m pytest tests -m gpu -x -q > "$O/gpu_tests.log" 2>&1; tail -n 3 "$O/gpu_tests.log"
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -n 1 "$O/smoke.log"
  ;;
sweep)
  timeout 300 python tools/sweep.py --out "$O/sweep.json" > "$O/sweep.log" 2>&1; cut -c1-200 "$O/sweep.log" | tail -n 7
  ;;
final)
  bash tools/final_round.sh "$TAG"
  ;;
*) echo "unknown stage $stage";;
esac
done
ls "$O" | head -60
