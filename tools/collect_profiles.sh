#!/bin/bash
# Copy the evidence set tools/final_round.sh left under gpurun_out/ into profiles/ under the round's tag, with the file
# names of the earlier rounds (profiles/README.md).      usage: tools/collect_profiles.sh r05_m
set -u
T=${1:?tag}
R=$(cd "$(dirname "$0")/.." && pwd)
F=$R/gpurun_out/final; P=$R/profiles
cp_() { [ -f "$1" ] && cp "$1" "$P/${T}_$2"; }
cp_ $F/tests.log gpu_tests.log; cp_ $F/smoke.log smoke.log; cp_ $F/bench.json bench_default.json; cp_ $F/bench_wall.txt bench_default_wall.txt
for n in contexts1 b256_fp16 cfg2_b256_mixed cfg2_b256_ort b64_ort tta30 tta30_ort capacity30s under_rocprof; do cp_ $F/bench_$n.json bench_$n.json; done
cp_ $F/sweep.json sweep.json; cp_ $F/post_bench.jsonl post_bench.jsonl; cp_ $F/tracker_bench.jsonl tracker_bench.jsonl
cp_ $F/ort_semantics_delta.json ort_semantics_delta.json; cp_ $F/att_bench.log att_bench.log
cp_ $F/prof3/p_kernel_stats.csv headline_kernel_stats.csv; cp_ $F/prof1/p_kernel_stats.csv contexts1_kernel_stats.csv
cp_ $F/profpost/p_kernel_stats.csv postlogits_verse_shaped_kernel_stats.csv; cp_ $F/prof_tta30/p_kernel_stats.csv tta30_contexts1_kernel_stats.csv
cp_ $F/prof_b64_ort/p_kernel_stats.csv b64_ort_contexts1_kernel_stats.csv
for p in fp16 mixed ort; do cp_ $F/prof_b256_$p/p_kernel_stats.csv b256_${p}_contexts1_kernel_stats.csv; done
cp_ $R/gpurun_out/pmc_$T/traffic.json pmc_traffic.json; cp_ $R/gpurun_out/pmc_$T/mfma.json mfma_busy.json
cp_ $F/mfma_in_situ_b64.json mfma_in_situ_b64.json; cp_ $F/mfma_in_situ_b256.json mfma_in_situ_b256.json
cp_ $F/post_bench_376.jsonl post_bench_376.jsonl; cp_ $F/bench_strong2048.json bench_strong2048.json; cp_ $F/bench_strong2048_contiguous.json bench_strong2048_contiguous.json
cp_ $F/ort_floor_table.json ort_floor_table.json
ls $P | grep "^${T}_" | wc -l
