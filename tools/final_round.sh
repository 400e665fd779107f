#!/bin/bash
# Round evidence in one GPU-box call: GPU tests, smoke, the bench lines (headline, one batch at a time, configs[2],
# configs[4] workload), rocprofv3 kernel stats of the same bench command, the clip-length sweep, the post-logits /
# tracker micro-benchmarks, the quantisation / resampler distance tools and the PMC passes of the GEMM shapes.
# Everything lands in gpurun_out/final/ (copy what is to be judged into profiles/).   usage: tools/final_round.sh [tag]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final
mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests -m gpu -x -q > "$O/tests.log" 2>&1; tail -n 2 "$O/tests.log"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > "$O/smoke.log" 2>&1; tail -n 1 "$O/smoke.log"
s0=$SECONDS; timeout 600 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "default bench.py wall $((SECONDS - s0)) s" | tee "$O/bench_wall.txt"; cut -c1-170 "$O/bench.json"
timeout 300 python bench.py --contexts 1 --no-cpu-baseline > "$O/bench_contexts1.json" 2>/dev/null; cut -c1-170 "$O/bench_contexts1.json"
timeout 300 python bench.py --batch 256 --precision mixed --steps 20 --no-cpu-baseline > "$O/bench_cfg2_b256_mixed.json" 2>/dev/null; cut -c1-170 "$O/bench_cfg2_b256_mixed.json"
timeout 300 python bench.py --batch 256 --steps 20 --no-cpu-baseline > "$O/bench_b256_fp16.json" 2>/dev/null; cut -c1-170 "$O/bench_b256_fp16.json"
timeout 300 python bench.py --precision ort --steps 30 --no-cpu-baseline --no-extra > "$O/bench_b64_ort.json" 2>/dev/null; cut -c1-170 "$O/bench_b64_ort.json"
timeout 300 python bench.py --batch 256 --precision ort --steps 12 --no-cpu-baseline --no-extra > "$O/bench_cfg2_b256_ort.json" 2>/dev/null; cut -c1-170 "$O/bench_cfg2_b256_ort.json"
timeout 300 python bench.py --workload tta30 --steps 6 --warmup 2 --no-cpu-baseline > "$O/bench_tta30.json" 2>/dev/null; cut -c1-170 "$O/bench_tta30.json"
timeout 300 python bench.py --workload tta30 --precision ort --steps 4 --warmup 2 --no-cpu-baseline --no-extra > "$O/bench_tta30_ort.json" 2>/dev/null; cut -c1-170 "$O/bench_tta30_ort.json"
timeout 300 python bench.py --capacity-seconds 30 --steps 40 --no-cpu-baseline --no-extra --no-post-logits > "$O/bench_capacity30s.json" 2>/dev/null; cut -c1-170 "$O/bench_capacity30s.json"
timeout 300 python tools/sweep.py --out "$O/sweep.json" > "$O/sweep.log" 2>&1; tail -n 3 "$O/sweep.log" | cut -c1-200
timeout 200 python tools/post_bench.py > "$O/post_bench.jsonl" 2>/dev/null; cut -c1-110 "$O/post_bench.jsonl"
timeout 200 python tools/post_bench.py --frames 376 > "$O/post_bench_376.jsonl" 2>/dev/null; cut -c1-110 "$O/post_bench_376.jsonl"
timeout 600 python bench.py --workload strong2048 --steps 3 --warmup 1 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_strong2048.json" 2>/dev/null; cut -c1-170 "$O/bench_strong2048.json"
timeout 600 python bench.py --workload strong2048 --deal contiguous --steps 3 --warmup 1 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_strong2048_contiguous.json" 2>/dev/null; cut -c1-170 "$O/bench_strong2048_contiguous.json"
timeout 900 python tools/ort_floor_table.py --out "$O/ort_floor_table.json" > "$O/ort_floor_table.log" 2>&1; tail -n 3 "$O/ort_floor_table.log" | cut -c1-200
timeout 200 python tools/tracker_bench.py --cpu-texts 4 > "$O/tracker_bench.jsonl" 2>/dev/null; tail -n 2 "$O/tracker_bench.jsonl"
timeout 300 python tools/ort_delta.py --seconds 10 --out "$O/ort_semantics_delta.json" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof3" -o p -- python "$R/bench.py" --steps 20 --no-cpu-baseline --no-post-logits --no-extra > "$O/bench_under_rocprof.json" 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof1" -o p -- python "$R/bench.py" --steps 20 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/profpost" -o p -- python "$R/tools/post_bench.py" --steps 5 > /dev/null 2>&1
for prec in fp16 mixed ort; do   # B = 256 (configs[2] / the per-rank slice of configs[3]), one batch at a time: which kernels the step is made of
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b256_$prec" -o p -- python "$R/bench.py" --precision $prec --batch 256 --steps 8 --warmup 2 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_tta30" -o p -- python "$R/bench.py" --workload tta30 --steps 3 --warmup 1 --contexts 1 --no-cpu-baseline --no-extra > /dev/null 2>&1
# in-situ matrix-pipe utilisation over ALL GEMM launches of the bench command itself (one batch at a time; B = 64 and B = 256)
for bb in 64 256; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/pmc_insitu_$bb" -o p -- python "$R/bench.py" --batch $bb --contexts 1 --steps 3 --warmup 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
  f=$(find "$O/pmc_insitu_$bb" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python "$R/tools/pmc_insitu.py" "$f" "$O/mfma_in_situ_b$bb.json" $((bb * 126)) fp16 | cut -c1-300
  rm -rf "$O/pmc_insitu_$bb"
done
timeout 120 "$R/tools/att_bench" 64 126 200 > "$O/att_bench.log" 2>&1; timeout 120 "$R/tools/att_bench" 256 126 100 >> "$O/att_bench.log" 2>&1; timeout 120 "$R/tools/att_bench" 64 376 50 >> "$O/att_bench.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b64_ort" -o p -- python "$R/bench.py" --precision ort --steps 16 --contexts 1 --no-cpu-baseline --no-post-logits --no-extra > /dev/null 2>&1
# the rejected fused feed-forward prototype next to the two GEMM kernels it would replace ("commit it with its rocprof table")
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_ffn_fused" -o p -- "$R/tools/ffn_fused_bench" 30 32256 1 > "$O/ffn_fused_bench_under_rocprof.log" 2>&1
cd "$R"
timeout 120 tools/ffn_fused_bench 30 8064 1 > "$O/ffn_fused_bench.log" 2>&1; timeout 120 tools/ffn_fused_bench 30 32256 1 >> "$O/ffn_fused_bench.log" 2>&1
find "$O" -name "*_kernel_trace.csv" -path "*prof*" -delete   # the traces are large; the stats are what is kept
bash tools/pmc_round.sh ${1:-final} 8064 > /dev/null 2>&1
ls "$O" | head -40
