set -u
O=gpurun_out/r06_k; mkdir -p $O
bash tools/pmc_round.sh r06_z 8064 > $O/pmc.log 2>&1; tail -5 $O/pmc.log | cut -c1-300
mkdir -p profiles; cp gpurun_out/pmc_r06_z/traffic.json profiles/r06_z_pmc_traffic.json; cp gpurun_out/pmc_r06_z/mfma.json profiles/r06_z_mfma_busy.json
s0=$SECONDS; timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "default bench.py wall $((SECONDS - s0)) s" | tee $O/bench_wall.txt; cut -c1-170 $O/bench.json
cp profiles/r06_z_pmc_traffic.json profiles/r06_z_mfma_busy.json $O/
