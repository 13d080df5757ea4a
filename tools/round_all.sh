#!/bin/bash
# GPU box: everything profiles/<tag>_* is made of, on the tree's final sources (run through gpurun; ~25 min of box time).
#   bash tools/round_all.sh [tag]   then locally: python tools/round_collect.py <tag>; cp gpurun_out/pmc_stalls/<tag>_*.txt profiles/
TAG=${1:-r06}
bash tools/round_profiles.sh $TAG
for w in fftfilter fm_chain fm_multi channelizer rtl_fm_example fir_1e8 fir_float; do
  bash tools/pmc_stalls.sh $w ${TAG}_${w}_stall_counters --no-dropin > /dev/null 2>&1
  mv gpurun_out/pmc_stalls/${TAG}_${w}_stall_counters.txt gpurun_out/prof_$TAG/${w}_stall_counters.txt
  rm -rf gpurun_out/pmc_stalls/raw_${TAG}_${w}_stall_counters
done
cd "$GRAFT_REPO_ROOT"
python3 -m tests.parity_allowance > gpurun_out/prof_$TAG/parity_allowance.log 2>&1
cp gpurun_out/parity_allowance.json gpurun_out/prof_$TAG/parity_allowance.json
cp gpurun_out/parity_allowance.json profiles/parity_allowance.json      # (the default line below carries its summary)
timeout 120 ./tools/micro/pcie_inplace.bin > gpurun_out/prof_$TAG/pcie_inplace.txt 2>&1
(timeout 60 ./tools/micro/rotor_rate.bin; timeout 200 python3 tools/replay_rate.py 2>/dev/null | grep -v amdgpu) > gpurun_out/prof_$TAG/rotor_rate.txt 2>&1
python3 tools/chain_tile_probe.py 2>/dev/null | grep -v amdgpu > gpurun_out/prof_$TAG/rtl_fm_tiles.txt
python3 tools/clock_probe.py channelizer > gpurun_out/prof_$TAG/clocks.txt 2>&1
bash tools/ab_nonfinite_pass.sh 2>&1 | grep -v amdgpu > gpurun_out/prof_$TAG/ab_nonfinite_pass.txt
python3 bench.py --detail-out "gpurun_out/prof_$TAG/bench_detail.json" > "gpurun_out/prof_$TAG/bench_default.json" 2> "gpurun_out/prof_$TAG/bench_default.log"
tail -c 300 "gpurun_out/prof_$TAG/bench_default.log"; ls gpurun_out/prof_$TAG | head -60
