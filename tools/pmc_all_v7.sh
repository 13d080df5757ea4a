#!/bin/bash
# GPU box: SQ counter passes (separate --pmc runs, never combined with trace domains) for the dominant kernel of each
# workload with the current kernels -> gpurun_out/pmc_v7/<w>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
P4="GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
mkdir -p gpurun_out/pmc_v7
for w in ${WORKLOADS:-fftfilter fm_chain fm_multi channelizer fir_1e8 fir_float}; do
  OUT=gpurun_out/pmc_v7/raw_$w; mkdir -p "$OUT"; i=0
  for c in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu > "$OUT/pass$i.log" 2>&1
  done
  { echo "# bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu ; rocprofv3 --pmc, 4 separate passes, per-launch averages"; python3 tools/pmc_summary.py "$OUT" rr::k_ | grep -v -A16 "k_vcopy" | head -40; } > gpurun_out/pmc_v7/$w.txt
done
cat gpurun_out/pmc_v7/*.txt | grep -c launches
