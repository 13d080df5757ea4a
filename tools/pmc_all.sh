#!/bin/bash
# GPU box: SQ counter passes (separate --pmc runs) for the dominant kernel of each workload -> gpurun_out/pmc_all/<w>.txt
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
P3="SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD"
P4="GRBM_GUI_ACTIVE SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR"
mkdir -p gpurun_out/pmc_all
for w in fftfilter fm_chain fm_multi channelizer channelizer_unfused; do
  OUT=gpurun_out/pmc_all/$w; mkdir -p $OUT; i=0
  for c in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu > "$OUT/pass$i.log" 2>&1
  done
  { echo "# bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu ; rocprofv3 --pmc, 4 separate passes, per-launch averages"; python3 tools/pmc_summary.py "$OUT" rr::k_ | grep -v "k_vcopy" ; } > gpurun_out/pmc_all/$w.txt
  rm -rf $OUT
done
python3 tools/fir_probe.py > /dev/null 2>&1
OUT=gpurun_out/pmc_all/fir_d1; mkdir -p $OUT; i=0
for c in "$P1" "$P2" "$P3" "$P4"; do i=$((i+1)); rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o p -- python3 tools/fir_probe.py > "$OUT/pass$i.log" 2>&1; done
{ echo "# tools/fir_probe.py (127 real / complex taps, d = 1, 1e8 samples); rocprofv3 --pmc, 4 separate passes"; python3 tools/pmc_summary.py "$OUT" rr::k_fir; } > gpurun_out/pmc_all/fir_d1.txt
rm -rf $OUT
ls gpurun_out/pmc_all
