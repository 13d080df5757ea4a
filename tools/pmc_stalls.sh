#!/bin/bash
# (The TA_* counters abort rocprofv3 on this pool and hang it at finalisation for the rest of the call: left out, and
# every pass runs under `timeout`.)
# GPU box: where does the dominant kernel of a workload wait?  Issue / wait / memory-pipeline stall counters in separate
# --pmc passes (never combined with trace domains) -> gpurun_out/pmc_stalls/<workload>.txt
W=${1:-fftfilter}; TAG=${2:-$W}; shift; shift; EXTRA="$@"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_stalls/raw_$TAG; mkdir -p "$OUT"; i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $line --output-format csv -d "$OUT/pass$i" -o p -- python3 bench.py --workload $W --steps 3 --warmup 1 --no-others --no-cpu --no-verify $EXTRA > "$OUT/pass$i.log" 2>&1
done <<'PASSES'
SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH
TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL
GRBM_GUI_ACTIVE GRBM_TA_BUSY GRBM_TC_BUSY GRBM_EA_BUSY
TCC_BUSY TCC_CYCLE TCC_REQ TCC_EA0_RDREQ
SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_WAVES
FETCH_SIZE
WRITE_SIZE
PASSES
{ echo "# bench.py --workload $W --steps 3 --warmup 1 --no-others --no-cpu ; rocprofv3 --pmc, one pass per line of tools/pmc_stalls.sh, per-launch averages"; python3 tools/pmc_summary.py "$OUT" rr::k_ | awk '/k_vcopy/{skip=1} /^== rr::k_f/{skip=0} !skip{print}'; } > gpurun_out/pmc_stalls/$TAG.txt
cat gpurun_out/pmc_stalls/$TAG.txt; grep -il "error\|invalid" $OUT/pass*.log | head
