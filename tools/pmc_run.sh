#!/bin/bash
# Collect rocprofv3 PMC counters for bench.py in separate passes (never combined with
# trace domains), on the GPU box:   bash tools/pmc_run.sh <outdir> [bench args...]
# Each pass runs:  rocprofv3 --pmc <counters> -- python3 bench.py ...
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:---steps 3 --warmup 1 --no-others --no-cpu}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $line --output-format csv -d "$OUT/pass$i" -o p -- python3 bench.py $ARGS > "$OUT/pass$i.log" 2>&1
done <<'PASSES'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAVES
FETCH_SIZE
WRITE_SIZE
GRBM_GUI_ACTIVE
TCC_HIT_sum TCC_MISS_sum
PASSES
python3 tools/pmc_summary.py "$OUT" | tee "$OUT/summary.txt"
