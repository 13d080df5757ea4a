#!/bin/bash
# GPU box: calibrate FETCH_SIZE / WRITE_SIZE against known byte counts per access shape (tools/micro/fetchcal.hip).
#   bash tools/fetchcal.sh [outdir]      -> <outdir>/fetch_calibration.json (copy to profiles/)
OUT=${1:-gpurun_out/fetchcal}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
[ -x tools/micro/fetchcal.bin ] || hipcc --offload-arch=gfx950 -O3 tools/micro/fetchcal.hip -o tools/micro/fetchcal.bin
tools/micro/fetchcal.bin > "$OUT/run.log" 2>&1
i=0
for c in FETCH_SIZE WRITE_SIZE; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o p -- tools/micro/fetchcal.bin > "$OUT/pass$i.log" 2>&1
done
python3 tools/fetchcal_summary.py "$OUT" | tee "$OUT/summary.txt"
