#!/bin/bash
# GPU box: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of the dominant kernel of several workloads.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in ${WORKLOADS:-fftfilter fm_chain channelizer fm_multi fir_1e8 fir_float}; do
  OUT=gpurun_out/traffic_$w; mkdir -p $OUT
  i=0
  for c in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    rocprofv3 --pmc $c --output-format csv -d "$OUT/pass$i" -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu > "$OUT/pass$i.log" 2>&1
  done
  python3 tools/pmc_summary.py "$OUT" rr:: > "$OUT/summary.txt"
  cat "$OUT/summary.txt"
done
