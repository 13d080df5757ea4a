#!/bin/bash
# GPU box: dominant-kernel time of workloads across variant libraries (rustradio_amd/<dir>/librustradio_amd.so, loaded through
# RR_LIB_PATH: the product .so is never overwritten), product first and last, same box, alternating.
#   bash tools/ab_libs.sh "<workloads>" "<dir> <dir> ..." [reps=2]
N=${3:-2}
one() { RR_LIB_PATH=$1 python bench.py --workload $2 --no-others --no-cpu --no-dropin --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; }
for w in $1; do
  for lib in "" $2 ""; do
    echo -n "$w ${lib:-product}: "; p=""; [ -n "$lib" ] && p=$PWD/rustradio_amd/$lib/librustradio_amd.so
    for i in $(seq $N); do one "$p" $w; done; echo
  done
done
