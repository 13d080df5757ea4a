#!/bin/bash
# GPU box: same-box A/B of two builds (the product vs rustradio_amd/<dir>, loaded through RR_LIB_PATH — the product .so is
# never overwritten).  Usage: bash tools/ab.sh "<workloads>" [dir=lib_prev] [reps=3]
DIR=${2:-lib_prev}; N=${3:-3}
one() { RR_LIB_PATH=$1 python bench.py --workload $2 --no-others --no-cpu --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; }
for w in $1; do
  echo -n "$w  new: "; for i in $(seq $N); do one "" $w; done
  echo -n " | $DIR: "; for i in $(seq $N); do one rustradio_amd/$DIR/librustradio_amd.so $w; done
  echo -n " | new: "; for i in $(seq $N); do one "" $w; done; echo
done
