#!/bin/bash
# GPU box: compile-time phase ablation of the FftFilter kernel.  Libraries are built beforehand with
#   for b in 1 2 3 16 32 48; do make -C rustradio_amd/csrc ABLATE=$b OUT=../lib_ab$b; done
# and selected through RR_LIB_PATH (rustradio_amd/_lib.py): the product library is never overwritten.
# Usage: bash tools/ablate.sh "1 2 3 16 32 48" [log2f]
run() { python bench.py --steps 20 --warmup 3 --no-cpu --no-others --no-dropin --opt fft_log2f=${LOG2F:-11} 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(r[\"avg_kernel_ms\"], r[\"achieved\"])"; }
LOG2F=${2:-11}
echo -n "bits=0: "; run
for a in $1; do
  echo -n "bits=$a: "; RR_LIB_PATH=$PWD/rustradio_amd/lib_ab$a/librustradio_amd.so run
done
