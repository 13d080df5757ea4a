#!/bin/bash
# GPU box: compile-time phase ablation of the FftFilter kernel.  Libraries are built beforehand with
#   for b in 1 2 3 16 32 48; do make -C rustradio_amd/csrc ABLATE=$b OUT=../lib_ab$b; done
# Usage: bash tools/ablate.sh "1 2 3 16 32 48" [log2f]
cp rustradio_amd/lib/librustradio_amd.so /tmp/rr_keep.so
run() { RR_FFT_LOG2F=${2:-11} python bench.py --steps 20 --warmup 3 --no-cpu --no-others 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(r[\"avg_kernel_ms\"], r[\"achieved\"])"; }
echo -n "bits=0: "; run
for a in $1; do
  cp rustradio_amd/lib_ab$a/librustradio_amd.so rustradio_amd/lib/librustradio_amd.so
  echo -n "bits=$a: "; run
done
cp /tmp/rr_keep.so rustradio_amd/lib/librustradio_amd.so
