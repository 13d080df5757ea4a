#!/bin/bash
# GPU box: phase ablation of the FftFilter kernel with the -DRR_FFT_ABLATE_BUILD library
# (make -C rustradio_amd/csrc ABLATE=1 OUT=../lib_ablate).  Usage: bash tools/ablate.sh "0 1 2 ..." [log2f]
cp rustradio_amd/lib/librustradio_amd.so /tmp/rr_keep.so
cp rustradio_amd/lib_ablate/librustradio_amd.so rustradio_amd/lib/librustradio_amd.so
for a in $1; do
  echo -n "ablate=$a: "
  RR_FFT_ABLATE=$a RR_FFT_LOG2F=${2:-11} python bench.py --steps 20 --warmup 3 --no-cpu --no-others 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(r[\"avg_kernel_ms\"], r[\"achieved\"])"
done
cp /tmp/rr_keep.so rustradio_amd/lib/librustradio_amd.so
