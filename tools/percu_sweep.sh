#!/bin/bash
# GPU box, measurement library (make -C rustradio_amd/csrc EXTRA=-DRR_MEASURE_KNOBS OUT=../lib_m): workgroups launched per CU
# (RR_FFT_PERCU) against kernel time — more workgroups than resident slots = finer tile granularity through the hardware
# dispatcher.  Usage: bash tools/percu_sweep.sh "fm_chain fftfilter" "2 3 4 6 8 12 16 32"
for w in $1; do
  for n in 0 $2; do
    echo -n "$w percu=$n: "
    for r in 1 2; do RR_FFT_PERCU=$n RR_LIB_PATH=$PWD/rustradio_amd/lib_m/librustradio_amd.so python bench.py --workload $w --no-others --no-cpu --no-dropin --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; done; echo
  done
done
