#!/bin/bash
# GPU box: kernel time of workloads across several variant libraries (rustradio_amd/lib_y_<VARIANT>, built with
#   make -C rustradio_amd/csrc EXTRA=-DRR_POLY_<VARIANT as NAME=VALUE> OUT=../lib_y_<NAME>_<VALUE>), product first and last.
one() { RR_LIB_PATH=$1 python bench.py --workload $2 --no-others --no-cpu --no-dropin --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; }
for w in fm_chain rtl_fm_chain; do
  for lib in "" lib_y_OVERSUB_1 lib_y_OVERSUB_2 lib_y_OVERSUB_4 lib_y_WIDE_0 ""; do
    echo -n "$w ${lib:-product}: "; p=""; [ -n "$lib" ] && p=$PWD/rustradio_amd/$lib/librustradio_amd.so; one "$p" $w; one "$p" $w; echo
  done
done
