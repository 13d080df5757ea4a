#!/bin/bash
# GPU box: d = 1 FIR inner-loop attribution with measurement builds (wrong results, right instruction mix):
#   for x in 1 2 3; do make -C rustradio_amd/csrc EXTRA=... OUT=../lib_x$x; done   (see tools/fir_exp.sh history)
cp rustradio_amd/lib/librustradio_amd.so /tmp/rr_keep.so
echo -n "product:        "; python tools/fir_probe.py | head -1
for x in 1 2 3; do cp rustradio_amd/lib_x$x/librustradio_amd.so rustradio_amd/lib/; echo -n "variant $x: "; python tools/fir_probe.py | head -1; done
cp /tmp/rr_keep.so rustradio_amd/lib/librustradio_amd.so
