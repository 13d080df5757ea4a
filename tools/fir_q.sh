#!/bin/bash
# GPU box: d = 1 FIR kernel time vs taps actually multiplied (RR_FIR_QCOMPUTE) and resident workgroups (RR_FIR_PERCU)
for q in 0 32 64 96 128; do echo -n "q $q: "; RR_FIR_QCOMPUTE=$q python tools/fir_probe.py | head -1; done
for pc in 1 2 3 4 8; do echo -n "percu $pc: "; RR_FIR_PERCU=$pc python tools/fir_probe.py | head -1; done
