#!/bin/bash
# GPU box: FftFilter kernel time vs resident workgroups per CU (RR_FFT_PERCU measurement knob).
one() { python bench.py --no-others --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'])"; }
for pc in 2 3 4 8; do echo -n "percu $pc: "; RR_FFT_PERCU=$pc one; done
