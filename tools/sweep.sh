#!/bin/bash
# GPU box: sweep the FftFilter tile size.  Usage: bash tools/sweep.sh "<log2f list>"
for f in $1; do
  echo -n "F=2^$f: "
  python bench.py --steps 20 --warmup 3 --no-cpu --no-others --no-dropin --opt fft_log2f=$f 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(d[\"value\"], r[\"avg_kernel_ms\"], r[\"achieved\"])"
done
