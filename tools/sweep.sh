#!/bin/bash
# GPU box: sweep FftFilter tuning knobs.  Usage: bash tools/sweep.sh "<log2f list>" "<var list>"
for f in $1; do for v in $2; do
  echo -n "F=2^$f VAR=$v: "
  RR_FFT_LOG2F=$f RR_FFT_VAR=$v python bench.py --steps 20 --warmup 3 --no-cpu --no-others 2>&1 | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d[\"roofline\"]; print(d[\"value\"], r[\"avg_kernel_ms\"], r[\"achieved\"])"
done; done
