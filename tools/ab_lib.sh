#!/bin/bash
# GPU box: A/B a product build against rustradio_amd/lib_prev on one bench workload.  Usage: bash tools/ab_lib.sh <workload> [reps]
W=${1:-fm_multi}; N=${2:-4}
one() { python bench.py --workload $W --no-others --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; }
cp rustradio_amd/lib/librustradio_amd.so /tmp/rr_new.so
for i in $(seq $N); do
  cp /tmp/rr_new.so rustradio_amd/lib/librustradio_amd.so; echo -n "new "; one
  cp rustradio_amd/lib_prev/librustradio_amd.so rustradio_amd/lib/librustradio_amd.so; echo -n " prev "; one; echo
done
cp /tmp/rr_new.so rustradio_amd/lib/librustradio_amd.so
