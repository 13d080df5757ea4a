#!/bin/bash
# GPU box: kernel time of workloads, with optional --opt overrides.  Usage: bash tools/r2_ab.sh "fm_chain fm_multi" ["--opt fm_poly=-1"]
for w in $1; do
  python bench.py --workload $w --no-others --no-cpu --steps 20 $2 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $2', 'ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'kernel', r['avg_kernel_ms'], r.get('achieved'), r.get('frac'), r.get('hbm_frac'))"
done
