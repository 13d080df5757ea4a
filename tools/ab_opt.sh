#!/bin/bash
# Same-box A/B of the headline workload (and the host-window drop-in runs) with and without one rr_build_opts override:
#   bash tools/ab_opt.sh KEY=VALUE [rounds] [dropin]
opt=$1; rounds=${2:-3}; nodrop=--no-dropin; [ "$3" = dropin ] && nodrop=
for i in $(seq $rounds); do
  for o in "" "--opt $opt"; do
    python bench.py --steps 40 --warmup 6 --no-others --no-cpu $nodrop $o 2>/dev/null \
      | python -c "
import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('${o:-default}', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['kernel'])
for k,v in d.get('others',{}).items():
    if k.startswith('dropin') and isinstance(v,dict): print('   ',k,{a:b for a,b in v.items() if 'msamples' in a or 'frac' in a or a.endswith('_us')})
"
  done
done
