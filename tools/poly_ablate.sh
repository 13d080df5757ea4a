#!/bin/bash
# GPU box: compile-time ablation of k_fm_chain_poly / k_fm_multi_poly (wrong results, right instruction mix).  Libraries:
#   for b in 1 2 4 8 16; do make -C rustradio_amd/csrc EXTRA=-DRR_POLY_ABLATE=$b OUT=../lib_x$b; done
# bits: 1 no input loads, 2 no atan2, 4 no output stores, 8 no H loads, 16 no LDS exchanges inside the transforms.
# Selected through RR_LIB_PATH; the product library is never overwritten.  Usage: bash tools/poly_ablate.sh "1 2 4 8 16" "fm_chain fm_multi"
one() { RR_LIB_PATH=$1 python bench.py --workload $2 --no-others --no-cpu --no-dropin --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; }
for w in ${2:-fm_chain}; do
  echo -n "$w  bits=0: "; one "" $w; one "" $w; echo
  for b in $1; do echo -n "$w  bits=$b: "; one $PWD/rustradio_amd/lib_x$b/librustradio_amd.so $w; one $PWD/rustradio_amd/lib_x$b/librustradio_amd.so $w; echo; done
  echo -n "$w  bits=0: "; one "" $w; echo
done
