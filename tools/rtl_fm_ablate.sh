#!/bin/bash
# GPU box: what the inverse side of the rational-ratio chain costs.  examples/rtl_fm.rs's front end
# (2467 taps, 25:128) on k_fm_chain_split<2>, measurement builds made beforehand with
#   for b in 512 1024 1536; do make -C rustradio_amd/csrc ABLATE=$b OUT=../lib_ab$b; done
# 512 = no inverse transforms / output butterfly (an upper bound on what ANY pruning of the inverse to the 25 of 128 kept
# positions could save), 1024 = no demodulation, 1536 = neither.  Wrong results, production register allocation.
one() { RR_LIB_PATH=$1 python bench.py --workload rtl_fm_example --no-others --no-cpu --no-dropin --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['roofline']['avg_kernel_ms'], end=' ')"; }
echo -n "full kernel:        "; for i in 1 2 3; do one ""; done; echo
for b in 512 1024 1536; do
  echo -n "ablate bits $b:   "; for i in 1 2 3; do one $PWD/rustradio_amd/lib_ab$b/librustradio_amd.so; done; echo
done
