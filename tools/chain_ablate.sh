#!/bin/bash
# GPU box: steady cost per tile of the single-chain decimate-first kernel (FmChain 463 taps 1:6, 9.6e7 samples per call)
# with parts removed at compile time — libraries built beforehand:
#   for b in 1 2 4 8 16 9 25; do make -C rustradio_amd/csrc EXTRA=-DRR_POLY_ABLATE=$b OUT=../lib_pa$b; done
# bits: 1 no input loads, 2 no atan2, 4 no output stores, 8 no response loads, 16 no LDS exchanges inside the forward transforms
for b in ${ABL:-"" 1 2 4 8 16 9 25 ""}; do
  p=""; [ -n "$b" ] && p=$PWD/rustradio_amd/lib_pa$b/librustradio_amd.so
  echo -n "ablate ${b:-none}: "
  RR_LIB_PATH=$p python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
out = []
for n in (24_000_000, 96_000_000):
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    cap = n // 6 + 1024
    y = torch.empty(cap, device="cuda")
    b = rr.FmChain(taps, 1, 6, 1.0)
    for _ in range(30): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / 30 * 1e3
    out.append(f"n={n // 1000000}M {us:7.1f} us = {us / (n / 5676) * 1e3:6.2f} ns/tile")
print("   ".join(out))
PY
done
