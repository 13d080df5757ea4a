#!/usr/bin/env python3
"""GPU box, timing library (make -C rustradio_amd/csrc TIMING=1 OUT=../lib_timing; run with
RR_LIB_PATH=rustradio_amd/lib_timing/librustradio_amd.so): phase durations (s_memtime ticks, 100 MHz) of one tile of
k_fm_chain_poly, both waves of workgroup 0, third tile."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustradio_amd as rr
n = 96_000_000       # (workgroup 0 must reach its third tile under the 3x oversubscribed grid)
taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(n // 6 + 1024, device="cuda")
b = rr.FmChain(taps, 1, 6, 1.0)
for _ in range(3):
    b.work_dev(x.data_ptr(), n, y.data_ptr(), n // 6 + 1024, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
k = rr.lib().rr_debug_fft_stamps(st)
s = list(st)
names = ["loads + forward + MAC", "barrier 1", "sum + inverse (wave 0)", "barrier 2", "demodulation"]
for w in range(2):
    print(f"wave {w}: tile total {s[16*w+5]-s[16*w]} ticks (x10 ns)")
    for i, nm in enumerate(names):
        print(f"   {nm:26s} {s[16*w+i+1]-s[16*w+i]:7d}")
