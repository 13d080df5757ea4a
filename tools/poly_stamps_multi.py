#!/usr/bin/env python3
"""GPU box, timing library (RR_LIB_PATH=rustradio_amd/lib_timing/librustradio_amd.so): phase durations (shader clocks)
of one tile of k_fm_multi_poly (32 channels), waves 0 and 1 of workgroup 0, second tile."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
from rustradio_amd import multi
n = 9_600_000        # (4 steps worth: workgroup 0 must reach its second tile under the 3x oversubscribed grid)
taps = multi.cfg4_taps(rr.low_pass_complex(2.4e6, 100e3, 12.5e3), range(32))
x = torch.rand(2 * n, device="cuda") * 2 - 1
cap = n // 6 + 1024
y = torch.empty(32 * cap, device="cuda")
b = rr.FmMulti(taps, 1, 6, 1.0)
for _ in range(3):
    b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
rr.lib().rr_debug_fft_stamps(st)
s = list(st)
names = ["load + forward + park", "barrier", "channel 0: MAC", "channel 0: inverse", "channel 0: demodulation", "channels 1..3", "barrier"]
for w in range(2):
    print(f"wave {w}: tile total {s[16*w+7]-s[16*w]} clocks")
    for i, nm in enumerate(names):
        print(f"   {nm:26s} {s[16*w+i+1]-s[16*w+i]:7d}")
