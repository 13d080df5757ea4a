#!/usr/bin/env python3
"""GPU box: FftFilter kernel time per sample vs batch size (cache-resident vs HBM-resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
taps = rr.low_pass_complex(10e6, 1e6, 60e3)
for n in (2_000_000, 8_000_000, 32_000_000, 100_000_000):
    x = (torch.rand(2 * n, device="cuda") * 2 - 1)
    y = torch.empty(2 * (n + 2048), device="cuda")
    b = rr.FftFilter(taps)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 2048, s)
    torch.cuda.synchronize()
    b.set_profiling(True)
    reps = max(5, 400_000_000 // n)
    for _ in range(reps):
        b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 2048, s)
    torch.cuda.synchronize()
    ms, k = b.profile()
    print(f"n={n:>11,d}  {ms/k*1e3:9.1f} us/launch  {16*n/(ms/k*1e-3)/1e12:.2f} TB/s alg")
