#!/usr/bin/env python3
"""GPU box: RationalResampler kernel rate for several ratios (Complex, 5e7 input samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 50_000_000
x = torch.rand(2 * n, device="cuda")
for I, D in ((1, 6), (1, 1), (3, 2), (25, 128), (6, 25), (7, 3), (2, 1)):
    cap = n * I // D + 16
    y = torch.empty(2 * cap, device="cuda")
    b = rr.RationalResampler(I, D, np.complex64)
    for _ in range(2):
        b2 = rr.RationalResampler(I, D, np.complex64); b2.work_dev(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    b.set_profiling(True)
    st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    ms, k = b.profile()
    print(f"{I}:{D}  {ms/k:.4f} ms  in {c} out {p}  {(8*c + 8*p)/(ms/k*1e-3)/1e12:.2f} TB/s alg")
