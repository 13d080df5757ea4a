#!/usr/bin/env python3
"""GPU box: decimating FirFilter<Complex> — the decimate-first (polyphase) tiles against the block's other kernels
(direct form / tiles with a decimating store / pruned or half-size inverse), ms per 1e8 input samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr

n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * (n // 2 + 1024), device="cuda")
s = torch.cuda.current_stream().cuda_stream


def t(blk, d):
    cap = n // d + 8
    for _ in range(2):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 5


rng = np.random.default_rng(1)
print("taps deci   poly    other   (ms per 1e8 samples)")
for d in (2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
    for L in (31, 63, 127, 255, 401, 1000, 2467):
        taps = (rng.standard_normal(L) / L).astype(np.complex64)
        try:
            with rr.build_options(fir_poly=1):
                p = t(rr.FirFilter(taps, deci=d), d)
        except Exception as e:
            p = float("nan")
        with rr.build_options(fir_poly=-1):
            o = t(rr.FirFilter(taps, deci=d), d)
        print(f"{L:5d} {d:3d}  {p:7.3f}  {o:7.3f}  {'POLY' if p < o else ''}")
