#!/usr/bin/env python3
"""GPU box: fused FM chain, decimate-first tiles (k_fm_chain_poly) against the half-size / full-size inverse kernels, per
decimation and filter length.  ms per 2.4e7 input samples (kernel time from HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 24_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(n // 2 + 4096, device="cuda")
rng = np.random.default_rng(1)
def t_of(blk, D):
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), n // 2 + 4096, s)
    torch.cuda.synchronize()
    blk.set_profiling(True)
    for _ in range(8):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), n // 2 + 4096, s)
    torch.cuda.synchronize()
    ms, k = blk.profile()
    return ms / k
for L in ([int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else (127, 463, 1000, 2467)):
    taps = ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
    for D in ([int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else (2, 3, 4, 5, 6, 7, 8, 10, 12, 16)):
        try:
            with rr.build_options(fm_poly=1):          # forced wherever the kernel exists (the default applies FmChain's rule)
                a = t_of(rr.FmChain(taps, 1, D), D)
        except Exception as e:
            a = float("nan")
        with rr.build_options(fm_poly=-1):
            b = t_of(rr.FmChain(taps, 1, D), D)
        print(f"L={L:5d} D={D:2d}  poly {a:.4f} ms   other {b:.4f} ms   {'POLY' if a < b else 'other'}")
