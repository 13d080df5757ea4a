#!/usr/bin/env python3
"""GPU box: FirFilter<Complex> (default path selection) over decimations 1..20 x tap counts, ms per 1e8 input samples,
next to the decimate-first tiles forced where they exist (fir_poly=1).  Looks for cliffs in the path selection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * (n + 1024), device="cuda")
s = torch.cuda.current_stream().cuda_stream
def t(blk, d):
    cap = n // d + 8
    for _ in range(2):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 3
rng = np.random.default_rng(1)
for L in (31, 127, 401, 1000, 2467, 5000):
    taps = (rng.standard_normal(L) / L).astype(np.complex64)
    row = []
    for d in list(range(1, 17)) + [20, 32]:
        o = t(rr.FirFilter(taps, deci=d), d)
        p = None
        if 2 <= d <= 16:
            try:
                with rr.build_options(fir_poly=1):
                    p = t(rr.FirFilter(taps, deci=d), d)
            except Exception:
                p = None
        row.append(f"/{d}={o:.3f}" + (f"({p:.3f})" if p is not None else ""))
    print(f"L={L:5d} " + " ".join(row), flush=True)
