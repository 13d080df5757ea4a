#!/usr/bin/env python3
"""GPU box: fused FM chain kernel time vs taps (2.4e7 samples)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 24_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(n // 6 + 4096, device="cuda")
rng = np.random.default_rng(0)
for L in (463, 1025, 2467, 3300, 5000):
    taps = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) / L).astype(np.complex64)
    b = rr.FmChain(taps, 1, 6)
    for _ in range(2):
        rr.FmChain(taps, 1, 6).work_dev(x.data_ptr(), n, y.data_ptr(), n // 6 + 4096)
    torch.cuda.synchronize()
    b.set_profiling(True)
    b.work_dev(x.data_ptr(), n, y.data_ptr(), n // 6 + 4096)
    torch.cuda.synchronize()
    ms, k = b.profile()
    print(f"taps {L}: {ms/k:.4f} ms per 2.4e7 samples ({n/(ms/k*1e-3)/1e9:.0f} Gsamples/s)")
