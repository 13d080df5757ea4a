#!/usr/bin/env python3
"""GPU box: decimating FirFilter<Complex> at 1e8 input samples, direct-form kernel vs overlap-save tiles with a
decimating store (k_fftfilt_deci), by (taps, deci): where FirC32's automatic choice should flip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * n, device="cuda")
rng = np.random.default_rng(1)
for L, d in ((32, 2), (64, 2), (127, 2), (255, 2), (401, 2), (1000, 2), (127, 6), (255, 6), (401, 6), (401, 10), (127, 3), (127, 4), (255, 4), (255, 8), (401, 16), (1000, 32), (2467, 32)):
    for cplx in (False, True):
        t = rng.uniform(-1, 1, L) + (1j * rng.uniform(-1, 1, L) if cplx else 0)
        t = (t / L).astype(np.complex64)
        row = []
        for opts in ({"fir_path": "direct"}, {"fir_path": "fft", "fir_prune": -1, "fir_half": -1, "fir_poly": -1},
                     {"fir_path": "fft", "fir_prune": -1, "fir_poly": -1}, {}):
            with rr.build_options(**opts):
                f = rr.FirFilter(t, deci=d)
            for _ in range(2):
                f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
            torch.cuda.synchronize()
            f.set_profiling(True)
            for _ in range(4):
                f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
            torch.cuda.synchronize()
            ms, k = f.profile()
            row.append(ms / k)
        print(f"L={L:5d} d={d:3d} {'complex' if cplx else 'real   '} taps: direct {row[0]:.4f}  deci-store {row[1]:.4f}  half {row[2]:.4f}  auto {row[3]:.4f} ms", flush=True)
