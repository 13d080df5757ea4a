#!/usr/bin/env python3
"""GPU box: non-decimating FirFilter<Complex> at 1e8 samples through the direct-form kernel (fir_path="direct")
and through the overlap-save FFT tiles (fir_path="fft"), real and Complex taps, by filter length: where the
crossover lies (FirC32's min_taps) and what the automatic choice costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * n, device="cuda")
rng = np.random.default_rng(1)
for L in (2, 4, 8, 12, 16, 24, 32, 64, 127, 255, 1000):
    for cplx in (False, True):
        t = rng.uniform(-1, 1, L) + (1j * rng.uniform(-1, 1, L) if cplx else 0)
        t = (t / L).astype(np.complex64)
        row = []
        for opts in ({"fir_path": "direct"}, {"fir_path": "fft"}, {}):
            with rr.build_options(**opts):
                f = rr.FirFilter(t)
            for _ in range(2):
                f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
            torch.cuda.synchronize()
            f.set_profiling(True)
            for _ in range(4):
                f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
            torch.cuda.synchronize()
            ms, k = f.profile()
            row.append(ms / k)
        print(f"L={L:5d} {'complex' if cplx else 'real   '} taps: direct {row[0]:.4f} ms  fft {row[1]:.4f} ms  auto {row[2]:.4f} ms", flush=True)
