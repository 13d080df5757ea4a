#!/usr/bin/env python3
"""GPU box: FftFilter kernel rate vs number of taps (1e8 samples per launch): which internal tile each tap count
gets and what it costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * (n + 65536), device="cuda")
rng = np.random.default_rng(0)
print("| taps | reference fft_size / nsamples | GPU tile | ms per 1e8 samples | TB/s (16 B/sample) |")
print("|---|---|---|---|---|")
for L in (5, 33, 127, 255, 401, 463, 512, 1000, 1025, 1500, 2000, 2467, 4096, 8191):
    taps = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) / L).astype(np.complex64)
    b = rr.FftFilter(taps)
    fs, ns, gf = rr.fftfilter_dims(b)
    for _ in range(2):
        b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 65536)
    torch.cuda.synchronize()
    b.set_profiling(True)
    for _ in range(5):
        b2 = b
        st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 65536)
    torch.cuda.synchronize()
    ms, k = b.profile()
    print(f"| {L} | {fs} / {ns} | {gf} | {ms/k:.3f} | {16*n/(ms/k*1e-3)/1e12:.2f} |")
