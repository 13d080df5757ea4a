#!/usr/bin/env python3
"""GPU box: default path selection of FirFilter<Float>, Hilbert->FirFilter (HilbertFir), FftFilter, FftFilterFloat and
AudioChain over their shape parameters; ms per 1e8 input samples.  Looks for cliffs (a neighbouring shape several times slower)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
s = torch.cuda.current_stream().cuda_stream
xf = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * n + 65536, device="cuda")
rng = np.random.default_rng(1)
def t(blk, nin, cap, reps=3):
    for _ in range(2): blk.work_dev(xf.data_ptr(), nin, y.data_ptr(), cap, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): blk.work_dev(xf.data_ptr(), nin, y.data_ptr(), cap, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
which = sys.argv[1:] or ["firf", "hilfir", "fft", "fftf", "audio"]
if "firf" in which:
    for L in (31, 127, 401, 1000, 2467, 3584, 5000):
        taps = (rng.standard_normal(L) / L).astype(np.float32)
        print(f"FirFilter<Float> L={L:5d} " + " ".join(f"/{d}={t(rr.FirFilter(taps, deci=d), n, n // d + 8):.3f}" for d in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 20, 32)), flush=True)
if "hilfir" in which:
    for L in (31, 127, 255, 1000, 2467):
        taps = (rng.standard_normal(L) / L).astype(np.complex64)
        print(f"HilbertFir(65) L={L:5d} " + " ".join(f"/{d}={t(rr.HilbertFir(65, taps, d), n, n // d + 8):.3f}" for d in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20, 32)), flush=True)
if "fft" in which:
    row = []
    for L in (3, 31, 127, 401, 511, 513, 1000, 1500, 2047, 2049, 2467, 3000, 4000, 5000, 8191, 8193, 12000, 16383, 16385, 20000):
        taps = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) / L).astype(np.complex64)
        row.append(f"{L}={t(rr.FftFilter(taps), n // 2, n // 2 + 65536):.3f}")
    print("FftFilter (5e7 Complex samples) " + " ".join(row), flush=True)
if "fftf" in which:
    row = []
    for L in (3, 31, 127, 401, 1000, 2467, 3584, 3585, 5000, 9000):
        taps = (rng.standard_normal(L) / L).astype(np.float32)
        row.append(f"{L}={t(rr.FftFilterFloat(taps), n, n + 65536):.3f}")
    print("FftFilterFloat (1e8 f32) " + " ".join(row), flush=True)
if "audio" in which:
    for L in (127, 963, 2467, 3584):
        taps = (rng.standard_normal(L) / L).astype(np.float32)
        row = []
        for I, D in ((1, 1), (1, 2), (1, 4), (6, 25), (1, 5), (1, 8), (3, 2), (48, 200), (1, 16)):
            row.append(f"{I}:{D}={t(rr.AudioChain(taps, I, D, 0.5), n, n * I // D + 65536):.3f}")
        print(f"AudioChain L={L:5d} " + " ".join(row), flush=True)
