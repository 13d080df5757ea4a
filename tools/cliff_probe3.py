#!/usr/bin/env python3
"""GPU box: FirFilter<Complex> with COMPLEX-valued taps over decimation x taps (ms per 1e8 samples), FmMulti (32 channels)
per-call time against window size for long filters, FmChain interpolating ratios over taps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
def ct(L): return ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
def t(blk, x, nin, y, cap, reps=3):
    for _ in range(2): blk.work_dev(x.data_ptr(), nin, y.data_ptr(), cap, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): blk.work_dev(x.data_ptr(), nin, y.data_ptr(), cap, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
which = sys.argv[1:] or ["firc", "multiwin", "interp"]
if "firc" in which:
    n = 100_000_000
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    y = torch.empty(2 * n + 65536, device="cuda")
    for L in (15, 31, 63, 127, 255, 401, 1000):
        taps = ct(L)
        print(f"FirFilter complex taps L={L:5d} " + " ".join(f"/{d}={t(rr.FirFilter(taps, deci=d), x, n, y, n // d + 8):.3f}" for d in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 16, 20, 32, 64)), flush=True)
    del x, y
if "multiwin" in which:
    for L, D in ((463, 6), (3000, 6), (4500, 6), (2467, 10), (3800, 4)):
        taps = np.stack([ct(L) for _ in range(32)])
        row = []
        for n in (131_072, 512_000, 2_400_000, 9_600_000):
            x = torch.rand(2 * n, device="cuda") * 2 - 1
            cap = n // D + 8192
            y = torch.empty(32 * cap, device="cuda")
            row.append(f"{n}={t(rr.FmMulti(taps, 1, D), x, n, y, cap) * 1e3:.1f}us")
            del x, y
        print(f"FmMulti 32ch L={L} 1:{D}: " + " ".join(row), flush=True)
if "interp" in which:
    n = 12_000_000
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    for L in (127, 463, 2467, 5000):
        taps = ct(L)
        row = []
        for I, D in ((2, 1), (3, 1), (5, 1), (3, 2), (5, 3), (7, 5), (10, 3), (2, 5), (4, 6), (100, 33)):
            cap = n * I // D + 65536
            y = torch.empty(cap, device="cuda")
            row.append(f"{I}:{D}={t(rr.FmChain(taps, I, D), x, n, y, cap):.3f}")
            del y
        print(f"FmChain (1.2e7 samples) L={L}: " + " ".join(row), flush=True)
