#!/usr/bin/env python3
"""GPU box: wall time per call (all kernels of the block) of every streaming block at 5e7 input elements — a scan
for outliers against each block's compulsory traffic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 50_000_000
xc = torch.rand(2 * n, device="cuda") * 2 - 1
xb = torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8)
y = torch.empty(4 * n + 65536, device="cuda")
at = rr.low_pass(200e3, 44.1e3, 500.0)
cases = [
    ("QuadratureDemod exact", lambda: rr.QuadratureDemod(1.0, rr.ATAN2_EXACT), xc, n, n, 12),
    ("QuadratureDemod fast-math", lambda: rr.QuadratureDemod(1.0, rr.ATAN2_FAST), xc, n, n, 12),
    ("FastFM", lambda: rr.FastFM(), xc, n, n, 12),
    ("MultiplyConst<Float>", lambda: rr.MultiplyConst(0.5), xc, 2 * n, 2 * n, 8),
    ("MultiplyConst<Complex>", lambda: rr.MultiplyConst(0.5 + 1j, np.complex64), xc, n, n, 16),
    ("RtlSdrDecode", lambda: rr.RtlSdrDecode(), xb, 2 * n, n, 5),
    ("FftStream 1024", lambda: rr.FftStream(1024), xc, n, n, 16),
    ("FftStream 4096", lambda: rr.FftStream(4096), xc, n, n, 16),
    ("FftStream 64", lambda: rr.FftStream(64), xc, n, n, 16),
    (f"FftFilterFloat {len(at)} taps", lambda: rr.FftFilterFloat(at), xc, 2 * n, 2 * n, 8),
    ("Hilbert 65", lambda: rr.Hilbert(65), xc, 2 * n, 2 * n, 12),
    ("FirFilter<Float> 127 d=1", lambda: rr.FirFilter(rr.low_pass(10e6, 1e6, 190e3)), xc, 2 * n, 2 * n, 8),
]
for name, mk, x, n_in, cap, bpe in cases:
    b = mk()
    for _ in range(2):
        b.work_dev(x.data_ptr(), n_in, y.data_ptr(), cap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    consumed = 0
    for _ in range(reps):
        st, c, p, need = b.work_dev(x.data_ptr(), n_in, y.data_ptr(), cap)
        consumed = c
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{name:32s} {dt*1e3:8.3f} ms  consumed {consumed:>11,d}  {bpe * consumed / dt / 1e12:6.2f} TB/s of {bpe} B/elem compulsory")
