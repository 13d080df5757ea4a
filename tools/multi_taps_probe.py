#!/usr/bin/env python3
"""GPU box: FmMulti (32 channels, 1:6) per-call GPU time against the tap count, across the 4094-tap edge of the fused
shared-forward kernels (beyond it: Parallel = one fused chain per channel, compose.cpp).

    python tools/multi_taps_probe.py [samples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000
s = torch.cuda.current_stream().cuda_stream
x = torch.rand(2 * n, device="cuda") * 2 - 1
cap = n // 6 + 4096
y = torch.empty(32 * cap, device="cuda")
rng = np.random.default_rng(0)
for L in (463, 2000, 3000, 3800, 4094, 4600, 5000, 5400, 5568, 5569, 8000):
    taps = (rng.standard_normal((32, L)) + 1j * rng.standard_normal((32, L))).astype(np.complex64) / L
    b = rr.FmMulti(taps, 1, 6, 1.0)
    for _ in range(3):
        b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    a.record()
    for _ in range(reps):
        st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    e.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(e) / reps
    print(f"L={L:5d}  {b.name[:60]:60s}  {ms:9.3f} ms per call  consumed {c}  {ms * 1e6 / max(c, 1) / 32:8.3f} ns per channel-sample")
