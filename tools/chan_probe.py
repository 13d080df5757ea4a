#!/usr/bin/env python3
"""GPU box: per-kernel time of the channelizer chain (Hilbert 65 -> FIR 255 taps / 8) at 1e8 samples."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(n, device="cuda") * 2 - 1
a = torch.empty(2 * n, device="cuda"); y = torch.empty(2 * (n // 8 + 8), device="cuda")
taps = rr.low_pass_complex(100e6, 5e6, 943e3)
h, f = rr.Hilbert(65), rr.FirFilter(taps, deci=8)
s = 0
for rep in range(3):
    h.work_dev(x.data_ptr(), n, a.data_ptr(), n, s); f.work_dev(a.data_ptr(), n, y.data_ptr(), n // 8 + 8, s)
torch.cuda.synchronize()
h.set_profiling(True); f.set_profiling(True)
for rep in range(10):
    h.work_dev(x.data_ptr(), n, a.data_ptr(), n, s); f.work_dev(a.data_ptr(), n, y.data_ptr(), n // 8 + 8, s)
torch.cuda.synchronize()
mh, kh = h.profile(); mf, kf = f.profile()
hf = rr.HilbertFir(65, taps, 8)
for rep in range(3):
    hf.work_dev(x.data_ptr(), n, y.data_ptr(), n // 8 + 8, s)
torch.cuda.synchronize()
hf.set_profiling(True)
for rep in range(10):
    hf.work_dev(x.data_ptr(), n, y.data_ptr(), n // 8 + 8, s)
torch.cuda.synchronize()
mc, kc = hf.profile()
print(f"fused hilbert+fir/8 {mc/kc:.4f} ms ({5*n/(mc/kc*1e-3)/1e12:.2f} TB/s alg, {n/8*319*4/(mc/kc*1e-3)/1e12:.1f} TFLOP/s)")
print(f"hilbert {mh/kh:.4f} ms  ({12*n/(mh/kh*1e-3)/1e12:.2f} TB/s alg)   fir/8 {mf/kf:.4f} ms ({9*n/(mf/kf*1e-3)/1e12:.2f} TB/s alg, {n/8*255*4/(mf/kf*1e-3)/1e12:.1f} TFLOP/s)")
