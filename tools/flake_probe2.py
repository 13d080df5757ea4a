#!/usr/bin/env python3
"""GPU box: as flake_probe.py, tile by tile — which tiles (if any) differ between two runs of FftFilter on the same input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
taps = rr.low_pass_complex(10e6, 1e6, 60e3)
L, S = len(taps), 623
g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0002)
x1 = torch.rand(2 * n, generator=g, device="cuda") * 2 - 1
x2 = torch.rand(2 * n, generator=g, device="cuda") * 2 - 1
s = torch.cuda.current_stream().cuda_stream
def filt(x, tag):
    y = torch.full((2 * (n + 1024),), float("nan"), device="cuda")
    b = rr.FftFilter(taps)
    st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 1024, s)
    torch.cuda.synchronize()
    print(tag, "status", st, c, p, need, "nan:", int(torch.isnan(y[:2 * p]).sum()), "absmax", float(y[:2*p].abs().max()))
    return y[:2 * p]
def check(y, x, tag):
    for s0 in (0, 50_000_000, 99_990_000):
        h0 = max(0, s0 - (L - 1))
        xin = x[2 * h0:2 * (s0 + 3000)].cpu().numpy().view(np.complex64)
        ref = np.convolve(xin.astype(np.complex128), taps.astype(np.complex128))[s0 - h0:s0 - h0 + 3000]
        got = y[2 * s0:2 * (s0 + 3000)].cpu().numpy().view(np.complex64)
        print("  ", tag, s0, "err", float(np.max(np.abs(got - ref)) / np.max(np.abs(ref))))
y1 = filt(x1, "y1"); check(y1, x1, "y1")
y2 = filt(x2, "y2"); check(y2, x2, "y2")
a, b = 0.75, -1.5
x12 = a * x1 + b * x2
y12 = filt(x12, "y12"); check(y12, x12, "y12")
comb = a * y1 + b * y2
diff = (y12 - comb).abs()
print("lin err", float(diff.max() / y12.abs().max()), "argmax sample", int(diff.argmax()) // 2)
check(y1, x1, "y1 again")
