#!/usr/bin/env python3
"""GPU box: repeat FftFilter over 1e8 samples and compare runs of the SAME input bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
taps = rr.low_pass_complex(10e6, 1e6, 60e3)
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.rand(2 * n, generator=g, device="cuda") * 2 - 1
s = torch.cuda.current_stream().cuda_stream
def filt():
    y = torch.full((2 * (n + 1024),), float("nan"), device="cuda")
    b = rr.FftFilter(taps)
    st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 1024, s)
    torch.cuda.synchronize()
    return y[:2 * p]
ref = filt()
print("nan in ref:", int(torch.isnan(ref).sum()))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    y = filt()
    bad = (y != ref) | torch.isnan(y)
    nb = int(bad.sum())
    if nb:
        idx = torch.nonzero(bad).flatten()
        print(f"run {i}: {nb} differing floats, first {int(idx[0])//2} last {int(idx[-1])//2} (sample index), tile {int(idx[0])//2//1648}")
    else:
        print(f"run {i}: identical")
