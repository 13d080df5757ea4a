#!/usr/bin/env python3
"""GPU box: the fused Hilbert->FIR/8 kernel alone (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.empty(2 * (n // 8 + 8), device="cuda")
taps = rr.low_pass_complex(100e6, 5e6, 943e3)
hf = rr.HilbertFir(65, taps, 8)
for rep in range(2):
    hf.work_dev(x.data_ptr(), n, y.data_ptr(), n // 8 + 8, 0)
torch.cuda.synchronize()
hf.set_profiling(True)
for rep in range(5):
    hf.work_dev(x.data_ptr(), n, y.data_ptr(), n // 8 + 8, 0)
torch.cuda.synchronize()
m, k = hf.profile()
print(f"fused hilbert+fir/8 {m/k:.4f} ms")
