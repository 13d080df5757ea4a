#!/usr/bin/env python3
"""GPU box: cost of .translate() (separate rotator kernel after the FIR) on the configs[4] channelizer, 1e8 samples."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.empty(2 * (n // 8 + 8), device="cuda")
taps = rr.low_pass_complex(100e6, 5e6, 943e3)
for name, blk in (("HilbertFir", rr.HilbertFir(65, taps, 8)), ("HilbertFir + translate", rr.HilbertFir(65, taps, 8, translate=(100e6, 12.5e6)))):
    for _ in range(3):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), n // 8 + 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), n // 8 + 8)
    torch.cuda.synchronize()
    print(f"{name:24s} {(time.perf_counter() - t0) / 20 * 1e3:.4f} ms per 1e8 samples (wall, all kernels)")
