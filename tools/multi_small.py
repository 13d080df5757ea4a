#!/usr/bin/env python3
"""GPU box: FmMulti (32 channels) per-call GPU time against window size, decimate-first tiles vs the 2048-point kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
from rustradio_amd import multi
s = torch.cuda.current_stream().cuda_stream
taps = multi.cfg4_taps(rr.low_pass_complex(2.4e6, 100e3, 12.5e3), range(32))
for n in (100_000, 256_000, 512_000, 1_000_000, 1_500_000, 2_400_000, 3_000_000, 5_000_000, 10_000_000, 24_000_000):
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    cap = n // 6 + 1024
    y = torch.empty(32 * cap, device="cuda")
    row = []
    variants = (("poly", {}),) if os.environ.get("POLY_ONLY") else (("poly", {}), ("half", {"fm_poly": -1}), ("full", {"fm_poly": -1, "fm_full": 1}))
    for nm, o in variants:
        with rr.build_options(**o):
            b = rr.FmMulti(taps, 1, 6, 1.0)
        for _ in range(4): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        e.record(); torch.cuda.synchronize()
        row.append(f"{nm} {a.elapsed_time(e) / 30 * 1e3:7.1f} us")
    print(f"n={n:9d}  " + "   ".join(row))
