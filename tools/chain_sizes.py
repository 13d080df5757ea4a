#!/usr/bin/env python3
"""GPU box: FmChain (configs[2]: 463 taps, 1:6, fused decimate-first tiles) per-call GPU time against window size —
the slope is the steady cost per tile, the intercept what a launch pays in ramp-up and in the last round of tiles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
rows = []
for n in (600_000, 1_200_000, 2_400_000, 6_000_000, 12_000_000, 24_000_000, 48_000_000, 96_000_000):
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    cap = n // 6 + 1024
    y = torch.empty(cap, device="cuda")
    b = rr.FmChain(taps, 1, 6, 1.0)
    for _ in range(40): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 40
    a.record()
    for _ in range(reps): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    e.record(); torch.cuda.synchronize()
    us = a.elapsed_time(e) / reps * 1e3
    tiles = n / (946 * 6)
    rows.append((tiles, us))
    print(f"n={n:9d}  tiles {tiles:8.0f}  {us:8.1f} us   {us / tiles * 1e3:7.2f} ns/tile   {n / us / 1e3:7.1f} Gsamples/s")
t = np.array([r[0] for r in rows[3:]]); u = np.array([r[1] for r in rows[3:]])
b1, b0 = np.polyfit(t, u, 1)
print(f"fit over the four largest: {b0:.1f} us + {b1 * 1e3:.2f} ns/tile")
