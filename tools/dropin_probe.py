#!/usr/bin/env python3
"""GPU box: where the time of one host-window rr_block_work() goes (4,096,000-byte windows, pageable vs registered)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import rustradio_amd as rr

def t_of(f, n=30):
    for _ in range(3): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6

rng = np.random.default_rng(0)
n = 512_000
x = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
out = np.zeros(n, np.complex64)
taps = rr.low_pass_complex(10e6, 1e6, 60e3)
for reg in (False, True):
    if reg:
        rr.host_register(x); rr.host_register(out)
    blk = rr.FftFilter(taps)
    us = t_of(lambda: blk.work_into(x, out, n))
    print(f"FftFilter work_host registered={reg}: {us:.0f} us/call -> {n / us:.0f} Msamples/s")
    d = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    xt = torch.from_numpy(x.view(np.float32))
    def h2d():
        d.copy_(xt, non_blocking=True); torch.cuda.synchronize()
    print(f"   torch H2D 4 MB: {t_of(h2d):.0f} us")
    ot = torch.from_numpy(out.view(np.float32))
    def d2h():
        ot.copy_(d, non_blocking=True); torch.cuda.synchronize()
    print(f"   torch D2H 4 MB: {t_of(d2h):.0f} us")
    dy = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    def dev():
        blk.work_dev(d.data_ptr(), n, dy.data_ptr(), n); torch.cuda.synchronize()
    print(f"   work_dev + sync: {t_of(dev):.0f} us")
b = rng.integers(0, 256, 4_096_000, dtype=np.uint8)
o = np.zeros(1_024_000, np.float32)
rr.host_register(b); rr.host_register(o)
fm = rr.FmChainU8(rr.low_pass_complex(2.4e6, 100e3, 12.5e3), 1, 6)
us = t_of(lambda: fm.work_into(b, o, len(o)))
print(f"FmChainU8 work_host registered: {us:.0f} us/call -> {2_048_000 / us:.0f} Msamples/s")
