#!/usr/bin/env python3
"""GPU box: decimate-first tiles for LONG phases (more than 448 taps per phase: fewer than 576 of a tile's 1024 positions are
output) against the block's other kernels.  FmChain: ms per 2.4e7 samples; FmMulti (32 channels): ms per 2.4e6 samples.
fm_poly=+1 forces the decimate-first tiles wherever the kernel exists, -1 forbids them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
rng = np.random.default_rng(1)
s = torch.cuda.current_stream().cuda_stream

def t_of(blk, x, n, y, cap, reps=6):
    for _ in range(2):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps

n = 24_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(n // 2 + 8192, device="cuda")
for D in (4, 5, 6, 8, 10):
    for Ls in (400, 448, 520, 600, 680, 760):
        L = D * Ls - 1
        taps = ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
        res = {}
        for name, opt in (("poly", 1), ("other", -1)):
            try:
                with rr.build_options(fm_poly=opt):
                    b = rr.FmChain(taps, 1, D)
                res[name] = t_of(b, x, n, y, n // 2 + 8192)
            except Exception as ex:
                res[name] = float("nan")
        print(f"FmChain  D={D:2d} Ls={Ls:4d} L={L:5d}  poly {res['poly']:.4f}  other {res['other']:.4f}  {'POLY' if res['poly'] < res['other'] else 'other'}", flush=True)
n = 2_400_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
cap = n // 2 + 8192
y = torch.empty(32 * cap, device="cuda")
for D in (2, 4, 6, 8):
    for Ls in (448, 520, 600, 680, 760):
        L = D * Ls - 1
        if L > 3329:
            continue
        taps = ((rng.uniform(-1, 1, (32, L)) + 1j * rng.uniform(-1, 1, (32, L))) / L).astype(np.complex64)
        res = {}
        for name, opt in (("poly", 1), ("other", -1)):
            try:
                with rr.build_options(fm_poly=opt):
                    b = rr.FmMulti(taps, 1, D)
                res[name] = t_of(b, x, n, y, cap, reps=3)
            except Exception as ex:
                res[name] = float("nan")
        print(f"FmMulti  D={D:2d} Ls={Ls:4d} L={L:5d}  poly {res['poly']:.4f}  other {res['other']:.4f}  {'POLY' if res['poly'] < res['other'] else 'other'}", flush=True)
