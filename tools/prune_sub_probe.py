#!/usr/bin/env python3
"""GPU box: decimations d = D * sub (D = 16 / 8 / 4 the pruned tile, every sub-th kept sample stored) against the blocks' other
kernels: FirFilter<Complex>, FirFilter<Float>, HilbertFir(65); ms per 1e8 input samples, pruned (fir_prune=1) / other (-1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * (n // 4 + 65536), device="cuda")
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
def t(mk, d):
    out = []
    for opt in (1, -1):
        with rr.build_options(fir_prune=opt):
            blk = mk()
        cap = n // d + 8
        for _ in range(2): blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / 3)
    return f"{out[0]:.3f}/{out[1]:.3f}" + ("*" if out[0] < out[1] else " ")
ds = (12, 20, 24, 32, 40, 48, 64, 96, 128)
for L in (31, 127, 255, 401, 1000, 2000):
    tc = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) / L).astype(np.complex64)
    tf = (rng.standard_normal(L) / L).astype(np.float32)
    print(f"FirC32 L={L:5d} " + " ".join(f"/{d}={t(lambda: rr.FirFilter(tc, deci=d), d)}" for d in ds), flush=True)
    print(f"FirF32 L={L:5d} " + " ".join(f"/{d}={t(lambda: rr.FirFilter(tf, deci=d), d)}" for d in ds), flush=True)
    print(f"HilFir L={L:5d} " + " ".join(f"/{d}={t(lambda: rr.HilbertFir(65, tc, d), d)}" for d in ds), flush=True)
