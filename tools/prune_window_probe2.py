#!/usr/bin/env python3
"""GPU box: as prune_window_probe.py for the real-stream forms — Hilbert->FirFilter (fused) and FirFilter<Float>."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(3)
def run(mk, n, cap, o):
    with rr.build_options(**o):
        b = mk()
    for _ in range(3): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / 10 * 1e3
for kind in ("hilbertfir", "firfloat"):
    for L, d in ((255, 4), (255, 8), (255, 16), (1000, 16)):
        tc = (rng.standard_normal(L) / L).astype(np.complex64)
        tf = tc.real.astype(np.float32)
        for n in (2_000_000, 8_000_000, 32_000_000, 64_000_000, 128_000_000):
            x = torch.rand(n, device="cuda") * 2 - 1
            y = torch.empty(2 * (n // d + 64), device="cuda")
            mk = (lambda: rr.HilbertFir(65, tc, d)) if kind == "hilbertfir" else (lambda: rr.FirFilter(tf, deci=d))
            row = [f"{nm} {run(mk, n, n // d + 8, o):7.1f}" for nm, o in (("pruned", {"fir_prune": 1}), ("other", {"fir_prune": -1}), ("auto", {}))]
            print(f"{kind:10s} L={L:5d} /{d:<2d} n={n // 1000:6d}k  " + "  ".join(row))
            del x, y
