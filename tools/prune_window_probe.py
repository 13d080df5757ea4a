#!/usr/bin/env python3
"""GPU box: where the pruned-inverse tiles (deci 4 / 8 / 16) start to pay against the block's small-window paths, by window
size: forced pruned (fir_prune=1) / never pruned (fir_prune=-1) / the block's own per-call choice — GPU us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(3)
for L, d in ((127, 4), (401, 4), (255, 8), (1000, 8), (401, 16), (2000, 16)):
    taps = (rng.standard_normal(L) / L).astype(np.complex64)
    for n in (512_000, 2_000_000, 8_000_000, 16_000_000, 32_000_000, 64_000_000):
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        y = torch.empty(2 * (n // d + 64), device="cuda")
        row = []
        for nm, o in (("pruned", {"fir_prune": 1}), ("other", {"fir_prune": -1, "fir_poly": -1}), ("auto", {})):
            with rr.build_options(**o):
                b = rr.FirFilter(taps, deci=d)
            for _ in range(3): b.work_dev(x.data_ptr(), n, y.data_ptr(), n // d + 8, s)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): b.work_dev(x.data_ptr(), n, y.data_ptr(), n // d + 8, s)
            e.record(); torch.cuda.synchronize()
            row.append(f"{nm} {a.elapsed_time(e) / 10 * 1e3:7.1f}")
        print(f"L={L:5d} /{d:<2d} n={n // 1000:6d}k  " + "  ".join(row))
        del x, y
