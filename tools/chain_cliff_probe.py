#!/usr/bin/env python3
"""GPU box: FmChain (default path selection) over decimations 1..20 and a few interpolating ratios x tap counts: ms per
2.4e7 input samples.  Looks for cliffs in the path selection (a neighbouring shape several times slower)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
rng = np.random.default_rng(1)
s = torch.cuda.current_stream().cuda_stream
n = 24_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
for L in (127, 463, 1000, 2467, 4000, 7000):
    taps = ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
    row = []
    for I, D in [(1, d) for d in range(1, 21)] + [(2, 3), (3, 7), (25, 128), (5, 1), (2, 1)]:
        cap = n * I // D + 16384
        y = torch.empty(cap, device="cuda")
        try:
            b = rr.FmChain(taps, I, D)
            for _ in range(2):
                b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
            e.record(); torch.cuda.synchronize()
            row.append(f"{I}:{D}={a.elapsed_time(e) / 4:.3f}")
        except Exception as ex:
            row.append(f"{I}:{D}=ERR")
        del y
    print(f"L={L:5d}  " + "  ".join(row), flush=True)
