#!/usr/bin/env python3
"""GPU box: FftFilter GPU time per call against window size for every admissible tile size (rr_build_opts.fft_log2f) and
the block's own choice — where smaller tiles (more workgroups) beat the tile chosen for large batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(3)
for L in (65, 127, 401, 700, 1000, 1500):
    taps = (rng.standard_normal(L) / L).astype(np.complex64)
    for n in (128_000, 512_000, 2_000_000, 8_000_000):
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        y = torch.empty(2 * (n + 4096), device="cuda")
        row = []
        for lg in (0, 10, 11, 12):
            if lg and (1 << lg) < L + 64: continue
            with rr.build_options(**({"fft_log2f": lg} if lg else {})):
                b = rr.FftFilter(taps)
            for _ in range(3): b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 1024, s)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(10): b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 1024, s)
            e.record(); torch.cuda.synchronize()
            row.append(f"{'auto' if not lg else 'F=' + str(1 << lg)} {a.elapsed_time(e) / 10 * 1e3:6.1f}")
        print(f"L={L:5d} n={n // 1000:5d}k  " + "  ".join(row))
        del x, y
