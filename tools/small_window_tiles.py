#!/usr/bin/env python3
"""FftFilter on a device window of n samples: us per work() call on the 1024-, 2048- and 4096-point tiles (rr_build_opts.fft_log2f)
for several tap counts — where the smaller tile's larger workgroup count beats the larger tile's efficiency (round 6: the
per-call tile choice of short filters, csrc/blocks.cpp FftFilter::alt_wins)."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rustradio_amd as rr

s = torch.cuda.current_stream().cuda_stream
for L in (65, 127, 255, 401, 511):
    taps = (np.hamming(L) / L).astype(np.complex64)
    for n in (128_000, 256_000, 512_000, 1_000_000, 1_500_000, 2_000_000, 3_000_000, 4_000_000, 8_000_000):
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        y = torch.empty(2 * (n + 4096), device="cuda")
        row = []
        for lg in (0, 10, 11, 12):
            if lg and (1 << lg) < L + 64:
                row.append(None); continue
            with rr.build_options(**({"fft_log2f": lg} if lg else {})):
                b = rr.FftFilter(taps)
            for _ in range(20):
                b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 4096, s)
            torch.cuda.synchronize()
            k = 300
            t0 = time.perf_counter()
            for _ in range(k):
                b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 4096, s)
            torch.cuda.synchronize()
            row.append(round((time.perf_counter() - t0) / k * 1e6, 2))
        print(f"L={L} n={n}: auto {row[0]}  1024-pt {row[1]}  2048-pt {row[2]}  4096-pt {row[3]}", flush=True)
