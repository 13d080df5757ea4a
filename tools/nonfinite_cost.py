#!/usr/bin/env python3
"""GPU: what non-finite input samples cost FftFilter's reference-block pass (csrc/kernels_misc.hip k_ref_blocks_nonfinite):
ms per call of 1e7 samples on device windows with no bad sample (the probe only), one, a hundred scattered, and a window
that is all NaN — 401 and 5000 taps.   python tools/nonfinite_cost.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import rustradio_amd as rr  # noqa: E402


def run(L, bad, n=10_000_000, calls=8):
    rng = np.random.default_rng(1)
    taps = ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
    blk = rr.FftFilter(taps)
    x = torch.rand(2 * n, dtype=torch.float32, device="cuda") - 0.5
    if bad == "all":
        x[:] = float("nan")
    else:
        for p in rng.integers(0, n, bad):
            x[2 * int(p)] = float("nan")
    y = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    ts = []
    for _ in range(calls):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
        b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts[2:]))


for L in (401, 5000):
    base = run(L, 0)
    print(f"FftFilter {L} taps, 1e7 samples per call: clean {base:.3f} ms" +
          "".join(f"; {k} bad {run(L, k):.3f} ms" for k in (1, 100, "all")))
