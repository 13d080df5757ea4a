#!/usr/bin/env python3
"""GPU box: Hilbert over ntaps, FmMulti over channels x decimation, FmChainU8 over decimation, FftStream over sizes:
ms per call on large windows (default path selection).  Looks for cliffs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
def t(blk, x, nin, y, cap, reps=3):
    for _ in range(2): blk.work_dev(x.data_ptr(), nin, y.data_ptr(), cap, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): blk.work_dev(x.data_ptr(), nin, y.data_ptr(), cap, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
which = sys.argv[1:] or ["hilbert", "multi", "u8", "fftstream"]
n = 100_000_000
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.empty(2 * n + 65536, device="cuda")
if "hilbert" in which:
    print("Hilbert (1e8 f32): " + " ".join(f"{L}={t(rr.Hilbert(L), x, n, y, n):.3f}" for L in (3, 31, 65, 129, 255, 511, 1001, 2001, 4001)), flush=True)
if "fftstream" in which:
    xc = torch.rand(2 * 50_000_000, device="cuda")
    print("FftStream (5e7 c32): " + " ".join(f"{N}={t(rr.FftStream(N), xc, 50_000_000, y, 50_000_000):.3f}" for N in (8, 64, 100, 256, 1000, 1024, 2048, 3000, 4096, 8192, 16384, 20000, 65536)), flush=True)
if "u8" in which:
    nb = 48_000_000
    xb = (torch.rand(nb, device="cuda") * 255).to(torch.uint8)
    for L in (463, 2467):
        taps = ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
        print(f"FmChainU8 (2.4e7 samples) L={L}: " + " ".join(f"1:{D}={t(rr.FmChainU8(taps, 1, D), xb, nb, y, nb // 2):.3f}" for D in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 20)) + " " +
              " ".join(f"{I}:{D}={t(rr.FmChainU8(taps, I, D), xb, nb, y, nb):.3f}" for I, D in ((25, 128), (3, 7))), flush=True)
if "multi" in which:
    nm = 2_400_000
    xm = torch.rand(2 * nm, device="cuda") * 2 - 1
    for L in (463, 2467):
        for C in (1, 2, 8, 32, 33, 64, 100):
            taps = ((rng.uniform(-1, 1, (C, L)) + 1j * rng.uniform(-1, 1, (C, L))) / L).astype(np.complex64)
            ym = torch.empty(C * (nm + 8192), device="cuda")
            print(f"FmMulti (2.4e6 samples) L={L} C={C:3d}: " + " ".join(f"{I}:{D}={t(rr.FmMulti(taps, I, D), xm, nm, ym, nm * I // D + 8192):.3f}" for I, D in ((1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8), (1, 9), (1, 10), (1, 16), (25, 128), (3, 7))), flush=True)
            del ym
