#!/usr/bin/env python3
"""GPU box: per-call GPU time (us) of the blocks' DEFAULT path selection against window size, 16 k ... 32 M input samples:
looks for windows where a call costs more than a larger one (a per-call kernel choice gone wrong)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(1)
NMAX = 32_000_000
x = torch.rand(2 * NMAX, device="cuda") * 2 - 1
xb = (torch.rand(2 * NMAX, device="cuda") * 255).to(torch.uint8)
y = torch.empty(2 * NMAX + 65536, device="cuda")
def ct(L): return ((rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)) / L).astype(np.complex64)
def rt(L): return (rng.uniform(-1, 1, L) / L).astype(np.float32)
def t(mk, xin, in_mult, out_div, reps=5):
    row = []
    for n in (16_384, 131_072, 512_000, 2_000_000, 8_000_000, 32_000_000):
        blk = mk()
        cap = n * 6 // out_div + 65536
        for _ in range(2): blk.work_dev(xin.data_ptr(), n * in_mult, y.data_ptr(), cap, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): blk.work_dev(xin.data_ptr(), n * in_mult, y.data_ptr(), cap, s)
        b.record(); torch.cuda.synchronize()
        row.append(f"{a.elapsed_time(b) / reps * 1e3:8.1f}")
    return " ".join(row)
cases = [
    ("FirFilter c32 127 /1", lambda: rr.FirFilter(ct(127)), x, 1, 6), ("FirFilter c32 255 /8", lambda: rr.FirFilter(ct(255), deci=8), x, 1, 6),
    ("FirFilter c32 401 /6", lambda: rr.FirFilter(ct(401), deci=6), x, 1, 6), ("FirFilter c32 127 /20", lambda: rr.FirFilter(ct(127), deci=20), x, 1, 6),
    ("FirFilter c32 2467 /9", lambda: rr.FirFilter(ct(2467), deci=9), x, 1, 6), ("FirFilter c32 3599 /6", lambda: rr.FirFilter(ct(3599), deci=6), x, 1, 6),
    ("FirFilter f32 127 /1", lambda: rr.FirFilter(rt(127)), x, 1, 6), ("FirFilter f32 5000 /4", lambda: rr.FirFilter(rt(5000), deci=4), x, 1, 6),
    ("FirFilter f32 31 /32", lambda: rr.FirFilter(rt(31), deci=32), x, 1, 6),
    ("FftFilter 401", lambda: rr.FftFilter(ct(401)), x, 1, 6), ("FftFilter 2467", lambda: rr.FftFilter(ct(2467)), x, 1, 6), ("FftFilter 3330", lambda: rr.FftFilter(ct(3330)), x, 1, 6),
    ("Hilbert 65", lambda: rr.Hilbert(65), x, 1, 6), ("Hilbert 1001", lambda: rr.Hilbert(1001), x, 1, 6), ("Hilbert 4001", lambda: rr.Hilbert(4001), x, 1, 6),
    ("HilbertFir 65*255 /8", lambda: rr.HilbertFir(65, ct(255), 8), x, 1, 6), ("HilbertFir 65*255 /1", lambda: rr.HilbertFir(65, ct(255), 1), x, 1, 6),
    ("HilbertFir 65*2467 /32", lambda: rr.HilbertFir(65, ct(2467), 32), x, 1, 6),
    ("FmChain 463 1:6", lambda: rr.FmChain(ct(463), 1, 6), x, 1, 6), ("FmChain 2467 1:9", lambda: rr.FmChain(ct(2467), 1, 9), x, 1, 6),
    ("FmChain 3599 1:6", lambda: rr.FmChain(ct(3599), 1, 6), x, 1, 6), ("FmChain 2467 25:128", lambda: rr.FmChain(ct(2467), 25, 128), x, 1, 6),
    ("FmChain 3330 1:7", lambda: rr.FmChain(ct(3330), 1, 7), x, 1, 6), ("FmChainU8 463 1:6", lambda: rr.FmChainU8(ct(463), 1, 6), xb, 2, 6),
    ("AudioChain 963 6:25", lambda: rr.AudioChain(rt(963), 6, 25, 1.0), x, 1, 6),
    ("Resampler 25:128", lambda: rr.RationalResampler(25, 128, np.complex64), x, 1, 6), ("QuadDemod", lambda: rr.QuadratureDemod(1.0), x, 1, 6),
]
print(f"{'window (input samples)':26s} " + " ".join(f"{n:>8s}" for n in ("16k", "128k", "512k", "2M", "8M", "32M")))
for name, mk, xin, im, od in cases:
    if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]): continue
    print(f"{name:26s} {t(mk, xin, im, od)}", flush=True)
