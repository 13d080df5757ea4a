import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import rustradio_amd as rr
from oracle import pyoracle as orc
from harness import run_chain
from test_gpu_edges_fullsize import _nan_poisoned, rnd_c
n = 400_000
ph = np.cumsum(0.3 * np.sin(2 * np.pi * 1e-3 * np.arange(n)))
clean = (np.exp(1j * ph) + 0.02 * rnd_c(n, 3)).astype(np.complex64)
taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
S = 561
x = _nan_poisoned(clean, 13, extra=[(n // 2 // S) * S - 3, (n // 2 // S) * S + 2])
bad_in = np.flatnonzero(~np.isfinite(x.real) | ~np.isfinite(x.imag))
print("bad inputs", bad_in, "blocks", bad_in // S)
I, D = 1, 6
blk = rr.FmChain(taps, I, D, 1.0)
got = run_chain([blk], x)
want = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x)
bo, bg = ~np.isfinite(want), ~np.isfinite(got)
mm = np.flatnonzero(bo != bg)
print("mismatches", len(mm))
# group
if len(mm):
    runs = np.split(mm, np.flatnonzero(np.diff(mm) > 1) + 1)
    for r in runs[:40]:
        u = r[0] + 1
        y = u * D // I
        print("run", r[0], r[-1], "len", len(r), "gpu_nan" , bool(bg[r[0]]), "ref_nan", bool(bo[r[0]]), "yb", y, "block", y // S, "off", y % S)
with rr.build_options(fft_nonfinite_tiles=1):
    blk2 = rr.FmChain(taps, I, D, 1.0)
got2 = run_chain([blk2], x)
b2 = ~np.isfinite(got2)
runs = np.split(np.flatnonzero(b2), np.flatnonzero(np.diff(np.flatnonzero(b2)) > 1) + 1)
print("no-pass NaN runs:", [(int(r[0]), int(r[-1])) for r in runs][:30])
runs = np.split(np.flatnonzero(bo), np.flatnonzero(np.diff(np.flatnonzero(bo)) > 1) + 1)
print("ref NaN runs:", [(int(r[0]), int(r[-1])) for r in runs][:30])
