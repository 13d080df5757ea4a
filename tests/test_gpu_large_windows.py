"""GPU: random LARGE windows (one work() call of 1.5e6 ... 1.2e7 samples) through the decimate-first chain and multi-channel
kernels — many tiles per workgroup, the next-tile prefetch of k_fm_chain_polyw, the 8- and 12-wave work splits of
k_fm_multi_poly*, partial channel rounds — every checked channel against its own oracle chain
FftFilter -> RationalResampler(1, D) -> QuadratureDemod (examples/rtl_fm.rs:381-419 wiring).  tools/fuzz_large.py runs the
same trial on further seeds."""
import numpy as np
import pytest

from harness import angle_parity, run_chain
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def large_window_trial(rr, seed):
    """one random trial; returns (description, worst share of the propagated allowance used); raises on a mismatch"""
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.choice([2, 3, 4, 5, 6, 6, 7, 8]))
    L = int(rng.choice([63, 127, 200, 401, 463, 600, 1025]))
    if (L + D - 1) // D > 448:
        L = 463
    multi = bool(rng.integers(0, 2))
    nch = int(rng.integers(9, 41)) if multi else 1
    n = int(rng.integers(1_500_000, 4_000_000 if multi else 12_000_000))
    fs = 2.4e6
    t = np.arange(n, dtype=np.float64)
    phi = 2 * np.pi * np.cumsum(50e3 + 60e3 * np.sin(2 * np.pi * 1.3e3 * t / fs)) / fs
    x = (np.exp(1j * phi) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    proto = (rng.uniform(-1, 1, L) + 1j * rng.uniform(-1, 1, L)).astype(np.complex64) / max(1, L // 8)
    kk = np.arange(L, dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * (c * 7e3) * kk / fs)).astype(np.complex64)
                     for c in range(nch)])
    cap = n // D + 2048
    blk = rr.FmMulti(taps, 1, D, 1.0) if multi else rr.FmChain(taps[0], 1, D, 1.0)
    st, c, p, need, out = blk.work(x, cap)
    outs = out.reshape(nch, -1)
    worst = 0.0
    check = range(nch) if nch <= 4 else sorted(set(int(v) for v in rng.choice(nch, 4, replace=False)) | {0, nch - 1})
    for ch in check:
        yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D), orc.QuadratureDemod(1.0)], x, stream_bytes=8 * (n + 16))
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D)], x, stream_bytes=8 * (n + 16))
        yg = outs[ch][:p]
        assert len(yg) == len(yo) > 0, (seed, ch, len(yg), len(yo))
        r = angle_parity(yg, yo, ro, 1e-5, L // D + 2)
        assert r["used"] <= 1.0, (seed, ch, r)
        worst = max(worst, r["used"])
    return f"{'multi' if multi else 'chain'} D={D} L={L} nch={nch} n={n}: produced {p}", worst


@pytest.mark.parametrize("seed", range(8))
def test_large_windows(seed):
    import rustradio_amd as rr
    what, worst = large_window_trial(rr, seed)
    print(f"{what}, at most {worst:.3f} of the propagated allowance used")
