"""GPU parity: every HIP block against the CPU oracle on identical seeded input, driven
through identical work() call sequences.  Bar (BASELINE.json north_star / SURVEY §8d):
    max|y_gpu - y_ref| / max|y_ref| <= 1e-5   (QuadDemod: scale = pi*|gain|),
lengths and the (status, consumed, produced, need) protocol must match exactly;
RationalResampler is a pure copy and must be bit-exact."""
import numpy as np
import pytest

from harness import AGAIN, WAIT_DST, WAIT_SRC, angle_parity, drive_pageable, drive_registered, knob, max_norm_err, run_chain
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def rr():
    import rustradio_amd
    return rustradio_amd


def rnd_c(n, seed):
    r = np.random.default_rng(seed)
    return (r.uniform(-1, 1, n) + 1j * r.uniform(-1, 1, n)).astype(np.complex64)


def rnd_f(n, seed):
    return np.random.default_rng(seed).uniform(-1, 1, n).astype(np.float32)


def both(rr, make, x, stream_bytes=4_096_000, scale=None, exact=False):
    lo, lg = [], []
    yo = run_chain(make(orc), x, stream_bytes=stream_bytes, log=lo)
    yg = run_chain(make(rr), x, stream_bytes=stream_bytes, log=lg)
    assert lo == lg, "work() protocol differs from the reference restatement"
    assert len(yo) == len(yg)
    if exact:
        assert np.array_equal(yo, yg)
        return 0.0
    e = max_norm_err(yg, yo, scale)
    assert e <= TOL, e
    return e


@pytest.mark.parametrize("L,deci,cplx", [(1, 1, False), (1, 7, False), (2, 1, True), (33, 3, True), (127, 1, False),
                                         (127, 1, True), (255, 8, False), (255, 8, True), (64, 5, False),
                                         (401, 2, False), (3, 40, True), (1000, 1, False), (9000, 16, False)])
def test_fir_complex(rr, L, deci, cplx):
    x = rnd_c(60000, L * 7 + deci)
    taps = rnd_c(L, L) / max(1, L // 8)
    if not cplx:
        taps = taps.real.astype(np.complex64)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x)


@pytest.mark.parametrize("cfg", range(8))
@pytest.mark.parametrize("L,deci,cplx", [(255, 8, False), (255, 8, True), (64, 5, False), (33, 3, True), (200, 1, False),
                                         (77, 2, True), (31, 12, False)])
def test_fir_every_tile_shape(rr, monkeypatch, cfg, L, deci, cplx):
    """Every (threads, outputs/thread, phase split) tile shape of the FIR kernel, forced through the
    RR_FIR_CFG knob (the launcher otherwise picks by input size), incl. streams with boundary tiles."""
    knob(rr, monkeypatch, fir_cfg=cfg)
    knob(rr, monkeypatch, fir_path="direct")      # d = 1 filters would otherwise take the overlap-save tiles
    x = rnd_c(40000, L * 11 + deci + cfg)
    taps = rnd_c(L, L + 1) / max(1, L // 8)
    if not cplx:
        taps = taps.real.astype(np.complex64)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x[:9000], stream_bytes=8 * 2500)
    xf = rnd_f(30000, cfg + 5)
    both(rr, lambda m: [m.FirFilter(taps.real.copy(), deci=deci)], xf)


@pytest.mark.parametrize("path", ["direct", "fft", "auto"])
@pytest.mark.parametrize("L,cplx", [(1, False), (3, True), (16, False), (27, True), (28, True), (39, False), (40, False), (127, False), (127, True), (401, True),
                                    (1000, False), (2467, False), (9000, True)])
def test_fir_nondecimating_both_paths(rr, monkeypatch, path, L, cplx):
    """FirFilter with deci = 1 through the direct-form kernel and through the overlap-save FFT tiles
    (chosen automatically beyond a few taps): same work() protocol, both within 1e-5 of the oracle,
    whole windows and small rings (boundary tiles, windows shorter than a tile)."""
    if path == "direct":
        knob(rr, monkeypatch, fir_path="direct")
    elif path == "fft":
        knob(rr, monkeypatch, fir_path="fft")
    x = rnd_c(90000, L * 5 + 1)
    taps = rnd_c(L, L + 9) / max(1, L // 8)
    if not cplx:
        taps = taps.real.astype(np.complex64)
    both(rr, lambda m: [m.FirFilter(taps)], x)
    both(rr, lambda m: [m.FirFilter(taps)], x[:30000], stream_bytes=8 * (L + 1500))
    f = rr.FirFilter(taps)
    assert rr.fir_uses_fft_tiles(f) == (path == "fft" or (path == "auto" and L >= (28 if cplx else 40)))


@pytest.mark.parametrize("path", ["direct", "fft", "auto", "prune", "half", "poly"])
@pytest.mark.parametrize("L,deci,cplx", [(5, 4, True), (40, 4, False), (500, 4, True), (64, 8, False), (1020, 8, True), (3, 16, False), (2049, 16, True),
                                         (5, 2, False), (127, 2, False), (128, 2, True), (600, 2, False), (601, 2, True), (401, 6, True), (90, 10, False), (700, 14, True), (33, 4096, True), (64, 22, False), (127, 3, True), (255, 8, True), (401, 7, False),
                                         (1000, 16, True), (64, 100, False), (2000, 5, False), (3584, 4096, False), (5000, 3, True), (9000, 16, False),
                                         (300, 3000, True),
                                         # round 4: decimations 9, 11, 13-15 and long phases on the decimate-first tiles; decimations
                                         # beyond 10 with short filters (off the direct form's one-thread-per-output fallback)
                                         (127, 20, False), (31, 32, True), (127, 13, False), (1000, 9, True), (2467, 11, False), (2467, 13, True),
                                         (3599, 6, False), (4799, 8, True), (5000, 15, False), (2279, 3, True), (3039, 4, False), (31, 11, True),
                                         # decimations D * sub on the pruned tile of D = 16 / 8 / 4 (every sub-th kept sample stored)
                                         (255, 32, True), (401, 12, False), (1000, 48, True), (127, 64, False), (64, 20, True), (300, 24, True), (500, 1024, True)])
def test_fir_decimating_both_paths(rr, monkeypatch, path, L, deci, cplx):
    """Decimating FirFilter through the direct-form kernel and through the overlap-save tiles with a decimating
    store: same protocol, same outputs (1e-5), incl. decimations beyond the tile's useful width and small rings."""
    if path == "direct":
        knob(rr, monkeypatch, fir_path="direct")
    elif path == "fft":                               # overlap-save tiles with a decimating store
        knob(rr, monkeypatch, fir_path="fft")
        knob(rr, monkeypatch, fir_prune=-1)
        knob(rr, monkeypatch, fir_half=-1)
        knob(rr, monkeypatch, fir_poly=-1)
    elif path == "prune":                             # deci 4 / 8 / 16: pruned inverse transform (else as "auto")
        knob(rr, monkeypatch, fir_prune=1)
        knob(rr, monkeypatch, fir_poly=-1)
    elif path == "half":                              # even deci, <= 1025 taps: half-size inverse on 2048-point tiles
        knob(rr, monkeypatch, fir_path="fft")
        knob(rr, monkeypatch, fir_prune=-1)
        knob(rr, monkeypatch, fir_poly=-1)
    elif path == "poly":                              # decimate-first tiles wherever the kernel exists (deci 2..8, 10, 12, 16)
        knob(rr, monkeypatch, fir_poly=1)
    if path == "direct" and L >= 5000:
        pytest.skip("direct-form fallback at thousands of taps: covered by test_fir_complex history, slow")
    x = rnd_c(120000, L * 3 + deci)
    taps = rnd_c(L, L + deci) / max(1, L // 8)
    if not cplx:
        taps = taps.real.astype(np.complex64)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x[:40000], stream_bytes=8 * (L + deci + 1700))
    if path == "fft":
        assert rr.fir_uses_fft_tiles(rr.FirFilter(taps, deci=deci))


def test_fir_complex_chunked(rr):
    x = rnd_c(50000, 3)
    taps = orc.low_pass_complex(10e6, 1e6, 190e3)
    both(rr, lambda m: [m.FirFilter(taps)], x, stream_bytes=8 * 3000)
    both(rr, lambda m: [m.FirFilter(taps, deci=4)], x, stream_bytes=8 * 1111)


@pytest.mark.parametrize("L,deci", [(1, 1), (65, 1), (128, 3), (463, 6)])
def test_fir_float(rr, L, deci):
    x = rnd_f(70000, L)
    taps = rnd_f(L, L + 1) / max(1, L // 8)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x)


def test_fir_cfg1_tone_and_noise(rr):
    # BASELINE configs[0]: 127 real taps, 1M samples
    taps = orc.low_pass_complex(10e6, 1e6, 190e3)
    assert len(taps) == 127
    n = 1_000_000
    ph = (np.arange(n, dtype=np.float32) * np.float32(0.013))
    tone = (np.sin(ph) + 1j * np.cos(ph)).astype(np.complex64)   # rustradio-ui/tests/fir_filter_bench.rs:40-45
    for x in (tone, rnd_c(n, 0x5EED0001)):
        yo = run_chain([orc.FirFilter(taps)], x)
        yg = run_chain([rr.FirFilter(taps)], x)
        assert len(yo) == len(yg) == 999_874
        assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("mode", ["replay", "model"])
def test_fir_translate(rr, mode):
    x = rnd_c(40000, 21)
    taps = orc.low_pass_complex(100e6, 5e6, 943e3 * 4)
    rot = rr.ROT_REPLAY if mode == "replay" else rr.ROT_MODEL
    for deci, f in ((1, 1.7e6), (8, -12.5e6), (3, 0.0)):
        lo, lg = [], []
        yo = run_chain([orc.FirFilter(taps, deci=deci, translate=(100e6, f))], x, log=lo, stream_bytes=8 * 9000)
        yg = run_chain([rr.FirFilter(taps, deci=deci, translate=(100e6, f), rotator=rot)], x, log=lg, stream_bytes=8 * 9000)
        assert lo == lg and len(yo) == len(yg)
        assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("deci,f", [(1, 2.1e6), (2, -7e6), (4, 11e6), (8, -12.5e6), (16, 3e6), (6, 1e6)])
def test_fir_translate_on_tiles(rr, deci, f):
    """.translate() (pre-rotated taps + output rotator, fir.rs:430-473) on the FFT-tile paths: 255 taps at deci 1 (plain
    tiles), 2 (half-size inverse), 6 (decimate-first tiles), 4 / 8 / 16 (pruned inverse)."""
    x = rnd_c(60000, 5 + deci)
    taps = orc.low_pass_complex(100e6, 5e6, 943e3)
    assert len(taps) == 255
    lo, lg = [], []
    yo = run_chain([orc.FirFilter(taps, deci=deci, translate=(100e6, f))], x, log=lo, stream_bytes=8 * 9000)
    # (forced: left to itself the block runs windows this small on the direct form, whatever the filter)
    with rr.build_options(**({"fir_prune": 1} if deci in (4, 8, 16) else {"fir_poly": 1} if deci == 6 else {"fir_path": "fft"})):
        blk = rr.FirFilter(taps, deci=deci, translate=(100e6, f), rotator=rr.ROT_REPLAY)
    assert rr.fir_uses_fft_tiles(blk)
    yg = run_chain([blk], x, log=lg, stream_bytes=8 * 9000)
    assert lo == lg and len(yo) == len(yg)
    assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("L", [1, 2, 5, 193, 401, 463, 512, 513, 1025, 2467, 3100, 3300, 5000, 8193, 12000, 15292, 15293, 16383])
def test_fftfilter(rr, L):
    n = 200_000 if L < 1025 else 400_000
    x = rnd_c(n, L)
    taps = rnd_c(L, 300 + L) / max(1, L // 4)
    e = both(rr, lambda m: [m.FftFilter(taps)], x)
    f_ref, s_ref, _ = rr.fftfilter_dims(rr.FftFilter(taps))
    assert (f_ref, s_ref) == orc.fftfilter_dims(orc.FftFilter(taps))


@pytest.mark.parametrize("L", [2467, 3300])
@pytest.mark.parametrize("forced", [13, 14, 0])
def test_long_filter_split_and_alternate_tiles(rr, monkeypatch, L, forced):
    """Filters of 2500-3500 taps run on 8192-point split tiles for large windows and on plain 4096-point tiles for windows of
    too few split tiles to fill the chip (chosen per call).  forced = 13 / 14: the split tiles on these small inputs too
    (rr_build_opts.fft_log2f); 0: the block's own choice (the alternate tile here).  FftFilter and the fused chains, c32 and u8."""
    if forced:
        knob(rr, monkeypatch, fft_log2f=forced)
    taps = rnd_c(L, 17 + L) / (L // 4)
    x = rnd_c(300_000, L)
    both(rr, lambda m: [m.FftFilter(taps)], x)
    both(rr, lambda m: [m.FftFilter(taps)], x[:120_000], stream_bytes=8 * (8192 - L + 3000))
    z = fm_signal(200_000, 1.024e6, 0.0, L)
    for I, D in ((25, 128), (1, 5)):
        knob(rr, monkeypatch, fm_poly=-1)                  # (1:5 would otherwise take the decimate-first tiles)
        yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], z)
        ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], z)
        yg = run_chain([rr.FmChain(taps, I, D, 1.0)], z)
        _demod_close(yg, yo, ro)
    b = np.empty(2 * len(z), np.uint8)
    b[0::2] = np.clip(np.round(z.real / 0.008 + 127), 0, 255)
    b[1::2] = np.clip(np.round(z.imag / 0.008 + 127), 0, 255)
    yo = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(25, 128), orc.QuadratureDemod(1.0)], b)
    ro = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(25, 128)], b)
    _demod_close(run_chain([rr.FmChainU8(taps, 25, 128, 1.0)], b), yo, ro)


def test_fftfilter_chunked_and_small_outputs(rr):
    taps = orc.low_pass_complex(10e6, 1e6, 60e3)
    x = rnd_c(150_000, 77)
    both(rr, lambda m: [m.FftFilter(taps)], x, stream_bytes=8 * 700)      # one block per call
    both(rr, lambda m: [m.FftFilter(taps)], x, stream_bytes=8 * 5000)
    # direct protocol probes
    for impl in (orc, rr):
        b = impl.FftFilter(taps)
        assert b.work(x[:100], 622)[:4] == (WAIT_DST, 0, 0, 623)
        assert b.work(x[:100], 10_000)[:4] == (WAIT_SRC, 100, 0, 523)
        st, c, p, need, out = b.work(x[100:5000], 1300)
        assert (st, c, p, need) == (WAIT_DST, 2 * 623 - 100, 2 * 623, 623)


def test_fftfilter_rejects_too_many_taps(rr):
    with pytest.raises(ValueError):
        rr.FftFilter(np.ones((1 << 19) + 1, np.complex64))  # frames of 2^m >= 2 L points beyond the any-size transform's 2^20
    with pytest.raises(ValueError):
        rr.FftFilter(np.ones(0, np.complex64))
    # (the fused chains have no limit of their own any more: beyond their tiles the constructor composes the three blocks —
    #  test_no_constructor_cliffs_*; what is left is FftFilter's own 2^19 taps and the reference's argument errors)
    assert rr.FmChain(np.ones(16384, np.complex64), 1, 6, 1.0) is not None
    with pytest.raises(ValueError):
        rr.FmChain(np.ones((1 << 19) + 1, np.complex64), 1, 6, 1.0)
    with pytest.raises(ValueError):
        rr.FmChain(np.ones(100, np.complex64), 1, 0, 1.0)     # "RationalResampler created using deci 0"


@pytest.mark.parametrize("inner", ["real", "complex"])
def test_fftfilter_float(rr, monkeypatch, inner):
    """FftFilterFloat on the real-stream tile kernel (two overlap-save segments per Complex tile) and on the
    f32 -> Complex -> FftFilter -> .re path it replaces (kept for filters beyond 3584 taps)."""
    if inner == "complex":
        knob(rr, monkeypatch, fftfloat_complex=1)
    x = rnd_f(300_000, 5)
    taps = orc.low_pass(200e3, 44.1e3, 500.0)
    both(rr, lambda m: [m.FftFilterFloat(taps)], x)
    both(rr, lambda m: [m.FftFilterFloat(taps)], x, stream_bytes=4 * 30_000)


@pytest.mark.parametrize("L", [1, 2, 7, 64, 401, 1024, 1025, 2500, 3584, 3585, 6000])
def test_fftfilter_float_lengths(rr, L):
    x = rnd_f(250_000, L)
    taps = rnd_f(L, L + 77) / max(1, L // 4)
    both(rr, lambda m: [m.FftFilterFloat(taps)], x)
    both(rr, lambda m: [m.FftFilterFloat(taps)], x[:120_000], stream_bytes=4 * 50_000)


@pytest.mark.parametrize("path", ["direct", "fft", "auto", "prune"])
@pytest.mark.parametrize("L,deci", [(1, 1), (5, 1), (39, 1), (40, 1), (65, 1), (463, 1), (463, 6), (128, 3), (1000, 16),
                                    (3584, 1), (3584, 4096), (64, 100), (330, 50), (2000, 7), (3, 4), (64, 4), (513, 4),
                                    (255, 8), (1025, 8), (2049, 16), (31, 16),
                                    # round 4: beyond the real-stream tiles (3584 taps) on the FirFilter<Complex> kernels; /20, /32 short
                                    (3585, 1), (5000, 1), (5000, 20), (9000, 3), (4000, 32), (127, 20), (31, 32),
                                    (255, 32), (401, 12), (1000, 48), (127, 64), (64, 20), (300, 24)])
def test_fir_float_both_paths(rr, monkeypatch, path, L, deci):
    """FirFilter<Float> through the direct-form kernel, the real-stream overlap-save tiles (decimating store) and —
    deci 4 / 8 / 16 — the tiles with the pruned inverse transform."""
    if path == "direct":
        knob(rr, monkeypatch, fir_path="direct")
    elif path == "fft":
        knob(rr, monkeypatch, fir_path="fft")
        knob(rr, monkeypatch, fir_prune=-1)
    elif path == "prune":
        knob(rr, monkeypatch, fir_prune=1)
    x = rnd_f(150_000, L * 3 + deci)
    taps = rnd_f(L, L + deci) / max(1, L // 8)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x)
    both(rr, lambda m: [m.FirFilter(taps, deci=deci)], x[:50_000], stream_bytes=4 * (L + deci + 2100))


@pytest.mark.parametrize("I,D", [(1, 1), (1, 6), (25, 128), (3, 2), (200000, 1024000), (48, 200), (7, 3), (1, 1000)])
@pytest.mark.parametrize("dtype", [np.complex64, np.uint32, np.uint8, np.uint16])
def test_resampler_bit_exact(rr, I, D, dtype):
    n = 100_003
    if np.dtype(dtype) == np.complex64:
        x = rnd_c(n, I + D)
    else:
        x = (np.arange(n) * 2654435761 % (1 << 32)).astype(np.uint64).astype(dtype)
    both(rr, lambda m: [m.RationalResampler(I, D, dtype)], x, exact=True)
    both(rr, lambda m: [m.RationalResampler(I, D, dtype)], x, exact=True, stream_bytes=np.dtype(dtype).itemsize * 4001)


def test_quaddemod_exact_atan2_accuracy(rr):
    """The exact flavour's atan2 against f64 atan2 over all octants, axes, equal magnitudes, zeros, tiny and huge
    magnitudes: <= 1e-6 rad; libm's special values (signed zeros, infinities, NaN)."""
    rng = np.random.default_rng(99)
    ang = np.concatenate([np.linspace(-np.pi, np.pi, 20001), np.arange(-8, 9) * np.pi / 4, rng.uniform(-np.pi, np.pi, 20000)])
    mag = np.concatenate([np.ones(20001), np.ones(17), 10.0 ** rng.uniform(-15, 15, 20000)])
    z = mag * np.exp(1j * ang)
    # x[2i] = 1, x[2i+1] = z_i  ->  conj(1) * z = z at the even outputs
    x = np.ones(2 * len(z), np.complex64)
    x[1::2] = z.astype(np.complex64)
    st, c, p, need, out = rr.QuadratureDemod(1.0, rr.ATAN2_EXACT).work(x, len(x))
    got = out[0::2][:len(z)].astype(np.float64)
    zz = x[1::2].astype(np.complex128)
    ref = np.arctan2(zz.imag, zz.real)
    d = np.abs(got - ref)
    d = np.minimum(d, 2 * np.pi - d)
    assert np.max(d) <= 1e-6, float(np.max(d))
    # special values (the oracle is libm's atan2f)
    sp = np.array([1, 0, 1, -1, 1, complex(0.0, 0.0), 1, complex(-1.0, 0.0), 1, complex(-1.0, -0.0), 1, complex(np.inf, np.inf),
                   1, complex(-np.inf, np.inf), 1, complex(np.nan, 1.0), 1, complex(1e-40, 1e-40), 1, complex(3e38, -3e38)], np.complex64)
    yo = orc.QuadratureDemod(1.0, orc.ATAN2_EXACT if hasattr(orc, "ATAN2_EXACT") else 0).work(sp, len(sp))[4]
    yg = rr.QuadratureDemod(1.0, rr.ATAN2_EXACT).work(sp, len(sp))[4]
    assert len(yo) == len(yg)
    for a, b in zip(yo, yg):
        assert (np.isnan(a) and np.isnan(b)) or abs(float(a) - float(b)) <= 1e-6, (a, b)


@pytest.mark.parametrize("mode", [0, 1])
def test_quaddemod(rr, mode):
    x = rnd_c(300_000, 8)
    x[100:110] = 0          # zero runs: atan2(0,0) must be exactly 0 (quadrature_demod.rs:211-219)
    gain = 0.7
    both(rr, lambda m: [m.QuadratureDemod(gain, mode)], x, scale=np.pi * gain)
    yg = run_chain([rr.QuadratureDemod(gain, mode)], x)
    assert np.all(yg[100:109] == 0.0)


@pytest.mark.parametrize("L,w", [(65, 0), (3, 0), (129, 1), (255, 2), (199, 0), (201, 0), (511, 1), (1001, 0), (2001, 2), (3583, 0), (3585, 0), (4001, 0), (9001, 1)])
def test_hilbert(rr, L, w):
    """(round 4: 201 ... 3583 taps run on real-stream overlap-save tiles with a Complex store, longer ones as the Complex
    filter delta + i h on Complex(x, 0); both only from windows of a few tiles on, so the small rings stay on the pair kernel)"""
    x = rnd_f(200_000, L)
    both(rr, lambda m: [m.Hilbert(L, w)], x)
    both(rr, lambda m: [m.Hilbert(L, w)], x, stream_bytes=4 * max(7777, 2 * L + 100))
    if 200 < L < 3584:                                 # on the tiles the real part is still a copy of the input: bit-exact
        yo = run_chain([orc.Hilbert(L, w)], x)
        yg = run_chain([rr.Hilbert(L, w)], x)
        assert np.array_equal(yg.real, yo.real)


@pytest.mark.parametrize("L", [63, 65, 31, 5])
@pytest.mark.parametrize("off", [0, 1, 2, 3])
def test_hilbert_device_windows_any_alignment(rr, L, off):
    """The Hilbert kernel reads PAIRS of input floats with 8-byte loads at any 4-byte alignment: device
    windows starting at every float offset, for both tap parities (L/2 even and odd), full-size tiles."""
    import torch
    n = 300_000
    x = rnd_f(n + 8, L + off)
    yo = run_chain([orc.Hilbert(L, 0)], x[off:off + n], stream_bytes=4 * n)
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros(2 * n + 64, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    b = rr.Hilbert(L, 0)
    st, c, p, need = b.work_dev(dx.data_ptr() + 4 * off, n, dy.data_ptr(), n)
    b.sync()
    assert p == len(yo) and p > 0
    yg = dy.cpu().numpy().view(np.complex64)[:p]
    assert np.array_equal(yg.real, yo.real)          # the real part is a copy: bit-exact
    assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 4096, 1_000_001, 3_999_999])
def test_rtlsdr_decode_bit_exact(rr, n):
    x = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8)
    both(rr, lambda m: [m.RtlSdrDecode()], x, exact=True)
    both(rr, lambda m: [m.RtlSdrDecode()], x, stream_bytes=8 * 4099, exact=True)


@pytest.mark.parametrize("off", [0, 1, 2, 3, 5])
def test_rtlsdr_decode_device_windows_any_alignment(rr, off):
    import torch
    n = 2_000_003
    x = np.random.default_rng(off).integers(0, 256, n + 8, dtype=np.uint8)
    dx = torch.from_numpy(x).cuda()
    dy = torch.zeros(n + 16, dtype=torch.float32, device="cuda")        # n/2 Complex
    b = rr.RtlSdrDecode()
    for oo in (0, 8):                                                   # 16- and 8-byte aligned outputs
        st, c, p, need = b.work_dev(dx.data_ptr() + off, n, dy.data_ptr() + oo, n // 2)
        b.sync()
        assert (st, c, p, need) == (WAIT_SRC, n - 1, n // 2, 2)
        yg = dy.cpu().numpy()[oo // 4: oo // 4 + 2 * p]
        yo = ((x[off:off + n - 1].astype(np.float32) - np.float32(127.0)) * np.float32(0.008))
        assert np.array_equal(yg, yo)


@pytest.mark.parametrize("stream_bytes", [4_096_000, 4 * 1_003])
def test_sync_blocks_bit_exact(rr, stream_bytes):
    """MultiplyConst<Float|Complex> and FastFM: same protocol log, bit-identical samples"""
    xf, xc = rnd_f(100_001, 4), rnd_c(100_001, 5)
    both(rr, lambda m: [m.MultiplyConst(0.37)], xf, stream_bytes=stream_bytes, exact=True)
    both(rr, lambda m: [m.MultiplyConst(0.6 - 1.25j, np.complex64)], xc, stream_bytes=max(stream_bytes, 8 * 1003), exact=True)
    both(rr, lambda m: [m.FastFM()], xc, stream_bytes=max(stream_bytes, 8 * 1003), exact=True)
    # the rtl_fm audio stage: demod -> FftFilterFloat -> RationalResampler -> MultiplyConst (examples/rtl_fm.rs:398-418)
    at = orc.low_pass(200e3, 44.1e3, 5e3)
    both(rr, lambda m: [m.QuadratureDemod(1.0), m.FftFilterFloat(at), m.RationalResampler(441, 2000, np.float32), m.MultiplyConst(0.5)],
         xc, stream_bytes=4_096_000)


@pytest.mark.parametrize("size", [2, 4, 8, 64, 256, 512, 1024, 2048, 4096, 8192, 16384,
                                  3, 5, 12, 100, 257, 511, 513, 1000, 1023, 1025, 1500, 2047])
def test_fftstream(rr, size):
    """FftStream: whole frames only, forward unnormalised transform in natural bin order; sizes that are not a power
    of two (the reference plans any size with rustfft) run as Bluestein chirp-z convolutions on the filter tiles"""
    n = max(5 * size + 3, 40_000)
    x = rnd_c(n, size)
    e = both(rr, lambda m: [m.FftStream(size)], x)
    both(rr, lambda m: [m.FftStream(size)], x[:3 * size + 1], stream_bytes=8 * (size + size // 2 + 1))
    st, c, p, need, out = rr.FftStream(size).work(x[:2 * size], 4 * size)
    ref = np.fft.fft(x[:2 * size].astype(np.complex128).reshape(2, size), axis=1).reshape(-1)
    assert (st, c, p) == (AGAIN, 2 * size, 2 * size)
    assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) <= TOL
    assert np.array_equal(rr.FftStream(size).work(np.zeros(size, np.complex64), size)[4], np.zeros(size, np.complex64))


@pytest.mark.parametrize("size", [2049, 3000, 4097, 8191, 12000, 16385, 32768, 65536, 100_000, 131_072, 262_144, 500_001, 512_000])
def test_fftstream_any_size(rr, size):
    """FftStream sizes beyond one LDS tile (rustfft plans any size, fft_stream.rs:43-44): the four-step decomposition for
    powers of two above 16384, Bluestein on a power of two M >= 2 size - 1 for everything else above 2048 — against
    numpy's f64 FFT (the oracle's defining sum is O(n^2)), whole frames only, work() protocol of the reference."""
    nfr = 2 if size <= 200_000 else 1
    x = rnd_c(nfr * size + 5, size % 1000)
    b = rr.FftStream(size)
    st, c, p, need, out = b.work(x, 512_000)
    assert (st, c, p) == (AGAIN, nfr * size, nfr * size)
    ref = np.fft.fft(x[:nfr * size].astype(np.complex128).reshape(nfr, size), axis=1).reshape(-1)
    assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) <= TOL
    assert b.work(x[:size - 1], 512_000)[:4] == (WAIT_SRC, 0, 0, size)
    assert b.work(x, size - 1)[:4] == (WAIT_DST, 0, 0, size)


def test_fft_message_block(rr):
    """Fft (src/fft.rs:19-56), the PDU form: one message in, its transform out; the reference's `zeroes` and
    `rejects_wrong_size` tests, plus random messages of tile, four-step and Bluestein sizes against numpy."""
    f = rr.Fft(1024)
    assert np.array_equal(f.process(np.zeros(1024, np.complex64)), np.zeros(1024, np.complex64))      # fft.rs:65-85
    with pytest.raises(ValueError, match="FFT expected 4 samples, got 3"):                            # fft.rs:87-104
        rr.Fft(4).process(np.zeros(3, np.complex64))
    with pytest.raises(ValueError):
        rr.Fft(0)
    for size in (4, 1000, 1024, 5000, 32768):
        x = rnd_c(size, size)
        ref = np.fft.fft(x.astype(np.complex128))
        assert np.max(np.abs(rr.Fft(size).process(x) - ref)) / np.max(np.abs(ref)) <= TOL


@pytest.mark.parametrize("L", [16384, 20_000, 40_000, 100_000])
def test_fftfilter_beyond_16383_taps(rr, L):
    """The reference has no limit on FftFilter's taps (fft_filter.rs:36-42): beyond the largest LDS-resident tile the
    block runs overlap-save frames of 2^m >= 2 L points through the any-size transform; same work() protocol and outputs."""
    taps = rnd_c(L, L) / (L // 8)
    x = rnd_c(2 * (2 * (1 << int(np.ceil(np.log2(L)))) - L) + 1000, 3)
    both(rr, lambda m: [m.FftFilter(taps)], x)
    if L == 20_000:                                   # rings barely larger than one block of 45,536 samples: one frame per call
        both(rr, lambda m: [m.FftFilter(taps)], x, stream_bytes=8 * 50_000)


def test_fftfilter_float_beyond_16383_taps(rr):
    L = 20_000
    taps = (rnd_c(L, 1).real / (L // 8)).astype(np.float32)
    x = rnd_c(2 * (65536 - L) + 1000, 4).real.astype(np.float32)
    both(rr, lambda m: [m.FftFilterFloat(taps)], x)


def test_fftstream_rejects(rr):
    for bad in (0, 1, 512_001):
        with pytest.raises(Exception):
            rr.FftStream(bad)
    b = rr.FftStream(1024)
    x = rnd_c(2000, 1)
    assert b.work(x[:1023], 4096)[:4] == (WAIT_SRC, 0, 0, 1024)
    assert b.work(x, 1023)[:4] == (WAIT_DST, 0, 0, 1024)


def test_hilbert_rejects_even(rr):
    for n in (0, 1, 2, 64):
        with pytest.raises(ValueError):
            rr.Hilbert(n)


def fm_signal(n, fs, f_center, seed):
    t = np.arange(n, dtype=np.float64)
    phi = 2 * np.pi * np.cumsum(f_center + 75e3 * np.sin(2 * np.pi * 1e3 * t / fs)) / fs
    r = np.random.default_rng(seed)
    return (np.exp(1j * phi) + 0.01 * (r.standard_normal(n) + 1j * r.standard_normal(n))).astype(np.complex64)


def test_fm_chain_cfg3(rr):
    """BASELINE configs[2]: FftFilter(463 taps) -> RationalResampler(1:6) -> QuadratureDemod @2.4 Msps.
    atan2 is ill-conditioned where the filtered magnitude is tiny (filter start-up transient,
    stop-band signal), so the end-to-end bound is the filter stage's own parity bound propagated
    through atan2:  |d angle| <= 1e-5 pi + eps/|r[m]| + eps/|r[m+1]|,  eps = 1e-5 max|r|;
    every stage is also checked alone at 1e-5, and for a station centred in the channel
    (|r| ~ 1 after the transient) the plain 1e-5 pi bound must hold end to end.
    (a) centred station; (b) SURVEY §8d's signal, 150 kHz off centre (mostly stop band)."""
    fs = 2.4e6
    n = 1_200_000
    taps = orc.low_pass_complex(fs, 100e3, 12.5e3)
    assert len(taps) == 463

    def chain(m):
        return [m.FftFilter(taps), m.RationalResampler(1, 6), m.QuadratureDemod(1.0)]

    for f_center in (0.0, 150e3):
        x = fm_signal(n, fs, f_center, 0x5EED0003)
        lo, lg = [], []
        yo = run_chain(chain(orc), x, log=lo)
        yg = run_chain(chain(rr), x, log=lg)
        assert lo == lg and len(yo) == len(yg) == -(-((n // 561) * 561) // 6) - 1
        ro = run_chain(chain(orc)[:2], x)
        rg = run_chain(chain(rr)[:2], x)
        assert max_norm_err(rg, ro) <= TOL                       # filter + resampler stages alone
        skip = len(taps) // 6 + 2                                 # start-up transient of the filter
        r = angle_parity(yg, yo, ro, TOL, skip)
        print(f"fm chain, station {f_center / 1e3:.0f} kHz off centre: {r['used']:.3f} of the propagated allowance used, "
              f"{100 * r['above_plain']:.4f} % of the samples above the plain 1e-5 pi, largest error {r['max_err_pi']:.2e} pi")
        assert r["used"] <= 1.0, r
        if f_center == 0.0:                                       # centred station: the plain bound, every sample
            assert r["above_plain"] == 0.0, r
        both(rr, lambda m: [m.QuadratureDemod(1.0)], ro, scale=np.pi)   # demod stage alone


def test_channelizer_cfg5(rr):
    """BASELINE configs[4]: Hilbert(65) -> FirFilter(255 taps, deci 8)."""
    x = rnd_f(800_000, 0x5EED0005)
    taps = orc.low_pass_complex(100e6, 5e6, 943e3)
    assert len(taps) == 255
    both(rr, lambda m: [m.Hilbert(65), m.FirFilter(taps, deci=8)], x)


@pytest.mark.parametrize("hn,L,deci,cplx,tr", [(65, 255, 8, False, None), (65, 255, 8, False, (100e6, 7e6)), (31, 64, 5, True, None),
                                               (129, 33, 1, False, None), (3, 1, 1, False, None), (63, 100, 12, True, (8.0, 2.0)),
                                               (65, 401, 16, False, None), (33, 90, 4, True, None), (65, 900, 8, True, None),
                                               (7, 2, 16, True, (8.0, 1.0)), (65, 1900, 16, False, None),
                                               (65, 1400, 8, True, None), (65, 700, 4, False, None), (129, 2900, 16, True, None),
                                               (65, 255, 32, True, None), (65, 401, 24, False, (100e6, 3e6)), (33, 127, 12, True, None), (65, 600, 64, True, None)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 4 * 5_003])
@pytest.mark.parametrize("prune", ["1", "0"])
def test_hilbert_fir_fused_block(rr, monkeypatch, hn, L, deci, cplx, tr, stream_bytes, prune):
    """rr.HilbertFir (one composite decimating FIR on the real input) == Hilbert -> FirFilter<Complex> of the
    oracle, whole stream, any chunking; with and without .translate().  prune = 1: decimations 4 / 8 / 16 run on
    real-stream overlap-save tiles with the pruned inverse transform (k_fftfilt_prune), 0: direct form."""
    knob(rr, monkeypatch, fir_prune=1 if prune == "1" else -1)
    x = rnd_f(300_000, hn * 1000 + L + deci)
    if L == 255:
        taps = orc.low_pass_complex(100e6, 5e6, 943e3)
    else:
        taps = rnd_c(L, L + 3) / max(1, L // 8)
        if not cplx:
            taps = taps.real.astype(np.complex64)
    kw = {"translate": tr} if tr else {}
    yo = run_chain([orc.Hilbert(hn), orc.FirFilter(taps, deci=deci, **kw)], x)
    # (the reference's rotator is a drifting f32 recurrence, fir.rs:465 TODO: replay it for long streams)
    kg = dict(kw, rotator=rr.ROT_REPLAY) if tr else {}
    yg = run_chain([rr.HilbertFir(hn, taps, deci, **kg)], x, stream_bytes=max(stream_bytes, 4 * (L + deci + 8)))
    assert len(yg) == len(yo) and len(yo) > 0
    assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("hn,L,deci,tr", [(65, 255, 1, None), (65, 255, 2, (100e6, 7e6)), (65, 1000, 5, None), (33, 127, 1, None),
                                          (65, 2467, 32, None), (129, 2467, 6, (8.0, 1.0)), (65, 1000, 20, None), (65, 600, 3, None)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 4 * 30_000])
def test_hilbert_fir_two_stage_shapes(rr, hn, L, deci, tr, stream_bytes):
    """Round 4: shapes where the composite direct form loses (long filters at small decimations, or no tile at all) run the
    two stages through an analytic buffer behind the same handle: Hilbert's kernel, then the FirFilter's own path selection.
    Same protocol, same stream as Hilbert -> FirFilter<Complex> of the oracle; chunked windows included."""
    x = rnd_f(400_000, hn + L + deci)
    taps = rnd_c(L, L + 5) / max(1, L // 8)
    kw = {"translate": tr} if tr else {}
    lo, lg = [], []
    yo = run_chain([orc.Hilbert(hn), orc.FirFilter(taps, deci=deci, **kw)], x)
    yg = run_chain([rr.HilbertFir(hn, taps, deci, **kw)], x, stream_bytes=max(stream_bytes, 4 * (L + deci + 8)))
    assert len(yg) == len(yo) and len(yo) > 0
    assert max_norm_err(yg, yo) <= TOL


def test_hilbert_fir_protocol(rr):
    """work() of the fused block follows FirFilter's protocol on the real input (fir.rs:496-549)."""
    taps = orc.low_pass_complex(100e6, 5e6, 943e3)
    b = rr.HilbertFir(65, taps, 8)
    x = rnd_f(5000, 1)
    assert b.work(x[:261], 100)[:4] == (WAIT_SRC, 0, 0, 262)
    assert b.work(x[:262], 0)[:4] == (WAIT_DST, 0, 0, 1)
    st, c, p, need, out = b.work(x[:1000], 100)
    assert (st, c, p) == (AGAIN, 8 * ((1000 - 254) // 8), (1000 - 254) // 8)
    st, c, p, need, out = b.work(x[c:], 7)
    assert (st, c, p) == (AGAIN, 56, 7)
    with pytest.raises(Exception):
        rr.HilbertFir(64, taps, 8)


def test_device_window_api(rr):
    """rr_block_work_dev: device-resident windows (torch tensors only provide the memory)."""
    import torch
    taps = orc.low_pass_complex(10e6, 1e6, 60e3)
    x = rnd_c(1_000_000, 9)
    yo = run_chain([orc.FftFilter(taps)], x, stream_bytes=8 * len(x))
    dx = torch.from_numpy(x.view(np.float32)).cuda()
    dy = torch.zeros(2 * len(x) + 4096, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    b = rr.FftFilter(taps)
    st, c, p, need = b.work_dev(dx.data_ptr(), len(x), dy.data_ptr(), len(x) + 2048)
    b.sync()
    assert st == WAIT_SRC and c == len(x) and p == len(yo)
    yg = dy.cpu().numpy().view(np.complex64)[:p]
    assert max_norm_err(yg, yo) <= TOL


def _demod_close(yg, yo, ro):
    """conditioning-aware end-to-end bound (see test_fm_chain_cfg3)."""
    assert len(yg) == len(yo)
    eps = TOL * float(np.max(np.abs(ro)))
    mag = np.abs(ro.astype(np.complex128))
    bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
    d = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
    d = np.minimum(d, 2 * np.pi - d)
    assert np.all(d <= bound[:len(d)]), float(np.max(d - bound[:len(d)]))
    return d


@pytest.mark.parametrize("L,I,D", [(463, 1, 6), (401, 1, 1), (127, 25, 128), (33, 3, 2), (463, 200000, 1024000), (5, 1, 40),
                                   (463, 1, 2), (400, 1, 6), (600, 1, 100), (463, 3, 18), (463, 1, 3), (461, 1, 5), (200, 1, 7),
                                   (463, 1, 8), (464, 1, 12), (1000, 1, 16), (462, 1, 10), (2467, 1, 6), (1, 1, 4), (7, 1, 6)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 20_000])
@pytest.mark.parametrize("kernel", ["poly", "half", "full"])
def test_fm_chain_fused_block(rr, monkeypatch, L, I, D, stream_bytes, kernel):
    """rr.FmChain (one fused kernel) == FftFilter -> RationalResampler -> QuadratureDemod of the oracle,
    for whole-stream output and any chunking.  kernel = poly (the default): integer decimations run on decimate-first
    tiles (k_fm_chain_poly: D phase transforms + one inverse per 1024 output-rate positions); half: reduced ratios 1:even
    on 2048-point tiles finish each tile with a folded 1024-point inverse on one wave (k_fm_chain_half); full: the
    full-size inverse everywhere."""
    if kernel == "full":
        knob(rr, monkeypatch, fm_full=1)
    elif kernel == "half":
        knob(rr, monkeypatch, fm_poly=-1)
    else:
        knob(rr, monkeypatch, fm_poly=1)          # (decimations beyond 6 are otherwise left to the other kernels)
    fs = 2.4e6
    n = 400_000
    x = fm_signal(n, fs, 0.0, 77 + L)
    if L == 463:
        taps = orc.low_pass_complex(fs, 100e3, 12.5e3)
    else:
        taps = (rnd_c(L, L) / max(1, L // 4)).astype(np.complex64)
    gain = 0.9
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(gain)], x, stream_bytes=stream_bytes)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], x, stream_bytes=stream_bytes)
    yg = run_chain([rr.FmChain(taps, I, D, gain)], x, stream_bytes=stream_bytes)
    d = _demod_close(yg / gain, yo / gain, ro)
    if L == 463 and (I, D) == (1, 6):
        assert np.max(d[len(taps) // 6 + 2:]) <= TOL * np.pi


@pytest.mark.parametrize("L,I,D", [(463, 1, 6), (127, 25, 128), (33, 3, 2), (2467, 1, 5), (5000, 1, 4)])
@pytest.mark.parametrize("stream_bytes,odd", [(4_096_000, False), (8 * 20_000, False), (30_001, True)])
def test_fm_chain_u8_fused_block(rr, L, I, D, stream_bytes, odd):
    """rr.FmChainU8 (RtlSdrDecode fused into the chain kernel) == RtlSdrDecode -> FftFilter ->
    RationalResampler -> QuadratureDemod of the oracle on an RTL-SDR byte stream, any chunking, odd
    total length (the trailing byte is never consumed)."""
    fs = 2.4e6
    n = 300_000
    z = fm_signal(n, fs, 0.0, 5 + L)
    b = np.empty(2 * n + (1 if odd else 0), np.uint8)                  # what an RTL-SDR would deliver
    b[0:2 * n:2] = np.clip(np.round(z.real / 0.008 + 127), 0, 255)
    b[1:2 * n:2] = np.clip(np.round(z.imag / 0.008 + 127), 0, 255)
    if odd:
        b[-1] = 200
    if L == 463:
        taps = orc.low_pass_complex(fs, 100e3, 12.5e3)
    else:
        taps = (rnd_c(L, L) / max(1, L // 4)).astype(np.complex64)
    gain = 0.9
    # the oracle's FftFilter ring holds Complex: give it the default stream size, chunk the byte side
    yo = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(gain)], b)
    ro = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(I, D)], b)
    yg = run_chain([rr.FmChainU8(taps, I, D, gain)], b, stream_bytes=stream_bytes)
    _demod_close(yg / gain, yo / gain, ro)


def test_fm_chain_u8_odd_device_pointer(rr):
    """an odd-addressed byte window takes the out-of-line decode path: same output"""
    import torch
    fs, n = 2.4e6, 200_000
    taps = orc.low_pass_complex(fs, 100e3, 12.5e3)
    b = np.random.default_rng(3).integers(0, 256, 2 * n + 1, dtype=np.uint8)
    db = torch.from_numpy(b).cuda()
    outs = []
    for off in (0, 1):
        blk = rr.FmChainU8(taps, 1, 6)
        dy = torch.zeros(n, dtype=torch.float32, device="cuda")
        src = torch.from_numpy(np.ascontiguousarray(b[:2 * n])).cuda() if off == 0 else db
        if off:
            db[1:] = torch.from_numpy(b[:2 * n]).cuda()
        st, c, p, need = blk.work_dev(src.data_ptr() + off, 2 * n, dy.data_ptr(), n)
        blk.sync()
        assert st == WAIT_SRC and c == 2 * n and p > 0
        outs.append(dy.cpu().numpy()[:p].copy())
    assert np.array_equal(outs[0], outs[1])


def test_fm_chain_fused_protocol(rr):
    taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)       # nsamples 561
    b = rr.FmChain(taps, 1, 6, 1.0)
    x = fm_signal(10_000, 2.4e6, 0.0, 5)
    # an output window smaller than one block's outputs (ceil(561/6) - 1 = 93) does not stop the INPUT side: like the
    # reference chain (FftFilter writes into its own inner stream) the samples join the pending block
    assert b.work(x[:100], 50)[:4] == (WAIT_SRC, 100, 0, 461)
    st, c, p, need, out = b.work(x[100:5000], 1000)
    # 100 + 4900 = 5000 = 8 blocks + 512: N1 = 4488, N2 = 748, out 747
    assert (st, c, p, need) == (WAIT_SRC, 4900, 747, 561 - 512)
    st, c, p, need, out = b.work(x[5000:], 100)
    # 512 + 5000 = 9 blocks (5049) + 463; outputs per block ~93.5 -> only one more block fits in 100
    assert st == WAIT_DST and p <= 100 and c == 561 - 512
    # ... and the window after that is too small for the next block (94 outputs): ONE block goes through the block's
    # output tail, 40 outputs now, the rest as room appears (WAIT_DST need 1, the resampler's `pending` protocol,
    # rational_resampler.rs:162-173), no input consumed while outputs are pending
    pos = 5000 + c
    st, c, p2, need, o1 = b.work(x[pos:], 40)
    assert (st, c, p2, need) == (WAIT_DST, 561, 40, 1)
    assert not b.eof(True)                                        # pending outputs: not EOF even when the source is
    pos += c
    st, c, p3, need, o2 = b.work(x[pos:], 40)
    assert (st, c, p3, need) == (WAIT_DST, 0, 40, 1)
    st, c, p4, need, o3 = b.work(x[pos:pos + 10], 40)             # the tail is flushed, then the input side goes on
    assert (st, c, need) == (WAIT_SRC, 10, 551) and p4 in (13, 14)
    assert b.eof(True)
    with pytest.raises(ValueError):
        rr.FmChain(taps, 0, 6)
    assert rr.FmChain(taps, 1, 100000) is not None              # beyond any tile: composed (test_no_constructor_cliffs_*)


@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 30_000])
@pytest.mark.parametrize("kernel", ["poly", "half", "full"])
def test_fm_multi_shared_source(rr, monkeypatch, stream_bytes, kernel):
    """rr.FmMulti: N channels on one shared input (forward FFT computed once per tile) — every
    channel must equal its own oracle chain FftFilter(taps_c) -> RationalResampler -> QuadratureDemod.
    kernel = half: interp 1 / even deci on 2048-point tiles runs folded 1024-point inverses (k_fm_multi_half)."""
    if kernel == "full":
        knob(rr, monkeypatch, fm_full=1)
    elif kernel == "half":
        knob(rr, monkeypatch, fm_poly=-1)
    fs, n, nch = 2.4e6, 300_000, 5
    proto = orc.low_pass_complex(fs, 100e3, 12.5e3)
    k = np.arange(len(proto), dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * ((c - 2) * 8e3) * k / fs)).astype(np.complex64)
                     for c in range(nch)])
    x = fm_signal(n, fs, 0.0, 123)
    blk = rr.FmMulti(taps, 1, 6, 1.0)
    # drive the multi-output block by hand with the reference's window sizes
    cap_in = stream_bytes // 8
    outs = [[] for _ in range(nch)]
    pos, ring = 0, np.zeros(0, np.complex64)
    while True:
        take = min(cap_in - len(ring), len(x) - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, stream_bytes // 4)
        ring = ring[c:]
        for ch in range(nch):
            outs[ch].append(out[ch])
        if take == 0 and c == 0 and p == 0:
            break
    for ch in range(nch):
        yg = np.concatenate(outs[ch])
        chain = [orc.FftFilter(taps[ch]), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)]
        yo = run_chain(chain, x, stream_bytes=stream_bytes)
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, 6)], x, stream_bytes=stream_bytes)
        d = _demod_close(yg, yo, ro)
        if ch == 2:     # centred channel: |r| ~ 1 after the start-up transient
            assert np.max(d[len(proto) // 6 + 2:]) <= TOL * np.pi


@pytest.mark.parametrize("kernel", ["auto", "half"])
@pytest.mark.parametrize("L,D,nch", [(463, 2, 1), (463, 4, 2), (400, 10, 3), (300, 64, 2), (513, 6, 4), (463, 200, 1), (700, 6, 2),
                                     (463, 3, 2), (461, 5, 3), (300, 7, 2), (463, 8, 9), (2467, 6, 2), (3, 6, 2),
                                     (463, 9, 3), (2467, 10, 9), (700, 11, 2), (5000, 9, 2)])
def test_fm_multi_even_decimations(rr, monkeypatch, L, D, nch, kernel):
    """FmMulti with interp 1 and integer decimations — auto: decimations 2..8 on decimate-first tiles (k_fm_multi_poly), the
    others as half: even decimations with half-size inverse transforms on 2048-point tiles where the tile choice allows,
    else full-size inverses; odd and even filter lengths (the reference's nsamples, hence the parity of every call's
    start, alternates), 1..9 channels, small windows."""
    if kernel == "half":
        knob(rr, monkeypatch, fm_poly=-1)
    fs, n = 2.4e6, 200_000
    proto = (rnd_c(L, L + D) / max(1, L // 8)).astype(np.complex64)
    k = np.arange(L, dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * (c * 20e3) * k / fs)).astype(np.complex64) for c in range(nch)])
    x = fm_signal(n, fs, 0.0, L + D)
    blk = rr.FmMulti(taps, 1, D, 1.0)
    cap_in = 41_000
    outs = [[] for _ in range(nch)]
    pos, ring = 0, np.zeros(0, np.complex64)
    while True:
        take = min(cap_in - len(ring), len(x) - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, 30_000)
        ring = ring[c:]
        out = out.reshape(nch, -1)                    # (a one-channel block returns a flat window)
        for ch in range(nch):
            if p:
                outs[ch].append(out[ch])
        if take == 0 and c == 0 and p == 0:
            break
    for ch in range(nch):
        yg = np.concatenate(outs[ch]) if outs[ch] else np.zeros(0, np.float32)
        yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D), orc.QuadratureDemod(1.0)], x, stream_bytes=8 * cap_in)
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D)], x, stream_bytes=8 * cap_in)
        _demod_close(yg, yo, ro)


@pytest.mark.parametrize("D,odd", [(6, False), (6, True), (5, False), (2, True)])
def test_fm_multi_u8_shared_source(rr, D, odd):
    """rr.FmMultiU8 (RtlSdrDecode fused in front of the multi-channel kernel) == RtlSdrDecode -> FftFilter(taps_c) ->
    RationalResampler -> QuadratureDemod of the oracle per channel; byte-counted windows, odd-addressed windows."""
    fs, n, nch = 2.4e6, 200_000, 3
    proto = orc.low_pass_complex(fs, 100e3, 12.5e3)
    k = np.arange(len(proto), dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * ((c - 1) * 30e3) * k / fs)).astype(np.complex64)
                     for c in range(nch)])
    z = fm_signal(n, fs, 0.0, 31 + D)
    b = np.empty(2 * n, np.uint8)
    b[0::2] = np.clip(np.round(z.real / 0.008 * 0.5 + 127), 0, 255).astype(np.uint8)
    b[1::2] = np.clip(np.round(z.imag / 0.008 * 0.5 + 127), 0, 255).astype(np.uint8)
    blk = rr.FmMultiU8(taps, 1, D, 1.0)
    cap_in = 81_001 if odd else 4_096_000
    outs = [[] for _ in range(nch)]
    pos, ring = 0, np.zeros(0, np.uint8)
    pad = np.zeros(1, np.uint8)
    while True:
        take = min(cap_in - len(ring), len(b) - pos)
        ring = np.concatenate([ring, b[pos:pos + take]]); pos += take
        win = np.concatenate([pad, ring])[1:] if odd else ring          # (odd: an odd-addressed view of the same bytes)
        st, c, p, need, out = blk.work(win, 60_000)
        ring = ring[c:]
        out = out.reshape(nch, -1)
        for ch in range(nch):
            if p:
                outs[ch].append(out[ch])
        if take == 0 and c == 0 and p == 0:
            break
    for ch in range(nch):
        yg = np.concatenate(outs[ch])
        front = [orc.RtlSdrDecode(), orc.FftFilter(taps[ch]), orc.RationalResampler(1, D)]
        yo = run_chain(front + [orc.QuadratureDemod(1.0)], b)
        ro = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps[ch]), orc.RationalResampler(1, D)], b)
        _demod_close(yg, yo, ro)


@pytest.mark.parametrize("L,D", [(3599, 6), (4559, 6), (3039, 4), (7599, 10), (4607, 6), (4608, 6), (1519, 2),
                                 (1000, 9), (2467, 11), (1000, 12), (2467, 13), (3001, 14), (4000, 15), (3500, 16), (300, 13)])
def test_fm_chain_long_phases_on_decimate_first_tiles(rr, monkeypatch, L, D):
    """Round 4: the decimate-first tiles take up to 768 taps per phase (256 of a tile's 1024 positions are output), default
    for 1:4 … 1:10; both sides of the limit and the forced form for 1:2 against the oracle chain, small rings included."""
    if D == 2 or L == 300:
        knob(rr, monkeypatch, fm_poly=1)
    n = 250_000
    x = fm_signal(n, 2.4e6, 0.0, L)
    taps = (rnd_c(L, L) / max(1, L // 4)).astype(np.complex64)
    for stream_bytes in (4_096_000, 8 * (2 * 8192)):
        yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, D), orc.QuadratureDemod(1.0)], x, stream_bytes=4_096_000)
        ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, D)], x, stream_bytes=4_096_000)
        yg = run_chain([rr.FmChain(taps, 1, D, 1.0)], x, stream_bytes=stream_bytes)
        _demod_close(yg, yo, ro)


@pytest.mark.parametrize("L,D,nch", [(3119, 6, 3), (4500, 6, 2), (3039, 4, 2), (6143, 8, 2), (1500, 2, 3), (5500, 6, 2), (5568, 6, 2)])
def test_fm_multi_long_phases_on_decimate_first_tiles(rr, L, D, nch):
    """... and FmMulti with them beyond the 4096-point shared-forward kernels' 4094 taps (the bookkeeping object is
    tile-agnostic): every channel against its own oracle chain, through reference-sized windows."""
    n = 300_000
    x = fm_signal(n, 2.4e6, 0.0, L + 1)
    taps = np.stack([(rnd_c(L, L + c) / max(1, L // 4)).astype(np.complex64) for c in range(nch)])
    blk = rr.FmMulti(taps, 1, D, 1.0)
    assert "per channel" not in blk.name
    outs, ring, pos = [[] for _ in range(nch)], np.zeros(0, np.complex64), 0
    while True:
        take = min(512_000 - len(ring), n - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, 1_024_000)
        ring = ring[c:]
        for ch in range(nch):
            outs[ch].append(out[ch])
        if take == 0 and c == 0 and p == 0:
            break
    for ch in range(nch):
        yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D), orc.QuadratureDemod(1.0)], x)
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D)], x)
        _demod_close(np.concatenate(outs[ch]), yo, ro)


def test_fm_multi_long_filters(rr):
    """FmMulti with the rtl_fm-sized filter (2467 taps -> 4096-point tiles); beyond 4094 taps: test_no_constructor_cliffs"""
    fs, n = 1.024e6, 250_000
    proto = orc.low_pass_complex(fs, 100e3, 1e3)
    assert len(proto) == 2467
    k = np.arange(len(proto), dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * (c * 50e3) * k / fs)).astype(np.complex64) for c in range(3)])
    x = fm_signal(n, fs, 0.0, 9)
    st, c, p, need, out = rr.FmMulti(taps, 25, 128, 1.0).work(x, 200_000)
    for ch in range(3):
        yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(25, 128), orc.QuadratureDemod(1.0)], x)
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(25, 128)], x)       # fresh blocks: they carry state
        assert p == len(yo)
        _demod_close(out[ch], yo, ro)


# ---- device-resident streams (rr_dstream) --------------------------------------------
def run_chain_device(rr, blocks, x, stream_bytes=4_096_000, streams=None):
    """the same stream graph as harness.run_chain, but every ring lives in HBM and blocks run through
    rr_block_work_streams; only the source push and the sink pop touch the host.  streams = (source, per block ..., sink) raw
    HIP stream handles: every stage on its OWN stream (the rings order them: rr_dstream will_read / will_write)."""
    rings = [rr.DeviceStream(blocks[0].in_dtype, stream_bytes)] + [rr.DeviceStream(b.out_dtype, stream_bytes) for b in blocks]
    x = np.asarray(x, blocks[0].in_dtype)
    streams = streams or [0] * (len(blocks) + 2)
    pos, outs = 0, []
    for _ in range(1_000_000):
        moved = rings[0].push(x[pos:], streams[0])
        pos += moved
        for i, b in enumerate(blocks):
            while True:
                st, c, p, need = b.work_streams(rings[i], rings[i + 1], streams[1 + i])
                moved += c + p
                if st != AGAIN or (c == 0 and p == 0):
                    break
        y = rings[-1].pop(stream=streams[-1])
        moved += len(y)
        if len(y):
            outs.append(y)
        if moved == 0:
            break
    return np.concatenate(outs) if outs else np.zeros(0, blocks[-1].out_dtype)


@pytest.mark.parametrize("no_vmm", [False, True])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 3_001])
def test_device_streams_equal_host_windows(rr, monkeypatch, stream_bytes, no_vmm):
    if no_vmm:
        knob(rr, monkeypatch, dstream_no_vmm=1)
    """chains over HBM-resident rings (small rings force the wrap-around move) produce bit-identical output
    to the same blocks driven through host windows"""
    fs = 2.4e6
    x = fm_signal(200_000, fs, 0.0, 5)
    taps = orc.low_pass_complex(fs, 100e3, 12.5e3 * 4)
    mk = lambda: [rr.FftFilter(taps), rr.RationalResampler(3, 7, np.complex64), rr.QuadratureDemod(0.5)]
    yh = run_chain(mk(), x, stream_bytes=stream_bytes)
    yd = run_chain_device(rr, mk(), x, stream_bytes=stream_bytes)
    assert len(yh) == len(yd) > 0 and np.array_equal(yh, yd)
    xr = rnd_f(150_000, 8)
    mk2 = lambda: [rr.Hilbert(65), rr.FirFilter(orc.low_pass_complex(100e6, 5e6, 943e3 * 2), deci=8)]
    yh = run_chain(mk2(), xr, stream_bytes=max(stream_bytes, 8 * 3_001))
    yd = run_chain_device(rr, mk2(), xr, stream_bytes=max(stream_bytes, 8 * 3_001))
    assert len(yh) == len(yd) > 0 and np.array_equal(yh, yd)
    b = np.random.default_rng(1).integers(0, 256, 100_001, dtype=np.uint8)
    mk3 = lambda: [rr.RtlSdrDecode(), rr.RationalResampler(5, 3, np.complex64)]
    assert np.array_equal(run_chain(mk3(), b, stream_bytes=stream_bytes), run_chain_device(rr, mk3(), b, stream_bytes=stream_bytes))


@pytest.mark.parametrize("no_vmm", [False, True])
def test_device_rings_order_their_streams(rr, monkeypatch, no_vmm):
    """Round 3: every stage of a device-resident graph on its OWN HIP stream — the source's pushes, each block's kernels, the
    sink's downloads — with small rings (so that a push of window k + 1 lands in the slots the blocks of window k are still
    reading): the rings order the streams themselves and the output stays bit-identical to the one-stream run."""
    import torch
    if no_vmm:
        knob(rr, monkeypatch, dstream_no_vmm=1)
    fs = 2.4e6
    x = fm_signal(300_000, fs, 0.0, 15)
    taps = orc.low_pass_complex(fs, 100e3, 12.5e3 * 4)
    mk = lambda: [rr.FftFilter(taps), rr.RationalResampler(3, 7, np.complex64), rr.QuadratureDemod(0.5)]
    for stream_bytes in (8 * 9_000, 4_096_000):
        y1 = run_chain_device(rr, mk(), x, stream_bytes=stream_bytes)
        ss = [torch.cuda.Stream() for _ in range(5)]
        yn = run_chain_device(rr, mk(), x, stream_bytes=stream_bytes, streams=[s.cuda_stream for s in ss])
        torch.cuda.synchronize()
        assert len(y1) == len(yn) > 0 and np.array_equal(y1, yn)
    # ... and a ring-to-ring copy on a third stream between a writer and a reader on two others
    a, b = rr.DeviceStream(np.float32, 4 * 50_000), rr.DeviceStream(np.float32, 4 * 50_000)
    s0, s1, s2 = (torch.cuda.Stream() for _ in range(3))
    v = np.arange(200_000, dtype=np.float32)
    got, pos = [], 0
    while pos < len(v) or a.readable() or b.readable():
        pos += a.push(v[pos:], s0.cuda_stream)
        n = min(a.readable(), b.free(s1.cuda_stream))
        if n:
            assert rr.lib().rr_dstream_copy(b._h, 0, a._h, 0, n, rr.C.c_void_p(s1.cuda_stream)) == 0
            assert rr.lib().rr_dstream_produce(b._h, n) == 0 and rr.lib().rr_dstream_consume(a._h, n) == 0
        got.append(b.pop(stream=s2.cuda_stream))
    assert np.array_equal(np.concatenate(got), v)


@pytest.mark.parametrize("no_vmm", [False, True])
def test_device_stream_ring_contract(rr, monkeypatch, no_vmm):
    if no_vmm:
        knob(rr, monkeypatch, dstream_no_vmm=1)      # the linear fallback
    s = rr.DeviceStream(np.uint32, 4 * 10)
    assert s.double_mapped == (not no_vmm)
    assert s.capacity == 10 and s.readable() == 0 and s.free() == 10
    assert s.push(np.arange(25, dtype=np.uint32)) == 10 and s.free() == 0
    assert np.array_equal(s.pop(4), np.arange(4, dtype=np.uint32)) and s.readable() == 6 and s.free() == 4
    total, nxt, got = 10, 10, list(range(4))
    for k in range(40):                                    # many wrap-arounds of a 10-element ring
        n = s.push(np.arange(nxt, nxt + 7, dtype=np.uint32)); nxt += n
        got += list(s.pop(1 + k % 5))
    got += list(s.pop())
    assert got == list(range(len(got))) and len(got) == nxt
    with pytest.raises(Exception):
        rr.DeviceStream(np.uint32, 2)                      # smaller than one element


@pytest.mark.parametrize("aligned", [True, False])
def test_reregistered_addresses_keep_working(rr, aligned):
    """csrc/blocks.cpp "WHICH ranges run zero-copy": kernels work in place only on page-aligned whole-page ranges none of whose
    pages was registered before (aligned=True, first cycle); a range that was released and registered again, and any range
    that is not page-aligned (aligned=False: a heap array shares its first and last page with its neighbours), is page-locked
    but staged through DMA.  Whatever path a window takes, the samples are the same."""
    n = 200_000
    x = rnd_f(n, 8)
    if aligned:
        a, b = rr.host_ring(4 * (n + 32)).view(np.float32), rr.host_ring(4 * (n + 32)).view(np.float32)
        assert a.ctypes.data % 4096 == 0 and a.nbytes % 4096 == 0
    else:
        a, b = np.zeros(n + 32, np.float32), np.zeros(n + 32, np.float32)
    want = x * np.float32(0.25)
    blk = rr.MultiplyConst(0.25)
    for cycle in range(4):
        rr.host_register(a); rr.host_register(b)
        try:
            for k in range(3):
                a[7:7 + n] = x
                b[:] = -1.0
                st, c, p, need = blk.work_into(a[7:7 + n], b[3:], n)
                assert (c, p) == (n, n) and np.array_equal(b[3:3 + n], want), (cycle, k)
        finally:
            rr.host_unregister(a); rr.host_unregister(b)


@pytest.mark.parametrize("aligned", [False, True])
def test_push_from_a_page_locked_window_is_done_on_return(rr, aligned):
    """A source ring's window is consumed right after GpuUpload's copy_in and its writer overwrites it: out of a
    page-locked (rr_host_register'd) ring the copy is really asynchronous, so the call has to wait for it (round 4: the
    thread-per-block runner saw the overwritten samples).  aligned = a page-aligned ring (rr.host_ring, what the shims
    register): copy_in / copy_out run as copy KERNELS on the ring's device view (round 5); otherwise as DMA."""
    from harness import _arena
    n = (32 << 20) if aligned else (64 << 20)              # 128 / 256 MB: the copy takes ms, the overwrite microseconds
    s = rr.DeviceStream(np.uint32, 4 * n)
    # (aligned: the process-wide arena of tests/harness.py, registered once and never released — a fresh mapping registered
    #  here could land on addresses an earlier test retired, and would then be staged like any other range)
    x = _arena(rr, "out", 4 * n).view(np.uint32) if aligned else np.empty(n, np.uint32)
    x[:] = np.arange(n, dtype=np.uint32)
    if not aligned:
        rr.host_register(x)
    assert rr.host_window_in_place(x) == aligned
    try:
        for k in range(3):
            assert s.push(x) == n
            x[:] = 0xDEADBEEF                              # the upstream writer reuses the window
            got = s.pop()
            assert np.array_equal(got, np.arange(n, dtype=np.uint32)), k
            x[:] = np.arange(n, dtype=np.uint32)
        if aligned:                                        # copy_out INTO the registered ring, odd offsets and lengths (byte-wise path too)
            y8 = x.view(np.uint8)
            s8 = rr.DeviceStream(np.uint8, 1 << 20)
            src = np.random.default_rng(1).integers(0, 256, 300_001, dtype=np.uint8)
            assert s8.push(src) == len(src)
            h = lib_pop_into(rr, s8, y8[3:3 + len(src)])
            assert h == len(src) and np.array_equal(y8[3:3 + len(src)], src)
    finally:
        if not aligned:
            rr.host_unregister(x)


def lib_pop_into(rr, stream, dst):
    """rr_dstream_copy_out into a caller-owned host window + consume -> elements moved"""
    import ctypes as C
    n = min(stream.readable(), len(dst))
    L = rr.lib()
    assert L.rr_dstream_copy_out(stream._h, 0, dst.ctypes.data_as(C.c_void_p), n, C.c_void_p(0)) == 0
    assert L.rr_dstream_consume(stream._h, n) == 0
    return n


# ---- FirFilter -> FftFilter fused into one convolution (the north star's "127-tap FIR + 1024-pt FftFilter chain") ----
@pytest.mark.parametrize("L1,L2,cplx", [(127, 401, False), (1, 1, True), (5, 300, True), (64, 64, False), (200, 17, True),
                                        (127, 2467, False), (33, 1000, True), (2, 512, True)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 9_000])
def test_fir_fftfilter_fused_block(rr, L1, L2, cplx, stream_bytes):
    """rr.FirFftFilter (ONE convolution with the composite taps) == FirFilter(t1) -> FftFilter(t2) of the oracle over the
    whole stream for any chunking — including the first L2 - 1 outputs, where FftFilter starts from ZERO history while a
    single long filter would see the FIR's warm output (the head fix)."""
    if L1 == 127 and L2 == 401:
        t1, t2 = orc.low_pass_complex(10e6, 1e6, 190e3), orc.low_pass_complex(10e6, 1e6, 60e3)
    else:
        t1 = rnd_c(L1, L1 + 3) / max(1, L1 // 8)
        t2 = rnd_c(L2, L2 + 5) / max(1, L2 // 8)
        if not cplx:
            t1, t2 = t1.real.astype(np.complex64), t2.real.astype(np.complex64)
    x = rnd_c(120_000, L1 * 31 + L2)
    yo = run_chain([orc.FirFilter(t1), orc.FftFilter(t2)], x, stream_bytes=stream_bytes)
    yg = run_chain([rr.FirFftFilter(t1, t2)], x, stream_bytes=stream_bytes)
    assert len(yo) == len(yg) > 0
    assert max_norm_err(yg, yo) <= TOL
    # the head alone, against its own scale (the first outputs are small: the filter is still filling)
    h = min(len(yo), L2 + 8)
    assert max_norm_err(yg[:h], yo[:h], scale=max(float(np.max(np.abs(yo[:h]))), 1e-3 * float(np.max(np.abs(yo))))) <= 10 * TOL


def test_fir_fftfilter_fused_protocol(rr):
    t1, t2 = orc.low_pass_complex(10e6, 1e6, 190e3), orc.low_pass_complex(10e6, 1e6, 60e3)    # 127, 401 -> nsamples 623
    b = rr.FirFftFilter(t1, t2)
    x = rnd_c(5000, 1)
    assert b.work(x[:100], 600)[:4] == (WAIT_DST, 0, 0, 623)            # FftFilter stage: room for one block first
    assert b.work(x[:100], 1000)[:4] == (WAIT_SRC, 0, 0, 127)           # below the FIR's minimum
    st, c, p, need, out = b.work(x[:1000], 5000)                        # 874 FIR outputs: one block + 251 pending
    assert (st, c, p, need) == (WAIT_SRC, 874, 623, 623 - 251 + 126)
    st, c, p, need, out = b.work(x[874:], 700)                          # output-limited: one more block, pending drained
    assert (st, c, p, need) == (WAIT_DST, 623 - 251, 623, 623)
    with pytest.raises(ValueError):
        rr.FirFftFilter(np.zeros(0, np.complex64), t2)


@pytest.mark.parametrize("L1,L2,I,D", [(127, 401, 1, 4), (127, 401, 1, 6), (31, 463, 1, 6), (64, 200, 3, 7), (9, 127, 25, 128), (127, 401, 1, 1),
                                       (16, 300, 1, 250)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 20_000])
def test_fir_fm_chain_fused_block(rr, L1, L2, I, D, stream_bytes):
    """rr.FirFmChain (FirFilter -> FftFilter -> RationalResampler -> QuadratureDemod as ONE kernel: BASELINE.json's
    metric chain) == the four oracle blocks, whole stream, any chunking, incl. the demodulated samples that touch
    FftFilter's zero-history head."""
    fs = 10e6
    if (L1, L2) == (127, 401):
        t1, t2 = orc.low_pass_complex(fs, 1e6, 190e3), orc.low_pass_complex(fs, 1e6, 60e3)
    else:
        t1 = (rnd_c(L1, L1) / max(1, L1 // 4)).astype(np.complex64)
        t2 = (rnd_c(L2, L2 + 1) / max(1, L2 // 4)).astype(np.complex64)
    x = fm_signal(150_000, fs, 0.0, L1 + L2 + D)
    gain = 0.7
    front = [orc.FirFilter(t1), orc.FftFilter(t2), orc.RationalResampler(I, D)]
    yo = run_chain([orc.FirFilter(t1), orc.FftFilter(t2), orc.RationalResampler(I, D), orc.QuadratureDemod(gain)], x, stream_bytes=max(stream_bytes, 8 * 20_000))
    ro = run_chain(front, x, stream_bytes=max(stream_bytes, 8 * 20_000))
    yg = run_chain([rr.FirFmChain(t1, t2, I, D, gain)], x, stream_bytes=stream_bytes)
    _demod_close(yg / gain, yo / gain, ro)


# ---- fused audio stage: FftFilterFloat -> RationalResampler -> MultiplyConst as one kernel --------------
@pytest.mark.parametrize("L,I,D", [(963, 48000, 200000), (963, 44100, 200000), (65, 1, 1), (127, 1, 4), (401, 3, 2), (1, 5, 7), (3584, 1, 5),
                                   (500, 7, 3)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 4 * 30_000])
def test_audio_chain_fused_block(rr, L, I, D, stream_bytes):
    """rr.AudioChain (one real-valued kernel) == FftFilterFloat -> RationalResampler -> MultiplyConst of the oracle
    (examples/rtl_fm.rs:398-418: low_pass(200 kHz, 44.1 kHz, 500 Hz) = 963 taps, 200000 -> 48000, volume), whole stream,
    any chunking."""
    if L == 963:
        taps = orc.low_pass(200_000.0, 44_100.0, 500.0)
        assert len(taps) == 963
    else:
        taps = rnd_f(L, L + 2) / max(1, L // 8)
    x = rnd_f(250_000, L + I + D)
    vol = 0.35
    yo = run_chain([orc.FftFilterFloat(taps), orc.RationalResampler(I, D, np.float32), orc.MultiplyConst(vol)], x, stream_bytes=max(stream_bytes, 4 * 30_000))
    yg = run_chain([rr.AudioChain(taps, I, D, vol)], x, stream_bytes=stream_bytes)
    assert len(yo) == len(yg) > 0
    assert max_norm_err(yg, yo) <= TOL


def test_audio_chain_protocol(rr):
    taps = orc.low_pass(200_000.0, 44_100.0, 500.0)           # 963 taps -> nsamples 2048 - 963 = 1085
    b = rr.AudioChain(taps, 48000, 200000, 1.0)                 # 6 : 25
    x = rnd_f(5000, 2)
    # a window below one block's outputs (ceil(1085 * 6 / 25) = 261) does not stop the input side (test_fm_chain_fused_protocol)
    assert b.work(x[:100], 200)[:4] == (WAIT_SRC, 100, 0, 985)
    st, c, p, need, out = b.work(x[100:3000], 1000)             # 3000 = 2 blocks + 830
    assert (st, c, p, need) == (WAIT_SRC, 2900, 521, 1085 - 830)
    with pytest.raises(ValueError):
        rr.AudioChain(taps, 0, 5)                               # the reference's own argument error, fused or not
    with pytest.raises(ValueError):
        rr.AudioChain(np.zeros(0, np.float32), 1, 5)


# ---- no constructor cliffs: the fused constructors take every shape the separate blocks take ------------
@pytest.mark.parametrize("L", [5000, 20000])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 70_000])
def test_no_constructor_cliffs_fm_chain(rr, L, stream_bytes):
    """rr.FmChain / FmChainU8 at 5000 taps (fused: split tiles) and 20000 taps (beyond the 16384-point tiles: the SAME
    constructor returns the unfused composition FftFilter -> RationalResampler -> QuadratureDemod behind one handle) against
    the oracle's three blocks; the reference sizes its transform from any tap count (fft_filter.rs:36-42)."""
    fs, n = 2.4e6, 300_000
    x = fm_signal(n, fs, 0.0, 7 + L)
    taps = (rnd_c(L, L) / (L // 4)).astype(np.complex64)
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(0.7)], x, stream_bytes=stream_bytes)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 6)], x, stream_bytes=stream_bytes)
    yg = run_chain([rr.FmChain(taps, 1, 6, 0.7)], x, stream_bytes=stream_bytes)
    assert len(yo) > 0
    _demod_close(yg / 0.7, yo / 0.7, ro)
    if stream_bytes == 4_096_000:                               # ... and from the RTL-SDR byte stream
        b = np.empty(2 * n, np.uint8)
        b[0::2] = np.clip(np.round(x.real / 0.008 + 127), 0, 255)
        b[1::2] = np.clip(np.round(x.imag / 0.008 + 127), 0, 255)
        front = [orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(1, 6)]
        yo8 = run_chain(front + [orc.QuadratureDemod(0.7)], b)
        ro8 = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(1, 6)], b)
        yg8 = run_chain([rr.FmChainU8(taps, 1, 6, 0.7)], b)
        _demod_close(yg8 / 0.7, yo8 / 0.7, ro8)


def test_no_constructor_cliffs_fm_chain_beyond_the_tile(rr):
    """a decimation larger than any tile (1:20000 on a 65-tap filter) is a shape the three blocks take: so does rr.FmChain"""
    x = fm_signal(400_000, 2.4e6, 0.0, 3)
    taps = orc.low_pass_complex(2.4e6, 100e3, 80e3)
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 20000), orc.QuadratureDemod(1.0)], x)
    yg = run_chain([rr.FmChain(taps, 1, 20000, 1.0)], x)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 20000)], x)
    assert len(yo) == len(yg) > 5
    _demod_close(yg, yo, ro)


@pytest.mark.parametrize("L", [5000, 20000])
def test_no_constructor_cliffs_fm_multi(rr, L):
    """rr.FmMulti beyond 4094 taps: one chain per channel on the shared window behind the same handle (each still fused
    up to 16383 taps), every channel against its own oracle chain"""
    fs, n = 2.4e6, 260_000
    x = fm_signal(n, fs, 0.0, 11 + L)
    proto = (rnd_c(L, L + 1) / (L // 4)).astype(np.complex64)
    k = np.arange(L, dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * c * 30e3 * k / fs)).astype(np.complex64) for c in range(3)])
    blk = rr.FmMulti(taps, 1, 6, 1.0)
    outs = [[] for _ in range(3)]
    ring, pos = np.zeros(0, np.complex64), 0
    while True:
        take = min(512_000 - len(ring), n - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, 100_000)
        ring = ring[c:]
        for ch in range(3):
            outs[ch].append(out.reshape(3, -1)[ch])
        if take == 0 and c == 0 and p == 0:
            break
    for ch in range(3):
        yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)], x)
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, 6)], x)
        _demod_close(np.concatenate(outs[ch]), yo, ro)


@pytest.mark.parametrize("L", [3585, 5000, 20000])
def test_no_constructor_cliffs_audio_chain(rr, L):
    """rr.AudioChain beyond 3584 taps: FftFilterFloat -> RationalResampler -> MultiplyConst unfused behind the same handle"""
    taps = rnd_f(L, L) / (L // 8)
    x = rnd_f(300_000, L + 1)
    yo = run_chain([orc.FftFilterFloat(taps), orc.RationalResampler(6, 25, np.float32), orc.MultiplyConst(0.35)], x)
    yg = run_chain([rr.AudioChain(taps, 6, 25, 0.35)], x)
    assert len(yo) == len(yg) > 0
    assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("L,I,D", [(463, 1, 6), (127, 25, 128), (2467, 1, 5)])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 20_000])
def test_fm_chain_with_fastfm_demodulator(rr, L, I, D, stream_bytes):
    """SURVEY §8 f3 / VERDICT r2 #5: FastFM (quadrature_demod.rs:144-165) as the chain's demodulator through the fused
    constructors (atan2_mode = RR_DEMOD_FASTFM): FftFilter -> RationalResampler -> FastFM.  FastFM multiplies differences of
    samples, so the bound is the filter stage's 1e-5 propagated through it: |d out| <= 8 eps max|r|, eps = 1e-5 max|r| (two
    products of a difference of two samples, within 2 eps and up to 2 max|r|, with a sample, within eps and up to max|r|)."""
    fs = 2.4e6
    x = fm_signal(300_000, fs, 0.0, 31 + L)
    taps = orc.low_pass_complex(fs, 100e3, 12.5e3) if L == 463 else (rnd_c(L, L) / max(1, L // 4)).astype(np.complex64)
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.FastFM()], x, stream_bytes=stream_bytes)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], x, stream_bytes=stream_bytes)
    yg = run_chain([rr.FmChain(taps, I, D, 1.0, rr.DEMOD_FASTFM)], x, stream_bytes=stream_bytes)
    assert len(yo) == len(yg) == len(ro) > 0
    rmax = float(np.max(np.abs(ro)))
    assert np.max(np.abs(yg.astype(np.float64) - yo.astype(np.float64))) <= 8 * TOL * rmax * rmax
    # ... and for N channels on one window
    taps2 = np.stack([taps, np.conj(taps)])
    st, c, p, need, out = rr.FmMulti(taps2, I, D, 1.0, rr.DEMOD_FASTFM).work(x, 4_096_000 // 4)
    yo2 = run_chain([orc.FftFilter(taps2[1]), orc.RationalResampler(I, D), orc.FastFM()], x)
    assert p == len(yo2) and np.max(np.abs(out.reshape(2, -1)[1][:p].astype(np.float64) - yo2)) <= 8 * TOL * rmax * rmax


# ---- .translate(): the exact rotator replay on the device, and the model's drift against it --------------------------
def test_translate_replay_on_device_matches_the_oracle_bit_for_bit_rotator(rr):
    """RR_ROT_REPLAY walks the reference's f32 recurrence (fir.rs:464-473) on the device from the phase carried between
    calls: the rotator itself is then bit-identical to the oracle's for ANY stream length, so a long translated stream
    (600,000 outputs through many windows) stays inside the FIR's own 1e-5, where the closed-form model has drifted."""
    fs, f, d = 2.4e6, 312_345.0, 2
    taps = orc.low_pass_complex(fs, 100e3, 25e3)
    x = rnd_c(1_200_400, 99)
    mk = lambda m, **kw: [m.FirFilter(taps, deci=d, translate=(fs, f), **kw)]
    yo = run_chain(mk(orc), x, stream_bytes=8 * 40_000)
    yg = run_chain(mk(rr, rotator=rr.ROT_REPLAY), x, stream_bytes=8 * 40_000)
    assert len(yo) == len(yg) >= 600_000
    assert max_norm_err(yg, yo) <= TOL
    assert max_norm_err(yg[-50_000:], yo[-50_000:]) <= TOL      # no growth towards the end of the stream


def test_zero_copy_host_windows_equal_staged_windows(rr):
    """Round 4: page-locked host windows are read and written by the kernels in place (Block::work_host: zero copy, the
    upload and the download overlap on the full-duplex link); pageable ones are staged through device memory.  Same kernels,
    same protocol, the same bits — for every kind of block, multi-window ones, blocks that copy from / to the caller's
    windows themselves (FftFilterFloat's inner streams, the resampler's pending sample, the fused chains' output tail)."""
    rng = np.random.default_rng(5)
    xc = rnd_c(300_000, 1)
    xf = rnd_f(300_000, 2)
    xb = rng.integers(0, 256, 600_001, dtype=np.uint8)
    tc = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
    tf = orc.low_pass(200_000.0, 44_100.0, 500.0)
    t3 = np.stack([tc, np.conj(tc), tc[::-1].copy()])
    long_taps = (rnd_c(20_000, 3) / 5000).astype(np.complex64)
    cases = [
        (lambda: rr.FirFilter(tc, deci=6), xc, 100_000, 50_000), (lambda: rr.FirFilter(tf, deci=4), xf, 120_000, 50_000),
        (lambda: rr.FirFilter(tc, translate=(2.4e6, 1e5)), xc, 100_000, 100_000),
        (lambda: rr.FftFilter(tc), xc, 80_000, 80_000), (lambda: rr.FftFilterFloat(tf), xf, 90_000, 90_000),
        (lambda: rr.RationalResampler(5, 3, np.complex64), xc, 50_000, 1_000), (lambda: rr.QuadratureDemod(0.7), xc, 70_000, 70_000),
        (lambda: rr.Hilbert(65), xf, 60_000, 60_000), (lambda: rr.Hilbert(1001), xf, 120_000, 120_000),
        (lambda: rr.HilbertFir(65, tc, 8), xf, 200_000, 50_000), (lambda: rr.HilbertFir(65, tc, 3), xf, 200_000, 100_000),
        (lambda: rr.FmChain(tc, 1, 6, 1.0), xc, 100_000, 50_000), (lambda: rr.FmChain(tc, 1, 6, 1.0), xc, 100_000, 7),
        (lambda: rr.FmChainU8(tc, 1, 6, 1.0), xb, 200_001, 50_000), (lambda: rr.AudioChain(tf, 6, 25, 0.5), xf, 100_000, 33),
        (lambda: rr.FmMulti(t3, 1, 6, 1.0), xc, 100_000, 20_000), (lambda: rr.FmMulti(t3, 2, 5, 1.0), xc, 100_000, 11),
        (lambda: rr.FmChain(long_taps, 1, 6, 1.0), xc, 200_000, 100_000),
        (lambda: rr.RtlSdrDecode(), xb, 100_001, 40_000), (lambda: rr.FftStream(1024), xc, 50_000, 50_000),
        (lambda: rr.MultiplyConst(0.5), xf, 50_000, 50_000),
    ]
    for k, (mk, x, in_cap, out_cap) in enumerate(cases):
        ya, la = drive_pageable(mk(), x, in_cap, out_cap)
        yb, lb = drive_registered(rr, mk(), x, in_cap, out_cap)
        assert la == lb, mk().name
        assert ya.shape == yb.shape and ya.shape[1] > 0, mk().name
        assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8)), mk().name
        # rr_build_opts.host_in_staged: the registered INPUT window copied down by a kernel first (+1; what the N-channel blocks do
        # by default) or read in place (-1): the same kernels behind it, the same bits
        with rr.build_options(host_in_staged=1 if k % 2 else -1):
            blk = mk()
        yc, lc = drive_registered(rr, blk, x, in_cap, out_cap)
        assert lc == la and np.array_equal(yc.view(np.uint8), ya.view(np.uint8)), (blk.name, k % 2)


def test_tag_rule_of_every_constructor(rr):
    """rr_block_tag_rule: what the reference block(s) behind a handle do with tags (include/rustradio_amd.h) — FirFilter
    `pos < n -> pos / deci` (fir.rs:536-545), FftFilter / FftFilterFloat / Hilbert / the sync blocks with their sample
    (fft_filter.rs:307-313,441-472; hilbert.rs:119-123), FftStream frame tags (fft_stream.rs:98-111), everything that holds a
    RationalResampler / QuadratureDemod / RtlSdrDecode drops them (rational_resampler.rs:156)."""
    import ctypes as C
    DROP, FORWARD, FRAMES = 0, 1, 2
    tc = orc.low_pass_complex(2.4e6, 100e3, 50e3)
    tf = tc.real.astype(np.float32)
    cases = [
        (rr.FirFilter(tc, deci=7), FORWARD, 7), (rr.FirFilter(tf, deci=3), FORWARD, 3), (rr.FirFilter(tc), FORWARD, 1),
        (rr.FftFilter(tc), FORWARD, 1), (rr.FftFilterFloat(tf), FORWARD, 1), (rr.Hilbert(65), FORWARD, 1),
        (rr.MultiplyConst(0.5), FORWARD, 1), (rr.FastFM(), FORWARD, 1), (rr.FirFftFilter(tc[:31], tc), FORWARD, 1),
        (rr.HilbertFir(65, tc, 8), FORWARD, 8), (rr.FftStream(256), FRAMES, 256),
        (rr.RationalResampler(2, 3, np.complex64), DROP, None), (rr.QuadratureDemod(1.0), DROP, None), (rr.RtlSdrDecode(), DROP, None),
        (rr.FmChain(tc, 1, 6, 1.0), DROP, None), (rr.FmChainU8(tc, 1, 6, 1.0), DROP, None), (rr.AudioChain(tf, 6, 25, 0.5), DROP, None),
        (rr.FirFmChain(tc[:31], tc, 1, 4, 1.0), DROP, None), (rr.FmMulti(np.stack([tc, np.conj(tc)]), 1, 6, 1.0), DROP, None),
    ]
    for blk, rule, param in cases:
        p = C.c_size_t(0)
        assert rr.lib().rr_block_tag_rule(blk._h, C.byref(p)) == rule, blk.name
        if param is not None:
            assert p.value == param, (blk.name, p.value)
    assert rr.lib().rr_block_tag_rule(None, None) == -1


@pytest.mark.parametrize("nbytes", [1, 4095, 2 << 20, (2 << 20) + 1, (6 << 20) + 12345, 21_000_003])
def test_pageable_copies_go_through_the_librarys_pinned_chunks(rr, nbytes):
    """Round 6 (csrc/stage.*): rr_dstream_copy_in / _out and rr_block_work on memory the caller did not register never hand the
    pointer to a GPU engine — the CPU copies it through two pinned 2 MB chunks.  Any size, chunk boundaries included, byte for
    byte; and a block on the same kind of window."""
    rng = np.random.default_rng(nbytes)
    x = rng.integers(0, 256, nbytes, dtype=np.uint8)
    assert not rr.host_window_in_place(x)
    s = rr.DeviceStream(np.uint8, max(nbytes, 4096))
    assert s.push(x) == nbytes
    x_copy = x.copy()
    x[:] = 0                                                   # (the source is the caller's again as soon as push returns)
    assert np.array_equal(s.pop(), x_copy)
    n = max(1, nbytes // 8)
    z = (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)
    st, c, p, need, y = rr.MultiplyConst(0.5 - 2j, np.complex64).work(z, n)
    st2, c2, p2, need2, yo = orc.MultiplyConst(0.5 - 2j, np.complex64).work(z, n)
    assert (st, c, p, need) == (st2, c2, p2, need2) and (c, p) == (n, n) and np.array_equal(y, yo)


def test_zz_fm_multi_per_channel_create_work_destroy_soak(rr):
    """500 create -> work -> destroy cycles of FmMulti beyond 4094 taps (the per-channel `Parallel`
    composition: N chains on a pool of forked streams joined by events, csrc/compose.cpp) on PAGEABLE windows, at the end of this
    file's run in the same process — the shape the abort() of rounds 4-5 was last seen in.  Every cycle's output must equal the
    first's (same input, fresh handles): a stale copy or a missed join shows as a different sample."""
    fs, n, L = 2.4e6, 40_000, 5000
    x = fm_signal(n, fs, 0.0, 77)
    proto = (rnd_c(L, L + 1) / (L // 4)).astype(np.complex64)
    k = np.arange(L, dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * c * 30e3 * k / fs)).astype(np.complex64) for c in range(3)])
    first = None
    for cycle in range(500):
        blk = rr.FmMulti(taps, 1, 6, 1.0)
        xin = x.copy()                                          # a fresh pageable window every cycle (recycled addresses)
        st, c, p, need, out = blk.work(xin, 100_000)
        assert p > 1000
        if first is None:
            first = out.copy()
            yo = run_chain([orc.FftFilter(taps[1]), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)], x)
            ro = run_chain([orc.FftFilter(taps[1]), orc.RationalResampler(1, 6)], x)
            _demod_close(out.reshape(3, -1)[1][:p], yo[:p], ro[:p + 1])
        else:
            assert np.array_equal(out, first), cycle
        del blk, xin, out
