"""CPU-only: pin the oracle (oracle/rr_oracle.c) against every known-answer test the
reference holds for the hot path, then against an independent numpy-f64
restatement of the whole-stream formulas."""
import numpy as np
import pytest

import known_answers as KA
from harness import AGAIN, WAIT_SRC, WAIT_DST, max_norm_err, run_chain
from oracle import pyoracle as orc


@pytest.mark.parametrize("check", KA.ALL_CHECKS, ids=lambda f: f.__name__)
def test_reference_known_answers(check):
    check(orc)


def test_quad_known_fast_mode():
    # fast-math flavour is only pinned at the reference's 1e-3 (quadrature_demod.rs:222-264)
    KA.check_quad_known(orc, orc.ATAN2_FAST)


def test_fast_atan2_error_bound():
    # documented max abs error of the approximation: < 0.0038 rad
    rng = np.random.default_rng(1)
    v = rng.standard_normal((20000, 2)).astype(np.float32)
    got = np.array([orc.fast_atan2(float(y), float(x)) for y, x in v])
    assert np.max(np.abs(got - np.arctan2(v[:, 0].astype(np.float64), v[:, 1].astype(np.float64)))) < 0.0039


# ---- f64 truths ---------------------------------------------------------------------
def rnd_c(n, seed):
    r = np.random.default_rng(seed)
    return (r.uniform(-1, 1, n) + 1j * r.uniform(-1, 1, n)).astype(np.complex64)


def fir_truth(taps, x, deci):
    """A.1: y[m] = sum_k t[k] x[m d + L-1-k], M = floor((N-L+1)/d), N >= L+d-1."""
    L = len(taps); N = len(x)
    if N < L + deci - 1:
        return np.zeros(0, np.complex128)
    full = np.convolve(x.astype(np.complex128), taps.astype(np.complex128))[L - 1:N]
    M = (N - L + 1) // deci
    return full[::deci][:M]


@pytest.mark.parametrize("L,deci", [(1, 1), (3, 2), (127, 1), (255, 8), (64, 5)])
def test_fir_matches_f64(L, deci):
    x = rnd_c(5000, L)
    taps = rnd_c(L, 100 + L)
    y = run_chain([orc.FirFilter(taps, deci=deci)], x)
    ref = fir_truth(taps, x, deci)
    assert len(y) == len(ref)
    assert max_norm_err(y, ref) < 2e-6


def test_fir_chunking_invariant():
    x = rnd_c(30000, 5); taps = rnd_c(33, 6)
    a = run_chain([orc.FirFilter(taps, deci=3)], x)
    b = run_chain([orc.FirFilter(taps, deci=3)], x, stream_bytes=8 * 1000)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("L", [1, 2, 5, 193, 256, 401, 463])
def test_fftfilter_matches_f64(L):
    x = rnd_c(20000, L)
    taps = rnd_c(L, 200 + L) / L
    blk = orc.FftFilter(taps)
    F, S = orc.fftfilter_dims(blk)
    n = 1
    while n < L:
        n <<= 1
    assert F == 2 * n and S == F - L
    y = run_chain([blk], x)
    assert len(y) == (len(x) // S) * S                   # A.4: remainder never flushed
    ref = np.convolve(x.astype(np.complex128), taps.astype(np.complex128))[:len(y)]
    assert max_norm_err(y, ref) < 3e-6


def test_fft_matches_numpy():
    for n in (2, 4, 8, 64, 512, 1024, 2048):
        x = rnd_c(n, n)
        assert max_norm_err(orc.fft(x), np.fft.fft(x.astype(np.complex128))) < 1e-6
        assert max_norm_err(orc.fft(x, inverse=True), np.fft.ifft(x.astype(np.complex128)) * n) < 1e-6


def test_fftfilter_equals_fir_shifted():
    # A.4: y_FIR[m] = y_FFT[m + L - 1]
    L = 127
    x = rnd_c(8000, 9); taps = orc.low_pass_complex(10e6, 1e6, 190e3)
    assert len(taps) == L
    yf = run_chain([orc.FirFilter(taps)], x)
    yo = run_chain([orc.FftFilter(taps)], x)
    n = min(len(yf), len(yo) - (L - 1))
    assert max_norm_err(yo[L - 1:L - 1 + n], yf[:n]) < 3e-6


def test_fftfilter_float():
    r = np.random.default_rng(3)
    x = r.uniform(-1, 1, 30000).astype(np.float32)
    taps = orc.low_pass(200e3, 44.1e3, 5000.0)
    y = run_chain([orc.FftFilterFloat(taps)], x)
    ref = np.convolve(x.astype(np.float64), taps.astype(np.float64))[:len(y)]
    assert len(y) > 0 and max_norm_err(y, ref) < 3e-6


def hilbert_truth(x, h):
    """A.8: xp = 0^L ++ x ; y[k] = (xp[k + L//2], sum_j h[L-1-j] xp[k+j])."""
    L = len(h)
    xp = np.concatenate([np.zeros(L), x.astype(np.float64)])
    im = np.convolve(xp, h.astype(np.float64))[L - 1:L - 1 + len(x)]
    return xp[L // 2:L // 2 + len(x)] + 1j * im


def test_hilbert_matches_f64_and_delay():
    r = np.random.default_rng(4)
    x = r.uniform(-1, 1, 10000).astype(np.float32)
    L = 65
    h = orc.hilbert_taps(orc.make_window(orc.WIN_HAMMING, L))
    assert abs(h[L // 2 + 1] - 0.6363) < 1e-3 and h[L // 2] == 0
    assert np.allclose(h, -h[::-1], atol=1e-7)
    y = run_chain([orc.Hilbert(L)], x)
    assert len(y) == len(x)
    assert np.array_equal(y.real[L - L // 2:], x[:len(x) - (L - L // 2)])   # F7: delay 33
    assert max_norm_err(y, hilbert_truth(x, h)) < 2e-6
    y2 = run_chain([orc.Hilbert(L)], x, stream_bytes=4 * 777)
    assert np.array_equal(y, y2)


def test_hilbert_rejects_even():
    for n in (0, 1, 2, 64):
        with pytest.raises(ValueError):
            orc.Hilbert(n)


def test_resampler_closed_form():
    # F3/A.6: y[m] = x[floor(m D / I)], count ceil(N I / D)
    x = np.arange(10007, dtype=np.uint32)
    for I, D in [(1, 6), (25, 128), (3, 2), (7, 7), (200000, 1024000), (48, 200)]:
        y = run_chain([orc.RationalResampler(I, D, np.uint32)], x, stream_bytes=4 * 5000)
        g = np.gcd(I, D); i, d = I // g, D // g
        cnt = -(-len(x) * i // d)
        assert len(y) == cnt
        assert np.array_equal(y, x[(np.arange(cnt, dtype=np.int64) * d) // i])


def test_quad_matches_f64():
    x = rnd_c(5000, 11)
    y = run_chain([orc.QuadratureDemod(0.7)], x)
    z = np.conj(x[:-1].astype(np.complex128)) * x[1:].astype(np.complex128)
    assert len(y) == len(x) - 1
    assert max_norm_err(y, 0.7 * np.angle(z), scale=np.pi * 0.7) < 1e-6


def test_fm_chain_counts():
    # cfg3 chain: FftFilter(463 taps) -> RationalResampler(1:6) -> QuadratureDemod
    taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
    assert len(taps) == 463
    x = rnd_c(100000, 12)
    y = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)], x)
    n1 = (len(x) // 561) * 561
    n2 = -(-n1 // 6)
    assert len(y) == n2 - 1


def test_tap_counts_for_baseline_configs():
    # SURVEY §8d pins
    assert orc.compute_ntaps(10e6, 190e3) == 127
    assert orc.compute_ntaps(10e6, 60e3) == 401
    assert orc.compute_ntaps(2.4e6, 12.5e3) == 463
    assert orc.compute_ntaps(100e6, 943e3) == 255
    assert orc.compute_ntaps(1.024e6, 1e3) == 2467


def test_sync_blocks_by_definition():
    """MultiplyConst / FastFM have no unit tests in the reference: pinned to their one-line definitions
    (multiply_const.rs:20-22, quadrature_demod.rs:158-164) evaluated in numpy f32 and to the sync
    work() protocol (rustradio_macros_code/src/lib.rs:458-515)."""
    import numpy as np
    from oracle import pyoracle as orc
    x = (np.arange(1, 8, dtype=np.float32) / 3).astype(np.float32)
    st, c, p, need, out = orc.MultiplyConst(1.7).work(x, 5)
    assert (st, c, p, need) == (2, 5, 5, 1) and np.array_equal(out, x[:5] * np.float32(1.7))
    st, c, p, need, out = orc.MultiplyConst(1.7).work(x, 50)
    assert (st, c, p, need) == (1, 7, 7, 1)
    assert orc.MultiplyConst(2.0).work(x[:0], 5)[:4] == (1, 0, 0, 1) and orc.MultiplyConst(2.0).work(x, 0)[:4] == (2, 0, 0, 1)
    z = (np.arange(6) * (0.5 - 0.25j) + 1j).astype(np.complex64)
    v = np.complex64(0.3 + 2j)
    want = (z.real * v.real - z.imag * v.imag) + 1j * (z.real * v.imag + z.imag * v.real)      # num-complex Mul
    assert np.array_equal(orc.MultiplyConst(v, np.complex64).work(z, 10)[4], want.astype(np.complex64))
    f = orc.FastFM()
    got = np.concatenate([f.work(z[:4], 10)[4], f.work(z[4:], 10)[4]])                            # state carries over
    s = np.concatenate([np.zeros(2, np.complex64), z])
    exp = [(np.float32(s[i + 2].imag - s[i].imag) * s[i + 1].real) - (np.float32(s[i + 2].real - s[i].real) * s[i + 1].imag)
           for i in range(len(z))]
    assert np.array_equal(got, np.asarray(exp, np.float32))


def test_fftstream_restatement():
    """FftStream (fft_stream.rs:71-117): frames against numpy's f64 FFT (rustfft's conventions: forward
    e^{-2 pi i kn/N}, unnormalised — SURVEY §8c), whole frames only, zero in -> zero out (fft.rs:65-85)."""
    import numpy as np
    from oracle import pyoracle as orc
    for size in (2, 4, 16, 128, 1024, 8192, 3, 12, 100, 1000, 1500):     # (not a power of two: the defining sum in f64)
        rng = np.random.default_rng(size)
        x = (rng.standard_normal(3 * size + 1) + 1j * rng.standard_normal(3 * size + 1)).astype(np.complex64)
        st, c, p, need, out = orc.FftStream(size).work(x, 10 * size)
        assert (st, c, p) == (0, 3 * size, 3 * size)
        ref = np.fft.fft(x[:3 * size].astype(np.complex128).reshape(3, size), axis=1).reshape(-1)
        assert np.max(np.abs(out - ref)) / np.max(np.abs(ref)) < 1e-6
    b = orc.FftStream(4)
    assert b.work(np.zeros(3, np.complex64), 100)[:4] == (1, 0, 0, 4) and b.work(np.zeros(8, np.complex64), 3)[:4] == (2, 0, 0, 4)
    st, c, p, need, out = b.work(np.zeros(8, np.complex64), 100)          # adds_frame_tags' data path (:130-150)
    assert (st, c, p) == (0, 8, 8) and np.array_equal(out, np.zeros(8, np.complex64))


def test_multiband_restatement():
    """fir::multiband (fir.rs:552-590, untested in the reference): the restatement against a numpy transcription
    (ifft of the mirrored brick response, rotate, window, 1/sqrt(N)) and its None cases."""
    import numpy as np
    from oracle import pyoracle as orc
    for ntaps, bands in ((101, [(0.1, 0.3)]), (64, [(0.0, 0.2), (0.5, 0.75)]), (7, [(0.0, 1.0)])):
        w = np.hamming(ntaps).astype(np.float32)
        t = orc.multiband(bands, w)
        ideal = np.zeros(ntaps, np.complex128)
        for lo, hi in bands:
            a, b = int(np.floor(np.float32(lo) * np.float32(ntaps / 2))), int(np.ceil(np.float32(hi) * np.float32(ntaps / 2)))
            for n in range(a, b):
                ideal[n] = 1.0
                ideal[ntaps - n - 1] = 1.0
        ref = np.roll(np.fft.ifft(ideal) * ntaps, ntaps // 2) * w / np.sqrt(np.float32(ntaps))
        assert t is not None and np.max(np.abs(t - ref)) <= 1e-6 * max(1.0, np.max(np.abs(ref)))
    assert orc.multiband([(0.5, 0.2)], np.ones(16, np.float32)) is None          # a > b
    assert orc.multiband([(0.0, 2.5)], np.ones(16, np.float32)) is None          # b > taps (:567)
    assert orc.multiband([(0.1, 0.2)], np.ones(0, np.float32)) is None           # taps == 0
