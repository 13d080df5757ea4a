"""GPU: seeded random sweep over block parameters, input lengths and ring sizes — every trial drives the oracle
and the HIP block through identical window sequences (harness.run_chain) and compares the protocol log and
the samples, like tests/test_gpu_parity.py::both().  Catches boundary-tile / carry-state / tile-shape corner
cases that fixed parameter lists miss."""
import numpy as np
import pytest

from harness import drive_pageable, drive_registered, knob, max_norm_err, run_chain
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def rr():
    import rustradio_amd
    return rustradio_amd


def _c(rng, n):
    return (rng.uniform(-1, 1, n) + 1j * rng.uniform(-1, 1, n)).astype(np.complex64)


def _both(rr, make, x, stream_bytes, exact=False, scale=None):
    lo, lg = [], []
    yo = run_chain(make(orc), x, stream_bytes=stream_bytes, log=lo)
    yg = run_chain(make(rr), x, stream_bytes=stream_bytes, log=lg)
    assert lo == lg, "work() protocol differs"
    assert len(yo) == len(yg)
    if exact:
        assert np.array_equal(yo, yg)
    elif len(yo):
        assert max_norm_err(yg, yo, scale) <= TOL


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_fir(rr, seed):
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.choice([1, 2, 3, 7, 8, 9, 31, 64, 65, 127, 200, 255, 256, 257, 401, 511, 1000, 2467]))
    d = int(rng.choice([1, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 16, 25, 64]))
    n = int(rng.integers(L + d, 120_000))
    cplx_taps = bool(rng.integers(0, 2))
    real_in = bool(rng.integers(0, 3) == 0)
    es = 4 if real_in else 8
    ring = int(rng.choice([4_096_000, es * (L + d + int(rng.integers(8, 5000)))]))
    if real_in:
        x = rng.uniform(-1, 1, n).astype(np.float32)
        taps = (rng.uniform(-1, 1, L) / max(1, L // 8)).astype(np.float32)
    else:
        x = _c(rng, n)
        taps = _c(rng, L) / max(1, L // 8)
        if not cplx_taps:
            taps = taps.real.astype(np.complex64)
    _both(rr, lambda m: [m.FirFilter(taps, deci=d)], x, ring)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fftfilter_and_chain(rr, seed):
    rng = np.random.default_rng(2000 + seed)
    L = int(rng.choice([1, 2, 5, 33, 100, 193, 255, 256, 257, 401, 463, 511, 512, 513, 1024, 1500, 2467]))
    n = int(rng.integers(10, 150_000))
    x = _c(rng, n)
    taps = _c(rng, L) / max(1, L // 4)
    nsamp = 2 * (1 << int(np.ceil(np.log2(L)))) - L if L > 1 else 1
    ring = int(rng.choice([4_096_000, 8 * (nsamp + int(rng.integers(1, 4000)))]))
    _both(rr, lambda m: [m.FftFilter(taps)], x, ring)
    I, D = int(rng.integers(1, 9)), int(rng.integers(1, 40))
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x, stream_bytes=ring)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], x, stream_bytes=ring)
    try:
        blk = rr.FmChain(taps, I, D, 1.0)
    except Exception:
        return                                          # decimation too large for the tile: refused, not wrong
    if ring // 4 < -(-nsamp * I // D) + 1:
        return                                          # the fused block emits whole filter blocks: needs room for one (header)
    yg = run_chain([blk], x, stream_bytes=ring)
    assert len(yg) == len(yo)
    if len(yo):
        eps = TOL * float(np.max(np.abs(ro)))
        mag = np.abs(ro.astype(np.complex128))
        bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
        dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
        dd = np.minimum(dd, 2 * np.pi - dd)
        assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_streaming_blocks(rr, seed):
    rng = np.random.default_rng(3000 + seed)
    n = int(rng.integers(1, 200_000))
    ring = int(rng.choice([4_096_000, 8 * int(rng.integers(70, 6000))]))
    x = _c(rng, n)
    I, D = int(rng.integers(1, 50)), int(rng.integers(1, 50))
    _both(rr, lambda m: [m.RationalResampler(I, D, np.complex64)], x, ring, exact=True)
    gain = float(rng.uniform(0.1, 3))
    _both(rr, lambda m: [m.QuadratureDemod(gain)], x, ring, scale=np.pi * gain)
    _both(rr, lambda m: [m.FastFM()], x, ring, exact=True)
    hn = int(rng.choice([3, 5, 31, 33, 63, 65, 127, 129]))
    xr = rng.uniform(-1, 1, n).astype(np.float32)
    _both(rr, lambda m: [m.Hilbert(hn)], xr, max(ring, 4 * 300))
    b = rng.integers(0, 256, n, dtype=np.uint8)
    _both(rr, lambda m: [m.RtlSdrDecode()], b, ring, exact=True)
    size = int(rng.choice([2, 8, 64, 512, 1024, 2048, 4096]))
    _both(rr, lambda m: [m.FftStream(size)], x, max(ring, 8 * (size + 3)))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_hilbert_fir(rr, seed):
    rng = np.random.default_rng(4000 + seed)
    hn = int(rng.choice([3, 31, 65, 129]))
    L = int(rng.choice([1, 8, 33, 100, 255, 400]))
    d = int(rng.choice([1, 2, 3, 5, 8, 12, 16]))
    n = int(rng.integers(L + d + hn, 150_000))
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps = _c(rng, L) / max(1, L // 8)
    if rng.integers(0, 2):
        taps = taps.real.astype(np.complex64)
    ring = int(rng.choice([4_096_000, 4 * (L + d + int(rng.integers(8, 5000)))]))
    yo = run_chain([orc.Hilbert(hn), orc.FirFilter(taps, deci=d)], x)
    yg = run_chain([rr.HilbertFir(hn, taps, d)], x, stream_bytes=ring)
    assert len(yg) == len(yo)
    if len(yo):
        assert max_norm_err(yg, yo) <= TOL


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_translate_and_u8_chain(rr, seed):
    rng = np.random.default_rng(5000 + seed)
    L = int(rng.choice([4, 33, 127, 255]))
    d = int(rng.choice([1, 3, 8]))
    n = int(rng.integers(L + d + 8, 60_000))
    fs = float(rng.choice([8.0, 2.4e6, 100e6]))
    f = float(rng.uniform(-0.4, 0.4)) * fs
    x = _c(rng, n)
    taps = _c(rng, L) / max(1, L // 8)
    ring = int(rng.choice([4_096_000, 8 * (L + d + int(rng.integers(8, 3000)))]))
    _both(rr, lambda m: [m.FirFilter(taps, deci=d, translate=(fs, f))], x, ring)      # the DEFAULT rotator: the reference's recurrence
    # RTL-SDR bytes -> fused chain vs the four oracle blocks
    b = rng.integers(0, 256, 2 * int(rng.integers(2000, 80_000)) + int(rng.integers(0, 2)), dtype=np.uint8)
    I, D = int(rng.integers(1, 5)), int(rng.integers(1, 12))
    yo = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], b)
    ro = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(I, D)], b)
    yg = run_chain([rr.FmChainU8(taps, I, D, 1.0)], b, stream_bytes=int(rng.choice([4_096_000, 20_001])))
    assert len(yg) == len(yo)
    if len(yo):
        eps = TOL * float(np.max(np.abs(ro)))
        mag = np.abs(ro.astype(np.complex128))
        bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
        dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
        dd = np.minimum(dd, 2 * np.pi - dd)
        assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_device_rings(rr, seed):
    """random chains over HBM-resident rings == the same chains over host windows, bit for bit"""
    from test_gpu_parity import run_chain_device
    rng = np.random.default_rng(6000 + seed)
    n = int(rng.integers(3000, 120_000))
    x = _c(rng, n)
    L = int(rng.choice([5, 64, 255]))
    taps = _c(rng, L) / max(1, L // 8)
    d = int(rng.choice([1, 2, 8]))
    I, D = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    ring = int(rng.choice([4_096_000, 8 * (2 * L + d + int(rng.integers(600, 4000)))]))
    mk = lambda: [rr.FirFilter(taps, deci=d), rr.RationalResampler(I, D, np.complex64), rr.MultiplyConst(0.5 - 2j, np.complex64),
                  rr.FastFM(), rr.MultiplyConst(1.5)]
    yh = run_chain(mk(), x, stream_bytes=ring)
    yd = run_chain_device(rr, mk(), x, stream_bytes=ring)
    assert len(yh) == len(yd) and np.array_equal(yh, yd)


_FIR_PATHS = [{}, {"fir_path": "direct"}, {"fir_path": "fft", "fir_prune": -1, "fir_half": -1, "fir_poly": -1},
              {"fir_path": "fft", "fir_prune": -1, "fir_poly": -1}, {"fir_path": "fft", "fir_prune": 1, "fir_poly": -1},
              {"fir_poly": 1}]


@pytest.mark.parametrize("seed", range(48))
def test_fuzz_fir_every_path(rr, monkeypatch, seed):
    """Random FirFilter shapes (Complex and Float streams, real and Complex taps, decimations incl. 4 / 8 / 16 and
    other even ones) through every arithmetic path the block can take: automatic choice, direct form, overlap-save
    tiles with a decimating store, half-size inverse, pruned inverse, decimate-first tiles — identical protocol, outputs
    within 1e-5."""
    rng = np.random.default_rng(5000 + seed)
    knob(rr, monkeypatch, **_FIR_PATHS[seed % len(_FIR_PATHS)])
    L = int(rng.choice([1, 5, 16, 33, 64, 127, 200, 255, 401, 600, 601, 1000, 1025, 2049]))
    d = int(rng.choice([1, 2, 2, 4, 4, 6, 8, 8, 10, 16, 16, 22, 3, 5, 7, 12]))
    if seed % len(_FIR_PATHS) == 1 and L > 700:
        L = 257                                       # (the direct form's fallback for long decimating filters is slow)
    n = int(rng.integers(L + d, 200_000))
    real_in = bool(rng.integers(0, 3) == 0)
    es = 4 if real_in else 8
    ring = int(rng.choice([4_096_000, es * (L + d + int(rng.integers(8, 9000)))]))
    if real_in:
        x = rng.uniform(-1, 1, n).astype(np.float32)
        taps = (rng.uniform(-1, 1, L) / max(1, L // 8)).astype(np.float32)
    else:
        x = _c(rng, n)
        taps = _c(rng, L) / max(1, L // 8)
        if rng.integers(0, 2):
            taps = taps.real.astype(np.complex64)
    _both(rr, lambda m: [m.FirFilter(taps, deci=d)], x, ring)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_hilbert_fir_and_even_ratio_chains(rr, monkeypatch, seed):
    """The fused Hilbert -> FirFilter block with random shapes (pruned-inverse tiles for deci 4 / 8 / 16 or direct form)
    and the fused FM chain with 1:even ratios (half-size inverse where the tile allows) against the oracle chains."""
    rng = np.random.default_rng(7000 + seed)
    knob(rr, monkeypatch, fir_prune=1 if seed % 2 else -1)
    hn = int(rng.choice([3, 17, 65, 129]))
    L = int(rng.choice([1, 31, 100, 255, 500, 900]))
    d = int(rng.choice([1, 4, 8, 16, 3, 6]))
    n = int(rng.integers(4 * (L + hn + d), 250_000))
    x = rng.uniform(-1, 1, n).astype(np.float32)
    taps = _c(rng, L) / max(1, L // 8)
    if rng.integers(0, 2):
        taps = taps.real.astype(np.complex64)
    ring = int(rng.choice([4_096_000, 4 * (L + d + int(rng.integers(8, 9000)))]))
    yo = run_chain([orc.Hilbert(hn), orc.FirFilter(taps, deci=d)], x)
    yg = run_chain([rr.HilbertFir(hn, taps, d)], x, stream_bytes=ring)
    assert len(yg) == len(yo)
    if len(yo):
        assert max_norm_err(yg, yo) <= TOL
    # fused chain, interp 1 / even decimation
    Lc = int(rng.choice([101, 300, 463, 800]))
    D = int(rng.choice([2, 4, 6, 10, 50]))
    xc = _c(rng, int(rng.integers(5000, 150_000)))
    tc = _c(rng, Lc) / max(1, Lc // 4)
    yo = run_chain([orc.FftFilter(tc), orc.RationalResampler(1, D), orc.QuadratureDemod(1.0)], xc)
    ro = run_chain([orc.FftFilter(tc), orc.RationalResampler(1, D)], xc)
    yg = run_chain([rr.FmChain(tc, 1, D, 1.0)], xc, stream_bytes=int(rng.choice([4_096_000, 8 * 30_000])))
    assert len(yg) == len(yo)
    if len(yo):
        eps = TOL * float(np.max(np.abs(ro)))
        mag = np.abs(ro.astype(np.complex128))
        bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
        dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
        dd = np.minimum(dd, 2 * np.pi - dd)
        assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fftstream_float_filters_and_multi(rr, seed):
    """FftStream of any size 2..2048 (Bluestein for sizes that are not a power of two), FftFilterFloat / FirFilter<Float>
    of random lengths, and the multi-channel fused chain (Complex and RTL-SDR byte input, even and odd decimations)."""
    rng = np.random.default_rng(9000 + seed)
    size = int(rng.integers(2, 2049))
    x = _c(rng, int(rng.integers(size, 12 * size + 5)))
    _both(rr, lambda m: [m.FftStream(size)], x, int(rng.choice([4_096_000, 8 * (size + int(rng.integers(1, 3 * size)))])))
    L = int(rng.choice([1, 3, 24, 65, 300, 1000, 3584, 3585, 5000]))
    xf = rng.uniform(-1, 1, int(rng.integers(10, 200_000))).astype(np.float32)
    tf = (rng.uniform(-1, 1, L) / max(1, L // 4)).astype(np.float32)
    nsamp = 2 * (1 << int(np.ceil(np.log2(L)))) - L if L > 1 else 1
    _both(rr, lambda m: [m.FftFilterFloat(tf)], xf, int(rng.choice([4_096_000, 4 * (nsamp + int(rng.integers(1, 4000)))])))
    # multi-channel chain
    Lm = int(rng.choice([100, 463, 700, 1500]))
    D = int(rng.choice([2, 5, 6, 8, 30]))
    nch = int(rng.integers(1, 5))
    u8 = bool(rng.integers(0, 2))
    n = int(rng.integers(20_000, 120_000))
    taps = np.stack([_c(rng, Lm) / max(1, Lm // 4) for _ in range(nch)])
    if u8:
        src = rng.integers(0, 256, 2 * n).astype(np.uint8)
        blk = rr.FmMultiU8(taps, 1, D, 1.0)
        front = lambda c: [orc.RtlSdrDecode(), orc.FftFilter(taps[c]), orc.RationalResampler(1, D)]
        cap_in = int(rng.choice([4_096_000, 2 * int(rng.integers(9000, 60000))]))
    else:
        src = _c(rng, n)
        blk = rr.FmMulti(taps, 1, D, 1.0)
        front = lambda c: [orc.FftFilter(taps[c]), orc.RationalResampler(1, D)]
        cap_in = int(rng.choice([512_000, int(rng.integers(9000, 60000))]))
    outs = [[] for _ in range(nch)]
    pos, ring = 0, np.zeros(0, src.dtype)
    while True:
        take = min(cap_in - len(ring), len(src) - pos)
        ring = np.concatenate([ring, src[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, 1_024_000)
        ring = ring[c:]
        out = out.reshape(nch, -1)
        for ch in range(nch):
            if p:
                outs[ch].append(out[ch])
        if take == 0 and c == 0 and p == 0:
            break
    for ch in range(nch):
        yg = np.concatenate(outs[ch]) if outs[ch] else np.zeros(0, np.float32)
        yo = run_chain(front(ch) + [orc.QuadratureDemod(1.0)], src)
        ro = run_chain(front(ch), src)
        assert len(yg) == len(yo)
        if len(yo):
            eps = TOL * float(np.max(np.abs(ro)))
            mag = np.abs(ro.astype(np.complex128))
            bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
            dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
            dd = np.minimum(dd, 2 * np.pi - dd)
            assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_any_size_transforms(rr, seed):
    """FftStream beyond one tile — four-step powers of two, Bluestein for everything else, up to the stream capacity —
    against numpy's f64 FFT with the reference's work() arithmetic (fft_stream.rs:71-117: whole frames of min(in, out)), and
    now and then an FftFilter beyond 16383 taps (overlap-save frames through the same engine) against the oracle."""
    rng = np.random.default_rng(11000 + seed)
    kind = seed % 4
    if kind == 0:
        size = 1 << int(rng.integers(15, 19))
    elif kind == 1:
        size = int(rng.integers(2049, 20_000))
    elif kind == 2:
        size = int(rng.integers(20_000, 200_000))
    else:
        size = int(rng.choice([16385, 32767, 65537, 99_991, 131_071, 262_145, 400_000, 511_999]))
    nfr = int(rng.integers(1, max(2, 512_000 // size + 1)))
    x = _c(rng, nfr * size + int(rng.integers(0, size)))
    cap = int(rng.choice([512_000, size + int(rng.integers(0, 512_000 - size + 1))]))
    b = rr.FftStream(size)
    st, c, p, need, out = b.work(x, cap)
    n = min(len(x), cap)
    n -= n % size
    assert (st, c, p) == (0, n, n)                                   # RR_AGAIN, whole frames only
    ref = np.fft.fft(x[:n].astype(np.complex128).reshape(-1, size), axis=1).reshape(-1)
    assert np.max(np.abs(out - ref)) <= TOL * np.max(np.abs(ref))
    assert b.work(x[:size - 1], cap)[:4] == (1, 0, 0, size)          # WAIT_SRC(size)
    if seed % 4 == 3:
        L = int(rng.integers(16_384, 30_000))
        taps = _c(rng, L) / (L // 8)
        xs = _c(rng, int(rng.integers(60_000, 200_000)))
        _both(rr, lambda m: [m.FftFilter(taps)], xs, int(rng.choice([4_096_000, 8 * (65_536 - L + int(rng.integers(1, 9000)))])))


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_compositions_and_fastfm(rr, seed):
    """Round 3: the fused-chain constructors beyond their tiles (the unfused composition behind the same handle: FmChain /
    FmChainU8 / AudioChain / FmMulti with random tap counts on both sides of every limit, random ratios and ring sizes) and
    FastFM as the chain's demodulator, against the oracle's separate blocks."""
    rng = np.random.default_rng(13000 + seed)
    L = int(rng.choice([300, 3584, 3585, 4094, 4095, 9000, 16383, 16384, 17000]))
    I, D = int(rng.integers(1, 4)), int(rng.integers(1, 9))
    n = int(rng.integers(3 * L, 3 * L + 200_000))
    x = _c(rng, n)
    taps = _c(rng, L) / max(1, L // 4)
    ring = int(rng.choice([4_096_000, 8 * (2 * L + int(rng.integers(100, 40_000)))]))
    kind = seed % 4
    if kind == 0:                                    # QuadratureDemod chain
        yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(0.8)], x, stream_bytes=max(ring, 8 * 140_000))
        ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], x, stream_bytes=max(ring, 8 * 140_000))
        yg = run_chain([rr.FmChain(taps, I, D, 0.8)], x, stream_bytes=max(ring, 8 * 140_000))
        assert len(yg) == len(yo)
        if len(yo):
            eps = TOL * float(np.max(np.abs(ro)))
            mag = np.abs(ro.astype(np.complex128))
            bound = 0.8 * (TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30))
            dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
            dd = np.minimum(dd, 0.8 * 2 * np.pi - dd)
            assert np.all(dd <= bound[:len(dd)])
    elif kind == 1:                                  # FastFM as the demodulator
        yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.FastFM()], x, stream_bytes=max(ring, 8 * 140_000))
        ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], x, stream_bytes=max(ring, 8 * 140_000))
        yg = run_chain([rr.FmChain(taps, I, D, 1.0, rr.DEMOD_FASTFM)], x, stream_bytes=max(ring, 8 * 140_000))
        assert len(yg) == len(yo)
        if len(yo):
            rmax = float(np.max(np.abs(ro)))
            assert np.max(np.abs(yg.astype(np.float64) - yo.astype(np.float64))) <= 8 * TOL * rmax * rmax
    elif kind == 2:                                  # the audio stage
        xf = rng.uniform(-1, 1, n).astype(np.float32)
        tf = (rng.uniform(-1, 1, L) / max(1, L // 8)).astype(np.float32)
        yo = run_chain([orc.FftFilterFloat(tf), orc.RationalResampler(I, D, np.float32), orc.MultiplyConst(0.5)], xf)
        yg = run_chain([rr.AudioChain(tf, I, D, 0.5)], xf)
        assert len(yg) == len(yo)
        if len(yo):
            assert max_norm_err(yg, yo) <= TOL
    else:                                            # two channels on one window
        t2 = np.stack([taps, np.conj(taps)])
        blk = rr.FmMulti(t2, I, D, 1.0)
        outs, ringbuf, pos = [[], []], np.zeros(0, np.complex64), 0
        while True:
            take = min(512_000 - len(ringbuf), n - pos)
            ringbuf = np.concatenate([ringbuf, x[pos:pos + take]]); pos += take
            st, c, p, need, out = blk.work(ringbuf, 1_024_000)
            ringbuf = ringbuf[c:]
            for ch in range(2):
                outs[ch].append(out.reshape(2, -1)[ch])
            if take == 0 and c == 0 and p == 0:
                break
        for ch in range(2):
            yo = run_chain([orc.FftFilter(t2[ch]), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x)
            ro = run_chain([orc.FftFilter(t2[ch]), orc.RationalResampler(I, D)], x)
            yg = np.concatenate(outs[ch])
            assert len(yg) == len(yo)
            if len(yo):
                eps = TOL * float(np.max(np.abs(ro)))
                mag = np.abs(ro.astype(np.complex128))
                bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
                dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
                dd = np.minimum(dd, 2 * np.pi - dd)
                assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fused_chains_any_output_window(rr, seed):
    """Round 4: the fused chains on output windows of ANY size — random tap counts, ratios (interpolating ones included) and
    ring sizes down to a handful of slots, where round 3's blocks waited for ever.  Whole-stream output
    against the oracle's separate blocks; the run must end (run_chain raises otherwise)."""
    from test_gpu_trickle import _drive
    rng = np.random.default_rng(17000 + seed)
    L = int(rng.choice([7, 63, 300, 463, 1200, 2467]))
    I, D = int(rng.integers(1, 6)), int(rng.integers(1, 9))
    n = int(rng.integers(2 * L + 500, 2 * L + 30_000))
    out_cap = int(rng.choice([1, 2, 5, 17, 64, 200, 1500]))
    in_cap = int(rng.choice([512_000, 2 * L + int(rng.integers(10, 5000))]))
    x = _c(rng, n)
    taps = _c(rng, L) / max(1, L // 4)
    kind = seed % 4
    if kind in (0, 1):                               # FmChain, Complex or RTL-SDR byte input
        if kind == 1:
            xb = rng.integers(0, 256, 2 * n, dtype=np.uint8)
            yo = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(0.8)], xb)
            ro = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(I, D)], xb)
            yg, _ = _drive(rr.FmChainU8(taps, I, D, 0.8), xb, 2 * in_cap, out_cap)
        else:
            yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D), orc.QuadratureDemod(0.8)], x)
            ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(I, D)], x)
            yg, _ = _drive(rr.FmChain(taps, I, D, 0.8), x, in_cap, out_cap)
        assert yg.shape[1] == len(yo)
        if len(yo):
            eps = TOL * float(np.max(np.abs(ro)))
            mag = np.abs(ro.astype(np.complex128))
            bound = 0.8 * (TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30))
            dd = np.abs(yg[0].astype(np.float64) - yo.astype(np.float64))
            dd = np.minimum(dd, 0.8 * 2 * np.pi - dd)
            assert np.all(dd <= bound[:len(dd)])
    elif kind == 2:                                  # the audio stage
        xf = rng.uniform(-1, 1, n).astype(np.float32)
        tf = (rng.uniform(-1, 1, L) / max(1, L // 8)).astype(np.float32)
        yo = run_chain([orc.FftFilterFloat(tf), orc.RationalResampler(I, D, np.float32), orc.MultiplyConst(0.5)], xf)
        yg, _ = _drive(rr.AudioChain(tf, I, D, 0.5), xf, 2 * in_cap, out_cap)
        assert yg.shape[1] == len(yo)
        if len(yo):
            assert max_norm_err(yg[0], yo) <= TOL
    else:                                            # three channels on one window
        t3 = np.stack([taps, np.conj(taps), taps[::-1].copy()])
        yg, _ = _drive(rr.FmMulti(t3, I, D, 1.0), x, in_cap, out_cap)
        for c in range(3):
            yo = run_chain([orc.FftFilter(t3[c]), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x)
            ro = run_chain([orc.FftFilter(t3[c]), orc.RationalResampler(I, D)], x)
            assert yg.shape[1] == len(yo)
            if len(yo):
                eps = TOL * float(np.max(np.abs(ro)))
                mag = np.abs(ro.astype(np.complex128))
                bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
                dd = np.abs(yg[c].astype(np.float64) - yo.astype(np.float64))
                dd = np.minimum(dd, 2 * np.pi - dd)
                assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_round4_paths(rr, seed):
    """Round 4 widened the path selection: decimate-first tiles for every decimation up to 16 and up to 768 / 928
    taps per phase, tiles for large decimations, FirFilter<Float> beyond 3584 taps, long Hilbert transformers, the two-stage
    HilbertFir, FmMulti up to 1:11.  Random shapes over those ranges, windows large enough for the large-window kernels AND
    small rings, against the oracle blocks."""
    rng = np.random.default_rng(19000 + seed)
    kind = seed % 6
    if kind == 0:                                    # FirFilter<Complex>, any decimation, long filters
        L = int(rng.choice([31, 127, 700, 1500, 2467, 3599, 4600, 6000]))
        d = int(rng.integers(1, 41))
        n = int(rng.integers(L + d + 200_000, L + d + 900_000))
        x = _c(rng, n)
        taps = _c(rng, L) / max(1, L // 8)
        if rng.integers(0, 2):
            taps = taps.real.astype(np.complex64)
        ring = int(rng.choice([4_096_000 * 4, 8 * (L + d + int(rng.integers(8, 20_000)))]))
        _both(rr, lambda m: [m.FirFilter(taps, deci=d)], x, ring)
    elif kind == 1:                                  # FirFilter<Float> on both sides of 3584 taps
        L = int(rng.choice([31, 300, 3584, 3585, 4100, 5000, 7000]))
        d = int(rng.integers(1, 41))
        n = int(rng.integers(L + d + 100_000, L + d + 700_000))
        x = rng.uniform(-1, 1, n).astype(np.float32)
        taps = (rng.uniform(-1, 1, L) / max(1, L // 8)).astype(np.float32)
        ring = int(rng.choice([4_096_000 * 4, 4 * (L + d + int(rng.integers(8, 20_000)))]))
        _both(rr, lambda m: [m.FirFilter(taps, deci=d)], x, ring)
    elif kind == 2:                                  # Hilbert, long transformers
        hn = int(rng.choice([129, 199, 201, 255, 999, 2047, 3583, 3585, 6001]))
        n = int(rng.integers(hn + 50_000, hn + 600_000))
        x = rng.uniform(-1, 1, n).astype(np.float32)
        ring = int(rng.choice([4_096_000 * 4, 4 * (2 * hn + int(rng.integers(100, 30_000)))]))
        w = int(rng.integers(0, 3))
        _both(rr, lambda m: [m.Hilbert(hn, w)], x, ring)
    elif kind == 3:                                  # HilbertFir: composite, pruned (down to a quarter of a tile) or two stages
        hn = int(rng.choice([31, 65, 129]))
        L = int(rng.choice([33, 255, 700, 1000, 1400, 2467]))
        d = int(rng.integers(1, 41))
        n = int(rng.integers(L + d + hn + 100_000, L + d + hn + 600_000))
        x = rng.uniform(-1, 1, n).astype(np.float32)
        taps = _c(rng, L) / max(1, L // 8)
        ring = int(rng.choice([4_096_000 * 4, 4 * (L + d + int(rng.integers(8, 20_000)))]))
        yo = run_chain([orc.Hilbert(hn), orc.FirFilter(taps, deci=d)], x)
        yg = run_chain([rr.HilbertFir(hn, taps, d)], x, stream_bytes=ring)
        assert len(yg) == len(yo) and len(yo)
        assert max_norm_err(yg, yo) <= TOL
    else:                                            # fused chains: 1:2 ... 1:16, long phases; one chain or three channels
        D = int(rng.integers(2, 17))
        L = int(rng.choice([127, 463, 1000, 2467, D * 500, D * 760, min(16000, D * 900)]))
        n = int(rng.integers(3 * L + 200_000, 3 * L + 700_000))
        x = _c(rng, n)
        multi = kind == 5
        nch = 3 if multi else 1
        taps = np.stack([_c(rng, L) / max(1, L // 4) for _ in range(nch)])
        blk = rr.FmMulti(taps, 1, D, 1.0) if multi else rr.FmChain(taps[0], 1, D, 1.0)
        outs, ringbuf, pos = [[] for _ in range(nch)], np.zeros(0, np.complex64), 0
        cap_in = int(rng.choice([2_000_000, 2 * L + int(rng.integers(1000, 60_000))]))
        for _ in range(1_000_000):
            take = min(cap_in - len(ringbuf), n - pos)
            ringbuf = np.concatenate([ringbuf, x[pos:pos + take]]); pos += take
            st, c, p, need, out = blk.work(ringbuf, 1_000_000)
            ringbuf = ringbuf[c:]
            out = np.atleast_2d(out)
            for ch in range(nch):
                outs[ch].append(out[ch])
            if take == 0 and c == 0 and p == 0:
                break
        for ch in range(nch):
            yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D), orc.QuadratureDemod(1.0)], x)
            ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, D)], x)
            yg = np.concatenate(outs[ch])
            assert len(yg) == len(yo)
            if len(yo):
                eps = TOL * float(np.max(np.abs(ro)))
                mag = np.abs(ro.astype(np.complex128))
                bound = TOL * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
                dd = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
                dd = np.minimum(dd, 2 * np.pi - dd)
                assert np.all(dd <= bound[:len(dd)])


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_zero_copy_windows(rr, seed):
    """Round 4: a block on PAGE-LOCKED windows (kernels read / write them in place over PCIe, windows start at odd, moving
    offsets like a ring's) against the same block on pageable windows (staged through device memory, the path every other
    test compares with the oracle): same protocol log, same bits — random block, taps, ratio and window sizes."""
    rng = np.random.default_rng(9000 + seed)
    L = int(rng.choice([1, 3, 17, 65, 127, 200, 463, 1000, 2467, 5000]))
    I, D = int(rng.integers(1, 7)), int(rng.integers(1, 17))
    tc = (_c(rng, L) / max(1, L // 4)).astype(np.complex64)
    if rng.integers(0, 2):
        tc = tc.real.astype(np.complex64)
    tf = (rng.uniform(-1, 1, L) / max(1, L // 4)).astype(np.float32)
    n = int(rng.integers(2 * L + 64, int(rng.choice([2 * L + 20_000, 400_000]))))
    xc, xf = _c(rng, n), rng.uniform(-1, 1, n).astype(np.float32)
    xb = rng.integers(0, 256, 2 * n + int(rng.integers(0, 2)), dtype=np.uint8)
    hn = int(rng.choice([3, 33, 65, 129, 301, 1001])) | 1
    kinds = [
        (lambda: rr.FirFilter(tc, deci=D), xc), (lambda: rr.FirFilter(tf, deci=D), xf), (lambda: rr.FftFilter(tc), xc),
        (lambda: rr.FftFilterFloat(tf), xf), (lambda: rr.RationalResampler(I, D, np.complex64), xc),
        (lambda: rr.RationalResampler(I, D, np.float32), xf), (lambda: rr.QuadratureDemod(0.5), xc), (lambda: rr.Hilbert(hn), xf),
        (lambda: rr.HilbertFir(hn, tc, D), xf), (lambda: rr.FmChain(tc, I, D, 1.0), xc), (lambda: rr.FmChainU8(tc, I, D, 1.0), xb),
        (lambda: rr.AudioChain(tf, I, D, 0.5), xf), (lambda: rr.FmMulti(np.stack([tc, np.conj(tc)]), I, D, 1.0), xc),
        (lambda: rr.RtlSdrDecode(), xb), (lambda: rr.FirFilter(tc, deci=D, translate=(2.4e6, 1.3e5)), xc),
    ]
    for k in rng.choice(len(kinds), 4, replace=False):
        mk, x = kinds[int(k)]
        try:
            blk = mk()
        except Exception:
            continue                                        # shape refused by the constructor: nothing to compare
        # windows: at least what the block may ask for in one go (a filter block of the longest filter), at most ~3e5 elements
        floor_in = 4 * (1 << int(np.ceil(np.log2(max(L, hn, 2))))) * (2 if x.dtype == np.uint8 else 1) + 64
        in_cap = int(rng.integers(floor_in, floor_in + 300_000))
        out_cap = int(rng.choice([int(rng.integers(1, 64)), int(rng.integers(64, 5000)), in_cap * I + 16]))
        out_cap = max(out_cap, n * I // 1500 + 1)           # (bounds the number of calls of a trial)
        try:
            ya, la = drive_pageable(mk(), x, in_cap, out_cap)
        except AssertionError:
            continue                                        # a window pair the block can never move through ("no termination")
        yb, lb = drive_registered(rr, mk(), x, in_cap, out_cap)
        assert la == lb, (blk.name, seed)
        assert ya.shape == yb.shape, (blk.name, seed)
        assert np.array_equal(ya.view(np.uint8), yb.view(np.uint8)), (blk.name, seed)


@pytest.mark.parametrize("seed", list(range(24)) + [10074])       # (10074: found by round 6's soak — the head fix of the fused
def test_fuzz_nonfinite_sets(rr, seed):                            #  FirFilter -> FftFilter cut a poisoned tile's run short of the probe stride)
    """Round 5: random filters, stream lengths, ring sizes and NaN / +-Inf positions (isolated, clustered, at the very ends,
    none at all) — the outputs that are not finite are EXACTLY the reference's for FirFilter (its ntaps windows), FftFilter /
    FftFilterFloat / the fused FirFilter -> FftFilter (the fft_size outputs from the start of the reference's block, across
    work() calls) and Hilbert, on
    pageable windows and on page-locked ones, and everything else stays within tolerance."""
    rng = np.random.default_rng(9000 + seed)
    for _ in range(3):
        kind = int(rng.integers(0, 6))
        L = int(rng.choice([1, 2, 5, 17, 64, 127, 128, 129, 401, 1000, 1025, 2467, 5000]))
        if kind == 4:
            L = int(rng.choice([3, 33, 65, 129, 301])) | 1
        n = int(rng.integers(3 * L + 100, 3 * L + int(rng.choice([3_000, 60_000, 300_000]))))
        real = kind in (1, 3, 4)
        x = rng.uniform(-1, 1, n).astype(np.float32) if real else _c(rng, n)
        nbad = int(rng.choice([0, 1, 2, 5, 12]))
        pos = [int(p) for p in rng.integers(0, n, nbad)]
        if nbad and rng.integers(0, 2):
            pos += [0, n - 1, pos[0] + 1 if pos[0] + 1 < n else 0]
        for p in pos:
            bad = [np.nan, np.inf, -np.inf][int(rng.integers(0, 3))]
            x[p] = bad if real else [complex(bad, 0.5), complex(0.25, bad), complex(bad, bad)][int(rng.integers(0, 3))]
        tc = (_c(rng, L) / max(1, L // 4)).astype(np.complex64)
        tf = (rng.uniform(-1, 1, L) / max(1, L // 4)).astype(np.float32)
        d = int(rng.choice([1, 1, 2, 3, 5, 8]))
        t1 = (_c(rng, int(rng.choice([1, 2, 9, 33, 127]))) / 8).astype(np.complex64)      # (kind 5: the FirFilter fused in front)
        mk = [lambda m: [m.FirFilter(tc, deci=d)], lambda m: [m.FirFilter(tf, deci=d)], lambda m: [m.FftFilter(tc)],
              lambda m: [m.FftFilterFloat(tf)], lambda m: [m.Hilbert(L)],
              lambda m: [m.FirFftFilter(t1, tc)] if m is rr else [m.FirFilter(t1), m.FftFilter(tc)]][kind]
        block = 4 * (1 << int(np.ceil(np.log2(max(L, 2)))))       # (a ring holds two of the reference's transform blocks)
        es = 4 if real else 8
        ring = int(rng.choice([4_096_000, es * (block + int(rng.integers(64, 20_000)))]))
        yo = run_chain(mk(orc), x, stream_bytes=ring)
        if rng.integers(0, 3) == 0:
            cap = ring // es
            yg = drive_registered(rr, mk(rr)[0], x, cap, cap)[0][0]
        else:
            yg = run_chain(mk(rr), x, stream_bytes=ring)
        assert len(yo) == len(yg), (kind, L, n, ring)
        fin = lambda y: np.isfinite(y.real) & np.isfinite(y.imag) if np.iscomplexobj(y) else np.isfinite(y)
        go, gg = fin(yo), fin(yg)
        assert np.array_equal(go, gg), (kind, L, n, d, ring, sorted(pos)[:6], np.flatnonzero(go != gg)[:6])
        if go.any():
            assert max_norm_err(yg[go], yo[go]) <= TOL, (kind, L, n, ring)


@pytest.mark.parametrize("seed", list(range(24)) + [10236, 20429])     # (found by the soak: the evidence of a poisoned block in the NEXT block's outputs)
def test_fuzz_chain_nan_sets(rr, seed):
    """Round 6: the fused chains (FmChain, FirFmChain, FmMulti, AudioChain) on random filters, ratios, stream lengths, ring sizes
    and NaN positions (isolated, clustered, at the ends, either side of block boundaries, none at all; whole samples and
    single components) — the outputs that are NaN are EXACTLY those of the reference's three blocks (FftFilter's blocks through
    the resampler's index map and the demodulator's pair, across work() calls), everything else within the chain tolerance."""
    from harness import angle_parity
    rng = np.random.default_rng(11000 + seed)
    for _ in range(3):
        kind = int(rng.integers(0, 4))                  # 0 FmChain, 1 FirFmChain, 2 FmMulti, 3 AudioChain
        L = int(rng.choice([2, 17, 64, 127, 128, 463, 1000, 2467]))
        I, D = [(1, 6), (1, 4), (1, 10), (2, 3), (3, 2), (25, 128), (1, 1), (1, 17)][int(rng.integers(0, 8))]
        n = int(rng.integers(6 * L + 2_000, 6 * L + int(rng.choice([20_000, 120_000, 400_000]))))
        Sref = (1 << int(np.ceil(np.log2(max(2 * L, 4))))) - L + 1        # (about the reference's nsamples: where block boundaries are)
        nbad = int(rng.choice([0, 1, 2, 5, 12]))
        pos = [int(p) for p in rng.integers(0, n, nbad)]
        if nbad and rng.integers(0, 2):
            pos += [0, n - 1, min(n - 1, pos[0] + 1), min(n - 1, (pos[0] // Sref + 1) * Sref), max(0, (pos[0] // Sref) * Sref - 1)]
        block = 4 * (1 << int(np.ceil(np.log2(max(L, 2)))))
        if kind == 3:
            x = rng.uniform(-1, 1, n).astype(np.float32)
            for p in pos:
                x[p] = np.nan
            tf = (rng.uniform(-1, 1, L) / max(1, L // 4)).astype(np.float32)
            ring = int(rng.choice([4_096_000, 4 * (block + int(rng.integers(64, 20_000)))]))
            yo = run_chain([orc.FftFilterFloat(tf), orc.RationalResampler(I, D, np.float32), orc.MultiplyConst(0.5)], x, stream_bytes=ring)
            yg = run_chain([rr.AudioChain(tf, I, D, 0.5)], x, stream_bytes=ring)
            assert len(yo) == len(yg), (kind, L, I, D, n, ring)
            bo, bg = ~np.isfinite(yo), ~np.isfinite(yg)
            assert np.array_equal(bo, bg), (kind, L, I, D, n, ring, sorted(pos)[:6], np.flatnonzero(bo != bg)[:6])
            if (~bo).any():
                assert max_norm_err(yg[~bo], yo[~bo]) <= TOL, (kind, L, I, D, n, ring)
            continue
        ph = np.cumsum(0.2 * np.sin(2 * np.pi * 1e-3 * np.arange(n)))
        x = (np.exp(1j * ph) + 0.02 * _c(rng, n)).astype(np.complex64)        # |r| away from 0: the plain bound nearly everywhere
        for p in pos:
            x[p] = [complex(np.nan, 0.5), complex(0.25, np.nan), complex(np.nan, np.nan)][int(rng.integers(0, 3))]
        k = np.arange(L)
        tc = (np.hamming(L) * np.sinc((k - (L - 1) / 2) * 0.2) * 0.2).astype(np.complex64)    # a low-pass: the carrier passes
        t1 = (_c(rng, int(rng.choice([1, 2, 9, 33]))) / 4).astype(np.complex64)
        ring = int(rng.choice([4_096_000, 8 * (block + int(rng.integers(64, 20_000)))]))
        if kind == 2:
            t3 = np.stack([tc, (tc * np.exp(1j * 0.05 * k)).astype(np.complex64)])
            blk = rr.FmMulti(t3, I, D, 1.0)
            cap = ring // 8
            yg2, _ = drive_pageable(blk, x, cap, cap)
            chans = [(None, t3[c], yg2[c]) for c in range(2)]
        else:
            blk = rr.FmChain(tc, I, D, 1.0) if kind == 0 else rr.FirFmChain(t1, tc, I, D, 1.0)
            chans = [(t1 if kind == 1 else None, tc, run_chain([blk], x, stream_bytes=ring))]
        for front, t, got in chans:
            pre = [] if front is None else [orc.FirFilter(front)]
            want = run_chain(pre + [orc.FftFilter(t), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x, stream_bytes=ring)
            ro = run_chain(pre + [orc.FftFilter(t), orc.RationalResampler(I, D)], x, stream_bytes=ring)
            assert len(want) >= len(got) and (len(got) == len(want) or kind == 2), (kind, L, I, D, n, ring, len(want), len(got))
            if len(got) == 0:
                continue
            want, ro = want[:len(got)], ro[:len(got) + 1]
            bo, bg = ~np.isfinite(want), ~np.isfinite(got)
            assert np.array_equal(bo, bg), (kind, L, I, D, n, ring, sorted(pos)[:6], int(bo.sum()), int(bg.sum()), np.flatnonzero(bo != bg)[:6])
            ro_ok = np.where(np.isfinite(ro.real) & np.isfinite(ro.imag), ro, 1.0)
            par = angle_parity(np.where(bo, 0.0, got), np.where(bo, 0.0, want), ro_ok)
            assert par["used"] <= 1.0, (kind, L, I, D, n, ring, par)
