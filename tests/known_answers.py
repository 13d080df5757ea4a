"""The reference's own unit tests for the hot path, restated once and run against
any implementation `impl` that exposes the block constructors (oracle.pyoracle for the
CPU oracle, rustradio_amd for the HIP product).  Data comes from
tests/golden/reference_known_answers.json; each check cites the reference test.

Like the reference's tests, a check builds source -> block by hand, calls
work() explicitly, asserts the BlockRet variant and compares at the reference's own
tolerance (1e-3 absolute, src/lib.rs:846-878) unless the vector is marked exact."""
from __future__ import annotations

import numpy as np

from harness import (AGAIN, WAIT_DST, WAIT_SRC, DEFAULT_STREAM_SIZE, cplx, golden, run_chain,
                     signal_source_complex, signal_source_complex_fast)

G = golden()
SIX = cplx(G["six_complex_input"])
TOL = 1e-3


def almost(a, b, tol=TOL):
    a = np.asarray(a); b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size:
        assert float(np.max(np.abs(a.astype(np.complex128) - b.astype(np.complex128)))) <= tol, (a, b)


def is_wait(st):
    return st in (WAIT_SRC, WAIT_DST)


# ---- FIR (src/fir.rs tests) ---------------------------------------------------------
def check_fir_test_complex(impl):
    g = G["fir_test_complex"]
    taps = cplx(g["taps"])
    for deci, key in ((1, "expect_deci1"), (2, "expect_deci2")):
        b = impl.FirFilter(taps, deci=deci)
        st, c, p, need, out = b.work(SIX, 100)
        assert st == AGAIN
        almost(out, cplx(g[key]))


def check_fir_test_identity(impl):
    g = G["fir_test_identity"]
    inp = np.concatenate([SIX, SIX])
    for deci in range(g["deci_range"][0], g["deci_range"][1] + 1):
        b = impl.FirFilter(cplx(g["taps"]), deci=deci)
        res = np.zeros(0, np.complex64)
        buf = inp
        if deci <= 2 * len(SIX):
            st, c, p, need, out = b.work(buf, 1000)
            assert st == AGAIN, deci
            buf = buf[c:]; res = out
        st, c, p, need, out = b.work(buf, 1000)
        assert is_wait(st) and c == 0 and p == 0, deci
        mx = 2 * len(SIX) // deci
        almost(res, inp[::deci][:mx])


def check_fir_test_invert(impl):
    g = G["fir_test_invert"]
    for deci in range(1, len(SIX) + 2):
        b = impl.FirFilter(cplx(g["taps"]), deci=deci)
        res = np.zeros(0, np.complex64); buf = SIX
        if deci <= len(SIX):
            st, c, p, need, out = b.work(buf, 1000)
            assert st == AGAIN
            buf = buf[c:]; res = out
        st, c, p, need, out = b.work(buf, 1000)
        assert is_wait(st)
        almost(res, -SIX[::deci][:len(SIX) // deci])


def check_fir_moving_avg(impl):
    g = G["fir_moving_avg"]
    full = cplx(g["full"])
    for deci in range(1, len(SIX) + 2):
        b = impl.FirFilter(cplx(g["taps"]), deci=deci)
        res = np.zeros(0, np.complex64); buf = SIX
        if deci < len(SIX):
            st, c, p, need, out = b.work(buf, 1000)
            assert st == AGAIN
            buf = buf[c:]; res = out
        st, c, p, need, out = b.work(buf, 1000)
        assert is_wait(st)
        almost(res, full[::deci][:(len(SIX) - 1) // deci])


def check_fir_translate_matches_mixed_input(impl):
    g = G["fir_translate_matches_mixed_input"]
    n = g["n_input"]
    inp = (np.arange(n, dtype=np.float32) + 1j * (np.arange(n, dtype=np.float32) * np.float32(0.25))).astype(np.complex64)
    taps = cplx(g["taps"])
    fs, f, deci = g["samp_rate"], g["freq"], g["deci"]
    # manual mixing exactly as the reference test does (f32 rotator, :758-768)
    step = -2.0 * np.pi * f / fs
    rot = np.complex64(complex(np.float32(np.cos(step)), np.float32(np.sin(step))))
    phase = np.complex64(1.0)
    mixed = np.zeros(n, np.complex64)
    for i in range(n):
        mixed[i] = np.complex64(inp[i] * phase)
        phase = np.complex64(phase * rot)
    tr = impl.FirFilter(taps, deci=deci, translate=(fs, f))
    st, c, p, need, out_t = tr.work(inp, 1000)
    assert st == AGAIN
    st2, *_ = tr.work(inp[c:], 1000)
    assert is_wait(st2)
    mn = impl.FirFilter(taps, deci=deci)
    st, c, p, need, out_m = mn.work(mixed, 1000)
    assert st == AGAIN
    assert len(out_t) == len(out_m) == (n - len(taps) + 1) // deci
    almost(out_t, out_m)


def _tone(n, fs, f):
    ph = (2.0 * np.pi * float(np.float32(f)) / float(np.float32(fs))) * np.arange(n, dtype=np.float64)
    return (np.cos(ph).astype(np.float32) + 1j * np.sin(ph).astype(np.float32)).astype(np.complex64)


def check_fir_translated_tone(impl):
    g = G["fir_translated_tone"]
    fs, f = g["samp_rate"], g["freq"]
    taps = impl.low_pass_complex(fs, g["cutoff"], g["twidth"], impl.WIN_HAMMING)
    for inp, cond in ((_tone(g["n"], fs, f), lambda m: m > g["pass_mean_gt"]),
                      (np.ones(g["n"], np.complex64), lambda m: m < g["dc_mean_lt"])):
        b = impl.FirFilter(taps, translate=(fs, f))
        st, c, p, need, out = b.work(inp, 100000)
        assert st == AGAIN
        st2, *_ = b.work(inp[c:], 100000)
        assert is_wait(st2)
        assert len(out) == g["n"] - len(taps) + 1
        assert cond(float(np.mean(np.abs(out))))


# ---- tap designers -------------------------------------------------------------------
def check_taps_filter_generator(impl):
    g = G["taps_test_filter_generator"]
    taps = impl.low_pass_complex(g["samp_rate"], g["cutoff"], g["twidth"], impl.WIN_HAMMING)
    assert len(taps) == g["ntaps"]
    almost(taps, np.asarray(g["expect"], np.float32).astype(np.complex64))
    assert np.all(taps.imag == 0)


def check_windows(impl):
    g = G["window_doctest"]
    w = impl.make_window(impl.WIN_HAMMING, g["ntaps"])
    assert len(w) == g["ntaps"]
    assert np.all(np.abs(w - np.asarray(g["expect"], np.float32)) < g["tol"])
    for wt in (impl.WIN_BLACKMAN, impl.WIN_BLACKMAN_HARRIS, impl.WIN_HAMMING):
        assert list(impl.make_window(wt, 1)) == G["window_one_tap"]["expect"]


# ---- FftFilter (src/fft_filter.rs tests) ---------------------------------------------
def check_fftfilter_filter_a_signal(impl):
    g = G["fftfilter_filter_a_signal"]
    fs = g["samp_rate"]
    taps = impl.low_pass_complex(fs, g["cutoff"], g["twidth"], impl.WIN_HAMMING)
    taps_len = len(taps)
    b = impl.FftFilter(taps)
    # SignalSourceComplex -> Head(8000): one window of 8000 samples reaches the filter.
    x = signal_source_complex(fs, g["signal"], g["amplitude"], int(fs))
    st, c, p, need, out = b.work(x, DEFAULT_STREAM_SIZE // 8)
    assert st == WAIT_SRC
    assert c == len(x)                       # all input is taken into buf (fft_filter.rs:306-314)
    assert p > 0 and p % (512 - taps_len) == 0
    m = float(np.max(np.abs(out[taps_len:])))
    assert 0.0 <= m < g["max_mag_lt"], m


def check_fftfilter_tag_propagation(impl):
    g = G["fftfilter_tag_propagation"]
    b = impl.FftFilter(cplx(g["taps"]))
    x = np.zeros(g["n_zeros"] * g["repeats"], np.complex64)
    st, c, p, need, out = b.work(x, DEFAULT_STREAM_SIZE // 8)
    assert st == WAIT_SRC and need == 1
    assert p == g["expect_out_len"] and c == len(x)
    assert np.all(out == 0)


# ---- RationalResampler (src/rational_resampler.rs tests) ------------------------------
def check_resampler_deci(impl):
    for deci in range(1, len(SIX) + 2):
        b = impl.RationalResampler(1, deci, np.complex64)
        st, c, p, need, out = b.work(SIX, 1000)
        assert is_wait(st)
        assert c == len(SIX)
        assert np.array_equal(out, SIX[::deci])


def check_resampler_examples(impl):
    for key in ("resampler_example64", "resampler_example128"):
        g = G[key]
        b = impl.RationalResampler(g["interp"], g["deci"], np.uint32)
        st, c, p, need, out = b.work(np.arange(g["n_input"], dtype=np.uint32), 100000)
        assert is_wait(st)
        assert list(out) == g["expect"]


def check_resampler_full_output_buffer(impl):
    g = G["resampler_full_output_buffer"]
    cap = g["output_capacity"]
    assert cap % 3 == 1
    boundary = cap // 3
    inp = np.arange(boundary + 1, dtype=np.uint32)
    b = impl.RationalResampler(g["interp"], g["deci"], np.uint32)
    st, c, p, need, first = b.work(inp, cap)
    assert st == WAIT_DST and need == 1
    assert len(first) == cap and first[cap - 1] == boundary
    assert not b.eof(True)                    # pending blocks EOF (rational_resampler.rs:209-213)
    inp = inp[c:]
    st, c, p, need, second = b.work(inp, cap)
    assert is_wait(st) and need == 1
    assert list(second) == [boundary, boundary]
    assert b.eof(True)


def check_resampler_chained(impl):
    g = G["resampler_chained"]
    inp = np.arange(g["n_input"], dtype=np.uint32)
    p1 = run_chain([impl.RationalResampler(*g["direct"], np.uint32)], inp)
    p2 = run_chain([impl.RationalResampler(*g["chain"][0], np.uint32),
                    impl.RationalResampler(*g["chain"][1], np.uint32)], inp)
    assert len(p1) == len(p2)
    assert np.max(np.abs(p1.astype(np.int64) - p2.astype(np.int64))) < g["max_abs_diff_lt"]


def check_resampler_rates(impl):
    for n, interp, deci, final in G["resampler_rates"]["cases"]:
        b = impl.RationalResampler(interp, deci, np.complex64)
        inp = np.arange(n, dtype=np.float32).astype(np.complex64)
        st, c, p, need, out = b.work(inp, 100000)
        assert len(out) == final, (n, interp, deci, final, len(out))


def check_resampler_rejects_zero(impl):
    import pytest
    with pytest.raises(ValueError):
        impl.RationalResampler(0, 1, np.complex64)
    with pytest.raises(ValueError):
        impl.RationalResampler(1, 0, np.complex64)


# ---- QuadratureDemod (src/quadrature_demod.rs tests) -----------------------------------
def check_quad_known(impl, mode=None):
    mode = impl.ATAN2_EXACT if mode is None else mode
    for key in ("quad_nulls", "quad_cw", "quad_ccw"):
        g = G[key]
        b = impl.QuadratureDemod(1.0, mode)
        st, c, p, need, out = b.work(cplx(g["input"]), 100)
        assert st == WAIT_SRC and need == 2 and c == 3
        exp = np.asarray(g["expect"], np.float32)
        if g.get("exact"):
            assert np.array_equal(out, exp)
        else:
            almost(out, exp)


def check_quad_fill_out(impl):
    g = G["quad_fill_out"]
    cap_in = DEFAULT_STREAM_SIZE // 8
    cap_out = DEFAULT_STREAM_SIZE // 4
    b = impl.QuadratureDemod(1.0, impl.ATAN2_EXACT)
    ring = signal_source_complex_fast(g["samp_rate"], g["freq"], g["amplitude"], cap_in)
    assert len(ring) == 512_000
    out_len = 0
    for expect in g["expect_lens"]:
        st, c, p, need, out = b.work(ring, cap_out - out_len)
        out_len += p
        assert out_len == expect
        ring = ring[c:]
        ring = np.concatenate([ring, signal_source_complex_fast(g["samp_rate"], g["freq"], g["amplitude"], cap_in - len(ring))])


# ---- RtlSdrDecode (src/rtlsdr_decode.rs tests) -------------------------------------------
def check_rtlsdr_decode(impl):
    b = impl.RtlSdrDecode()                                      # empty (:54-64)
    st, c, p, need, out = b.work(np.zeros(0, np.uint8), 100)
    assert is_wait(st) and p == 0
    g = G["rtlsdr_some_input"]                                   # some_input: exact f32 equality
    st, c, p, need, out = impl.RtlSdrDecode().work(np.asarray(g["input"], np.uint8), 100)
    assert st == WAIT_SRC and need == 2 and c == 6
    exp = np.asarray([complex(*e) for e in g["expect"]], np.complex64)
    assert np.array_equal(out, exp), (out, exp)
    g = G["rtlsdr_uneven"]                                       # uneven: the odd byte stays
    st, c, p, need, out = impl.RtlSdrDecode().work(np.asarray(g["input"], np.uint8), 100)
    assert st == WAIT_SRC and p == g["expect_len"] and c == 2 * g["expect_len"]
    g = G["rtlsdr_overflow"]                                     # overflow: 4x expansion, 4 rounds
    b = impl.RtlSdrDecode()
    ring = np.zeros(g["input_bytes"], np.uint8)
    for _ in range(g["rounds"]):
        st, c, p, need, out = b.work(ring, DEFAULT_STREAM_SIZE // 8)
        assert is_wait(st) and p == g["expect_len_each"] and c == 2 * p
        ring = ring[c:]
    st, c, p, need, out = b.work(ring, DEFAULT_STREAM_SIZE // 8)
    assert len(ring) == 0 and st == WAIT_SRC and p == 0


ALL_CHECKS = [
    check_fir_test_complex, check_fir_test_identity, check_fir_test_invert, check_fir_moving_avg,
    check_fir_translate_matches_mixed_input, check_fir_translated_tone,
    check_taps_filter_generator, check_windows,
    check_fftfilter_filter_a_signal, check_fftfilter_tag_propagation,
    check_resampler_deci, check_resampler_examples, check_resampler_full_output_buffer,
    check_resampler_chained, check_resampler_rates, check_resampler_rejects_zero,
    check_quad_known, check_quad_fill_out,
    check_rtlsdr_decode,
]
