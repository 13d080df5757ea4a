"""CPU: bench_verify's float64 restatement of the chains (what bench.py checks its timed outputs against) agrees with the
oracle's blocks — every stage kind, incl. the translated FIR's f32 recurrences and the RTL-SDR decode."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench_verify  # noqa: E402
import rustradio_amd as rr  # noqa: E402  (tap designers only: host code, no GPU)
from harness import run_chain  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402


def _angle_err(a, b):
    d = np.abs(a - b)
    return np.minimum(d, 2 * np.pi - d)


def test_complex_chain_fir_fft_resampler_demod():
    rng = np.random.default_rng(5)
    x = (rng.uniform(-1, 1, 40_000) + 1j * rng.uniform(-1, 1, 40_000)).astype(np.complex64)
    t1 = rr.low_pass_complex(10e6, 1e6, 190e3)
    t2 = rr.low_pass_complex(10e6, 1e6, 60e3)
    stages = [("fir", t1, 1), ("fft", t2), ("rs", 3, 7), ("demod", 0.5)]
    ch = bench_verify._Chain(stages, lambda a, b: x[a:b].astype(np.complex128))
    yo = run_chain([orc.FirFilter(t1), orc.FftFilter(t2)], x)
    assert np.max(np.abs(ch.get(1, 0, len(yo)) - yo)) / np.max(np.abs(yo)) < 1e-5 and len(yo) > 30_000
    yo = run_chain([orc.FirFilter(t1), orc.FftFilter(t2), orc.RationalResampler(3, 7), orc.QuadratureDemod(0.5)], x)
    assert np.median(_angle_err(ch.get(3, 0, len(yo)), yo)) < 1e-6 and len(yo) > 10_000
    # a range in the middle equals the same range of the whole evaluation (the lazy index arithmetic)
    whole = ch.get(3, 0, len(yo))
    assert np.array_equal(ch.get(3, 5000, 5256), whole[5000:5256])


def test_real_chain_hilbert_fir_and_translate():
    rng = np.random.default_rng(6)
    xr = rng.uniform(-1, 1, 30_000).astype(np.float32)
    t3 = rr.low_pass_complex(100e6, 5e6, 943e3)
    src = lambda a, b: xr[a:b].astype(np.float64)          # noqa: E731
    yo = run_chain([orc.Hilbert(65), orc.FirFilter(t3, 8)], xr)
    ch = bench_verify._Chain([("hilbert", 65), ("fir", t3, 8)], src)
    assert np.max(np.abs(ch.get(1, 0, len(yo)) - yo)) / np.max(np.abs(yo)) < 1e-5 and len(yo) > 3000
    yo = run_chain([orc.Hilbert(65), orc.FirFilter(t3, 8, translate=(100e6, 9.375e6))], xr)
    ch = bench_verify._Chain([("hilbert", 65), ("fir_translate", t3, 8, 100e6, 9.375e6, True)], src)
    assert np.max(np.abs(ch.get(1, 0, len(yo)) - yo)) / np.max(np.abs(yo)) < 1e-5
    # the replayed rotator is the oracle's, bit for bit
    taps, ph = bench_verify.rotated_taps_and_phases(t3, 8, 100e6, 9.375e6, 64)
    assert np.array_equal(taps, orc.fir_translated_taps(orc.FirFilter(t3, 8, translate=(100e6, 9.375e6))))


def test_rtlsdr_bytes_chain():
    rng = np.random.default_rng(7)
    b = rng.integers(0, 256, 20_000, dtype=np.uint8)
    taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
    yo = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(25, 128), orc.QuadratureDemod(1.0)], b)

    def get(a, bb):
        v = (b[2 * a:2 * bb].astype(np.float32) - np.float32(127.0)) * np.float32(0.008)
        return v[0::2].astype(np.float64) + 1j * v[1::2]
    ch = bench_verify._Chain([("u8",), ("fft", taps), ("rs", 25, 128), ("demod", 1.0)], get)
    assert np.median(_angle_err(ch.get(3, 0, len(yo)), yo)) < 1e-6 and len(yo) > 1500
