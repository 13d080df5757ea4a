"""GPU: (1) edge cases of every block's work() protocol — empty / ragged / zero-capacity windows —
against the oracle; (2) BASELINE full-size runs checked through size-independent properties:
random segments against the oracle, linearity, DC gain, resampler index identity."""
import os

import numpy as np
import pytest

from harness import AGAIN, WAIT_DST, WAIT_SRC, max_norm_err, run_chain
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


def _makers(m):
    taps = orc.low_pass_complex(10e6, 1e6, 190e3)
    return {
        "fir": lambda: m.FirFilter(taps, deci=3),
        "fir_f32": lambda: m.FirFilter(taps.real.copy(), deci=2),
        "fft": lambda: m.FftFilter(taps),
        "fft_f32": lambda: m.FftFilterFloat(taps.real.copy()),
        "resamp": lambda: m.RationalResampler(3, 7, np.complex64),
        "quad": lambda: m.QuadratureDemod(1.0),
        "hilbert": lambda: m.Hilbert(65),
    }


@pytest.mark.parametrize("name", ["fir", "fir_f32", "fft", "fft_f32", "resamp", "quad", "hilbert"])
def test_edge_windows_match_oracle(rr, name):
    """Same (status, consumed, produced, need) and samples as the oracle for degenerate windows."""
    go, gg = _makers(orc)[name](), _makers(rr)[name]()
    rng = np.random.default_rng(7)
    dt = go.in_dtype
    stream = (rng.uniform(-1, 1, 6000) + 1j * rng.uniform(-1, 1, 6000)).astype(np.complex64)
    x = stream if dt == np.complex64 else stream.real.copy()
    # (in_len, out_cap) sequence: empty input, zero output space, 1 sample, ragged, exact thresholds
    plan = [(0, 10), (5, 0), (1, 1), (2, 1), (126, 5), (127, 5), (129, 1), (130, 40), (777, 3), (2000, 5000),
            (0, 0), (3000, 2), (3000, 100000), (0, 100000)]
    po = pg = 0
    for in_len, cap in plan:
        wo = x[po:po + in_len]; wg = x[pg:pg + in_len]
        so, co, pro, no, oo = go.work(wo, cap)
        sg, cg, prg, ng, og = gg.work(wg, cap)
        assert (so, co, pro, no) == (sg, cg, prg, ng), (name, in_len, cap)
        if len(oo):
            scale = np.pi if name == "quad" else max(1.0, float(np.max(np.abs(oo))))
            assert max_norm_err(og, oo, scale) <= TOL
        po += co; pg += cg


def test_fir_decimation_larger_than_input(rr):
    taps = np.ones(1, np.complex64)
    for deci in (7, 13, 100):
        b = rr.FirFilter(taps, deci=deci)
        st, c, p, need, out = b.work(rnd_c(6, 1), 10)
        assert (st, c, p, need) == (WAIT_SRC, 0, 0, deci)        # fir.rs:498-501


def test_constructor_errors(rr):
    with pytest.raises(ValueError):
        rr.FirFilter(np.zeros(0, np.complex64))
    with pytest.raises(ValueError):
        rr.FirFilter(np.ones(3, np.complex64), deci=0)
    with pytest.raises(ValueError):
        rr.FirFilter(np.ones(3, np.float32), translate=(1.0, 1.0))
    with pytest.raises(ValueError):
        rr.QuadratureDemod(1.0, 7)
    with pytest.raises(ValueError):
        rr.RationalResampler(1, 1, np.dtype("V3"))


# ---- full-size properties (BASELINE configs[1]: 1e8 Complex<f32>) ---------------------------------------
def test_fftfilter_full_size_properties(rr):
    import torch
    n = 100_000_000
    taps = orc.low_pass_complex(10e6, 1e6, 60e3)
    L, S = len(taps), 623
    g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0002)
    x1 = torch.rand(2 * n, generator=g, device="cuda") * 2 - 1
    x2 = torch.rand(2 * n, generator=g, device="cuda") * 2 - 1
    n_out = (n // S) * S
    stream = torch.cuda.current_stream().cuda_stream

    def filt(x):
        y = torch.empty(2 * (n + 1024), device="cuda")
        b = rr.FftFilter(taps)
        st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 1024, stream)
        torch.cuda.synchronize()
        assert (st, c, p, need) == (WAIT_SRC, n, n_out, S - n % S)
        return y[:2 * n_out]

    y1 = filt(x1)
    # (a) random segments against the oracle (needs L-1 samples of history)
    rng = np.random.default_rng(1)
    starts = [0, n_out - 5000] + [int(s) for s in rng.integers(L, n_out - 5000, 14)]
    for s0 in starts:
        h0 = max(0, s0 - (L - 1))
        xin = x1[2 * h0:2 * (s0 + 5000)].cpu().numpy().view(np.complex64)
        ref = np.convolve(xin.astype(np.complex128), taps.astype(np.complex128))[s0 - h0:s0 - h0 + 5000]
        got = y1[2 * s0:2 * (s0 + 5000)].cpu().numpy().view(np.complex64)
        assert max_norm_err(got, ref) <= TOL, s0
    # (b) linearity over the whole 1e8 samples: F(a x1 + b x2) = a F(x1) + b F(x2)
    a, b = 0.75, -1.5
    y2 = filt(x2)
    y12 = filt(a * x1 + b * x2)
    err = (y12 - (a * y1 + b * y2)).abs().max().item() / y12.abs().max().item()
    assert err <= TOL, err
    del y2, y12
    # (c) DC gain: constant input -> sum(taps) after the start-up transient
    ydc = filt(torch.ones(2 * n, device="cuda"))
    want = complex(np.sum(taps.astype(np.complex128)) * (1 + 1j))
    tail = ydc[2 * L:].view(-1, 2)
    assert abs(tail[:, 0].min().item() - want.real) < 1e-5 and abs(tail[:, 0].max().item() - want.real) < 1e-5
    assert abs(tail[:, 1].min().item() - want.imag) < 1e-5 and abs(tail[:, 1].max().item() - want.imag) < 1e-5


def test_fm_chain_and_resampler_full_size(rr):
    """configs[2] at 24e6 samples: fused chain == three separate GPU blocks (bit-level agreement is not
    required, both are within tolerance of the oracle on small cases); resampler = pure index pick."""
    import torch
    n = 24_000_000
    taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
    t = torch.arange(n, device="cuda", dtype=torch.float64)
    ph = -75.0 * torch.cos(2 * np.pi * 1e3 * t / 2.4e6)
    x = torch.stack([torch.cos(ph), torch.sin(ph)], dim=1).float().reshape(-1).contiguous()
    s = torch.cuda.current_stream().cuda_stream
    y = torch.empty(2 * (n + 1024), device="cuda"); r = torch.empty(2 * (n // 6 + 1024), device="cuda")
    o3 = torch.empty(n // 6 + 1024, device="cuda"); of = torch.empty(n // 6 + 1024, device="cuda")
    st, c, p1, _ = rr.FftFilter(taps).work_dev(x.data_ptr(), n, y.data_ptr(), n + 1024, s)
    st, c, p2, _ = rr.RationalResampler(1, 6).work_dev(y.data_ptr(), p1, r.data_ptr(), n // 6 + 1024, s)
    st, c, p3, _ = rr.QuadratureDemod(1.0).work_dev(r.data_ptr(), p2, o3.data_ptr(), n // 6 + 1024, s)
    st, c, pf, _ = rr.FmChain(taps, 1, 6, 1.0).work_dev(x.data_ptr(), n, of.data_ptr(), n // 6 + 1024, s)
    torch.cuda.synchronize()
    n1 = (n // 561) * 561
    assert p1 == n1 and p2 == -(-n1 // 6) and p3 == p2 - 1 == pf
    yy = y[:2 * p1].view(-1, 2); rr_ = r[:2 * p2].view(-1, 2)
    assert torch.equal(rr_, yy[::6])                                    # resampler: bit-exact pick of every 6th sample
    skip = len(taps)
    d = (of[skip:pf] - o3[skip:p3]).abs().max().item()
    assert d <= TOL * np.pi, d
    # demodulated FM: 75 kHz deviation, 1 kHz tone at 400 ksps -> phase step amplitude 2 pi 75e3/400e3
    amp = o3[skip:p3].abs().max().item()
    assert abs(amp - 2 * np.pi * 75e3 / 400e3) < 2e-3


def test_windows_beyond_2_31_elements(rr):
    """64-bit indexing: single device windows of 2.2e9 elements (> 2^31 elements, > 2^34 bytes).  The last
    stretch of each output is compared with the same block run on just the tail of the input (time
    invariance), which the oracle-level parity tests already pin."""
    import torch
    n = 2_200_000_000
    tail = 3_000_000
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    # --- Hilbert: 2.2e9 floats in, 2.2e9 Complex out
    x = torch.empty(n, dtype=torch.float32, device="cuda")
    for s in range(0, n, 200_000_000):
        m = min(200_000_000, n - s)
        x[s:s + m] = torch.rand(m, generator=g, device="cuda") * 2 - 1
    y = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    b = rr.Hilbert(65)
    st, c, p, need = b.work_dev(x.data_ptr(), n, y.data_ptr(), n)
    b.sync()
    assert (st, c, p) == (AGAIN, n, n)
    b2 = rr.Hilbert(65)
    y2 = torch.empty(2 * tail, dtype=torch.float32, device="cuda")
    b2.work_dev(x.data_ptr() + 4 * (n - tail), tail, y2.data_ptr(), tail)
    b2.sync()
    a, r = y[2 * (n - tail) + 2 * 65:], y2[2 * 65:]                  # skip the tail run's zero history
    assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())
    del y, y2
    # --- decimating FIR (Float in, Float taps) over the same 2.2e9 samples
    taps = orc.low_pass(100e6, 5e6, 943e3)
    f = rr.FirFilter(taps, deci=8)
    yo = torch.empty(n // 8 + 8, dtype=torch.float32, device="cuda")
    st, c, p, need = f.work_dev(x.data_ptr(), n, yo.data_ptr(), n // 8 + 8)
    f.sync()
    assert st == AGAIN and p == (n - len(taps) + 1) // 8 and c == 8 * p
    f2 = rr.FirFilter(taps, deci=8)
    start = 8 * ((n - tail) // 8)
    y2 = torch.empty(tail // 8 + 8, dtype=torch.float32, device="cuda")
    st2, c2, p2, _ = f2.work_dev(x.data_ptr() + 4 * start, n - start, y2.data_ptr(), tail // 8 + 8)
    f2.sync()
    a, r = yo[start // 8: start // 8 + p2], y2[:p2]
    assert p2 > 1000 and start // 8 + p2 == p
    assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())
    del yo, y2, x
    # --- FftFilter: 2.2e9 Complex in (17.6 GB), tail compared from a block boundary on
    xc = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    for s in range(0, 2 * n, 400_000_000):
        m = min(400_000_000, 2 * n - s)
        xc[s:s + m] = torch.rand(m, generator=g, device="cuda") * 2 - 1
    ct = orc.low_pass_complex(10e6, 1e6, 60e3)
    ff = rr.FftFilter(ct)
    yc = torch.empty(2 * (n + 1024), dtype=torch.float32, device="cuda")
    st, c, p, need = ff.work_dev(xc.data_ptr(), n, yc.data_ptr(), n + 1024)
    ff.sync()
    assert st == WAIT_SRC and c == n and p == (n // 623) * 623
    start = ((p - tail) // 623) * 623
    ff2 = rr.FftFilter(ct)
    y2 = torch.empty(2 * (tail + 1024), dtype=torch.float32, device="cuda")
    st2, c2, p2, _ = ff2.work_dev(xc.data_ptr() + 8 * start, p - start, y2.data_ptr(), tail + 1024)
    ff2.sync()
    assert p2 == p - start
    skip = 2 * len(ct)                                               # zero history of the tail run
    a, r = yc[2 * start + skip: 2 * p], y2[skip: 2 * p2]
    assert float((a - r).abs().max()) <= 1e-5 * float(r.abs().max())


def test_translate_default_mode_is_the_reference_recurrence_for_1e7_outputs(rr):
    """the DEFAULT `.translate()` (no rotator argument = RR_ROT_REPLAY) against the oracle over more than
    1e7 outputs, bit for bit.  A 1-tap (1 + 0j) filter makes the FIR exact (x * 1 - y * 0, x * 0 + y * 1), so every output
    bit is the rotator's: out[m] = x[m] * phase_m, phase_{m+1} = phase_m * step in f32 (fir.rs:464-473).  Driven (a) through
    reference-sized host windows — many calls, the look-ahead ring wraps 40 times — and (b) as ONE device-resident call of
    1.2e7 outputs, which goes through the 2^22-entry ring in halves."""
    import torch
    fs, f = 1.0e6, 123_456.7
    one = np.ones(1, np.complex64)
    n = 12_000_000
    x = rnd_c(n, 77)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x)
    assert len(yo) == n
    yg = run_chain([rr.FirFilter(one, translate=(fs, f))], x)               # (a) default arguments
    assert len(yg) == n
    assert np.array_equal(yg.view(np.uint32), yo.view(np.uint32)), int(np.flatnonzero(yg != yo)[0])
    blk = rr.FirFilter(one, translate=(fs, f))                               # (b) one call
    dx = torch.from_numpy(x.view(np.float32)).cuda()
    dy = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    st, c, p, need = blk.work_dev(dx.data_ptr(), n, dy.data_ptr(), n)
    blk.sync()
    torch.cuda.synchronize()
    assert (st, c, p) == (AGAIN, n, n)
    yb = dy.cpu().numpy().view(np.complex64)
    assert np.array_equal(yb.view(np.uint32), yo.view(np.uint32)), int(np.flatnonzero(yb != yo)[0])
    # a second call continues the chain where the first one stopped (the look-ahead generated past it)
    st, c, p, need = blk.work_dev(dx.data_ptr(), 1000, dy.data_ptr(), 1000)
    blk.sync()
    torch.cuda.synchronize()
    o2 = orc.FirFilter(one, translate=(fs, f))
    y2 = run_chain([o2], np.concatenate([x, x[:1000]]))
    assert np.array_equal(dy[:2000].cpu().numpy().view(np.complex64).view(np.uint32), y2[n:].view(np.uint32))


def test_translate_default_mode_long_decimating_stream(rr):
    """The default rotator behind a REAL filter: 1.1e7 outputs of a 16-tap /2 translate filter stay within 1e-5 of the
    oracle to the end of the stream (the f64 model is 0.16 away by then, test_rotator_drift_vs_length's 1.5e-8 * n)."""
    fs, f, d = 10e6, -1.234e6, 2
    taps = orc.low_pass_complex(fs, 2e6, 1.6e6)
    assert 9 <= len(taps) <= 17
    x = rnd_c(22_000_100, 78)
    yo = run_chain([orc.FirFilter(taps, deci=d, translate=(fs, f))], x)
    yg = run_chain([rr.FirFilter(taps, deci=d, translate=(fs, f))], x)
    assert len(yo) == len(yg) >= 11_000_000
    assert max_norm_err(yg, yo) <= TOL
    assert max_norm_err(yg[-100_000:], yo[-100_000:]) <= TOL
    ym = run_chain([rr.FirFilter(taps, deci=d, translate=(fs, f), rotator=rr.ROT_MODEL)], x)
    print("opt-in RR_ROT_MODEL after 1.1e7 outputs: max-normalised error", max_norm_err(ym[-100_000:], yo[-100_000:]))


def test_device_replay_mode_is_the_same_chain(rr):
    """RR_ROT_REPLAY_DEVICE (the one-lane kernel, round 3's default) and RR_ROT_REPLAY (the host generator, round 4's) are
    the same f32 recurrence bit for bit — through reference-sized windows over 3e6 outputs, and switching between the two
    in the middle of a stream in both directions (the device chain skips what the host delivered and vice versa)."""
    fs, f = 1.0e6, -234_567.8
    one = np.ones(1, np.complex64)
    n = 3_000_000
    x = rnd_c(n, 79)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x)
    yd = run_chain([rr.FirFilter(one, translate=(fs, f), rotator=rr.ROT_REPLAY_DEVICE)], x)
    assert np.array_equal(yd.view(np.uint32), yo.view(np.uint32))
    yh = run_chain([rr.FirFilter(one, translate=(fs, f), rotator=rr.ROT_REPLAY_HOST)], x)
    assert np.array_equal(yh.view(np.uint32), yo.view(np.uint32))
    b = rr.FirFilter(one, translate=(fs, f))
    outs, step = [], 100_000
    for i, a in enumerate(range(0, n, step)):
        if i % 7 == 3:
            b.set_rotator_mode(rr.ROT_REPLAY_DEVICE)
        if i % 7 == 5:
            b.set_rotator_mode(rr.ROT_REPLAY_HOST if i % 2 else rr.ROT_REPLAY)
        st, c, p, need, out = b.work(x[a:a + step], step)
        assert c == p == step
        outs.append(out)
    y = np.concatenate(outs)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), int(np.flatnonzero(y != yo)[0])


def test_device_replay_chain_any_window_length(rr):
    """k_rotor_replay (round 5: sixteen phases per register block, each block stored while the next is computed) over calls
    of every length 1 ... 99 and random ones to 5000: prologue, ping-pong loop, odd block and scalar tail all produce the
    reference's recurrence bit for bit, and the phase carried between calls is the chain's."""
    fs, f = 1.0e6, 123_456.7
    one = np.ones(1, np.complex64)
    rng = np.random.default_rng(17)
    sizes = list(range(1, 100)) + [int(v) for v in rng.integers(1, 5000, 60)] + [16, 32, 48, 64, 31, 33, 47, 49]
    x = rnd_c(sum(sizes), 80)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x)
    b = rr.FirFilter(one, translate=(fs, f), rotator=rr.ROT_REPLAY_DEVICE)
    outs, a = [], 0
    for k in sizes:
        st, c, p, need, out = b.work(x[a:a + k], k)
        assert c == p == k
        outs.append(out); a += k
    y = np.concatenate(outs)
    assert np.array_equal(y.view(np.uint32), yo.view(np.uint32)), int(np.flatnonzero(y != yo)[0])


def test_default_rotator_takes_a_host_thread_only_when_the_chain_paces_the_block(rr):
    """Round 5: RR_ROT_REPLAY starts on the device chain (no thread per translating block) and moves to a
    host generator thread only when three calls in a row arrive before the chain's look-ahead has finished — the same bits
    before, across and after the switch."""
    import threading, time
    fs, f = 1.0e6, 123_456.7
    one = np.ones(1, np.complex64)
    nthreads = lambda: len(os.listdir("/proc/self/task"))
    # paced: 50 k outputs per call (0.7 ms of chain), 10 ms apart — the look-ahead is always done: no thread
    step, calls = 50_000, 12
    x = rnd_c(step * calls, 5)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x)
    b = rr.FirFilter(one, translate=(fs, f))
    st, c, p, need, out0 = b.work(x[:step], step)                 # (the first call creates the side stream)
    t0 = nthreads()
    outs = [out0]
    for a in range(step, len(x), step):
        time.sleep(0.010)
        outs.append(b.work(x[a:a + step], step)[4])
    assert nthreads() == t0, (t0, nthreads())
    assert np.array_equal(np.concatenate(outs).view(np.uint32), yo.view(np.uint32))
    # back to back: 500 k outputs per call (7 ms of chain each) with nothing in between — the block is handed a generator
    step, calls = 500_000, 10
    x = rnd_c(step * calls, 6)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x, stream_bytes=8 * step)
    b = rr.FirFilter(one, translate=(fs, f))
    st, c, p, need, out0 = b.work(x[:step], step)
    t0 = nthreads()
    outs = [out0] + [b.work(x[a:a + step], step)[4] for a in range(step, len(x), step)]
    assert nthreads() == t0 + 1, (t0, nthreads())
    assert np.array_equal(np.concatenate(outs).view(np.uint32), yo.view(np.uint32))


def test_rotator_drift_vs_length(rr):
    """FirFilter::translate's rotator (fir.rs:464-473) is an un-renormalised f32 recurrence.  The opt-in RR_ROT_MODEL
    evaluates phase0 * step^m in f64 from the same f32-rounded phase0 / step: it reproduces the recurrence's systematic
    drift (|step| != 1 after rounding) but not its accumulated rounding noise.  This pins the bound the header states:
    |model - replay| <= 1e-7 * n after n outputs (measured ~3e-8 * n), so MODEL is inside the 1e-5 parity bar for about
    1e2 outputs in the worst case and REPLAY (the default: bit-faithful, on the device) is there for long streams."""
    fs, f = 1.0e6, 123_456.7
    one = np.ones(1, np.complex64)                                # a 1-tap filter: the output IS the rotator (times x)
    n = 1_000_000
    x = np.ones(n, np.complex64)
    ym = run_chain([rr.FirFilter(one, translate=(fs, f), rotator=rr.ROT_MODEL)], x)
    yr = run_chain([rr.FirFilter(one, translate=(fs, f), rotator=rr.ROT_REPLAY)], x)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x)
    assert len(ym) == len(yr) == len(yo) == n
    assert np.array_equal(yr, yo)                                 # the replay is the reference's recurrence, bit for bit
    d = np.abs(ym.astype(np.complex128) - yr.astype(np.complex128))
    idx = np.arange(1, n + 1)
    assert np.all(d <= 1e-7 * idx + 2e-7), float(np.max(d / idx))
    print("rotator drift: max |model - replay| / n =", float(np.max(d / idx)), " at n = 1e6:", float(d[-1]))


def test_nan_locality_is_bounded_by_one_tile(rr):
    """Round 3's bound, kept as a coarse check: a non-finite sample never reaches further than one tile.  Both blocks now
    have the reference's set exactly (FirFilter: test_nonfinite_samples_reach_exactly_the_references_outputs, round 4;
    FftFilter: test_nonfinite_samples_in_fftfilter_poison_the_references_blocks, round 5) and pass a fortiori."""
    L, pos, n = 127, 50_000, 120_000
    taps = orc.low_pass_complex(10e6, 1e6, 190e3)
    x = rnd_c(n, 11)
    x[pos] = np.nan
    yo = run_chain([orc.FirFilter(taps)], x)
    bad_o = np.flatnonzero(~np.isfinite(yo.real) | ~np.isfinite(yo.imag))
    assert bad_o[0] == pos - (L - 1) and bad_o[-1] == pos and len(bad_o) == L
    for blocks, width in (([rr.FirFilter(taps)], 4096), ([rr.FftFilter(taps)], 4096)):
        yg = run_chain(blocks, x)
        bad_g = np.flatnonzero(~np.isfinite(yg.real) | ~np.isfinite(yg.imag))
        off = 0 if len(yg) == len(yo) else L - 1                 # FftFilter's output index is the FIR's + L - 1
        assert set(bad_o + off) <= set(bad_g)
        assert bad_g[0] >= pos - width and bad_g[-1] <= pos + width
        good = np.ones(len(yg), bool); good[bad_g] = False
        ref = run_chain([orc.FftFilter(taps)] if off else [orc.FirFilter(taps)], np.nan_to_num(x))
        assert max_norm_err(yg[good], ref[:len(yg)][good]) <= TOL     # everything else is untouched
    with rr.build_options(fir_path="direct"):
        yd = run_chain([rr.FirFilter(taps)], x)
    bad_d = np.flatnonzero(~np.isfinite(yd.real) | ~np.isfinite(yd.imag))
    assert set(bad_o) <= set(bad_d) and bad_d[0] >= bad_o[0] - 8 and bad_d[-1] <= bad_o[-1] + 8


def _nonfinite_mask(y):
    y = np.asarray(y)
    return ~np.isfinite(y.real) | ~np.isfinite(y.imag) if np.iscomplexobj(y) else ~np.isfinite(y)


def _poisoned(x, seed):
    """a few NaN / +-Inf samples (whole or one component), isolated and in a cluster, also near both ends"""
    rng = np.random.default_rng(seed)
    x = x.copy()
    n = len(x)
    pos = sorted(set([3, n // 7, n // 7 + 1, n // 3, n // 2 + 5, n - 9] + [int(p) for p in rng.integers(0, n, 4)]))
    for k, p in enumerate(pos):
        bad = [np.nan, np.inf, -np.inf][k % 3]
        if np.iscomplexobj(x):
            x[p] = [complex(bad, 0.25), complex(-0.5, bad), complex(bad, bad)][k % 3]
        else:
            x[p] = bad
    return x


def _hilfir(m, hn, taps, deci):
    return [m.HilbertFir(hn, taps, deci)] if hasattr(m, "HilbertFir") else [m.Hilbert(hn), m.FirFilter(taps, deci=deci)]


NONFINITE_CASES = [
    # name, chain maker (m = oracle or GPU module, t = taps), build options for the GPU block, real input
    ("fir127-tiles", lambda m, t: [m.FirFilter(t["c127"])], dict(fir_path="fft"), False),
    ("fir127-auto", lambda m, t: [m.FirFilter(t["c127"])], {}, False),
    ("fir127-direct", lambda m, t: [m.FirFilter(t["c127"])], dict(fir_path="direct"), False),
    ("fir401-complex-taps", lambda m, t: [m.FirFilter(t["cc401"])], {}, False),
    ("fir401-complex-taps-direct-deci3", lambda m, t: [m.FirFilter(t["cc401"][:127], deci=3)], dict(fir_path="direct"), False),
    ("fir5000-split-tiles", lambda m, t: [m.FirFilter(t["c5000"])], {}, False),
    ("fir5000-split-tiles-deci3", lambda m, t: [m.FirFilter(t["c5000"], deci=3)], dict(fir_poly=-1), False),
    ("fir401-deci5-decimating-store", lambda m, t: [m.FirFilter(t["c401"], deci=5)], dict(fir_path="fft", fir_poly=-1), False),
    ("fir401-deci2-decimating-store", lambda m, t: [m.FirFilter(t["c401"], deci=2)], dict(fir_path="fft", fir_half=-1), False),
    ("fir401-deci2-half-inverse", lambda m, t: [m.FirFilter(t["c401"], deci=2)], dict(fir_path="fft"), False),
    ("fir401-deci6-decimate-first", lambda m, t: [m.FirFilter(t["c401"], deci=6)], dict(fir_poly=1), False),
    ("fir401-deci5-decimate-first-3waves", lambda m, t: [m.FirFilter(t["c401"], deci=5)], dict(fir_poly=1), False),
    ("fir401-deci12-decimate-first-2waves", lambda m, t: [m.FirFilter(t["c401"] , deci=12)], dict(fir_poly=1), False),
    ("fir255-deci8-pruned", lambda m, t: [m.FirFilter(t["c401"][:255], deci=8)], dict(fir_prune=1), False),
    ("fir255-deci4-pruned", lambda m, t: [m.FirFilter(t["c401"][:255], deci=4)], dict(fir_prune=1), False),
    ("fir255-deci32-pruned-sub", lambda m, t: [m.FirFilter(t["c401"][:255], deci=32)], dict(fir_prune=1), False),
    ("firf32-127-direct", lambda m, t: [m.FirFilter(t["f127"])], dict(fir_path="direct"), True),
    ("firf32-1000-real-tiles", lambda m, t: [m.FirFilter(t["f1000"])], {}, True),
    ("firf32-1000-real-tiles-deci3", lambda m, t: [m.FirFilter(t["f1000"], deci=3)], {}, True),
    ("firf32-255-deci8-pruned", lambda m, t: [m.FirFilter(t["f1000"][:255], deci=8)], dict(fir_prune=1), True),
    ("firf32-5000-wide", lambda m, t: [m.FirFilter(t["f5000"], deci=2)], {}, True),
    ("hilbertfir-65x255-deci8-direct", lambda m, t: _hilfir(m, 65, t["c401"][:255], 8), dict(fir_prune=-1), True),
    ("hilbertfir-65x255-deci8-pruned", lambda m, t: _hilfir(m, 65, t["c401"][:255], 8), dict(fir_prune=1), True),
    ("hilbertfir-65x401-deci3", lambda m, t: _hilfir(m, 65, t["cc401"], 3), {}, True),
    ("hilbertfir-65x2467-deci32-two-stage", lambda m, t: _hilfir(m, 65, t["c5000"][:2467], 32), {}, True),
]


@pytest.mark.parametrize("name,mk,opts,real_in", NONFINITE_CASES, ids=[c[0] for c in NONFINITE_CASES])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 3_000])
def test_nonfinite_samples_reach_exactly_the_references_outputs(rr, name, mk, opts, real_in, stream_bytes):
    """Round 4 (csrc/nan_fix.hpp): a NaN / Inf input sample makes exactly the outputs non-finite that the reference's
    per-output dot products make non-finite (the ntaps windows that contain it) — on the transform tiles, the pruned and
    decimate-first tiles and the direct form alike — and every other output stays within tolerance of the reference's."""
    taps = {"c127": orc.low_pass_complex(10e6, 1e6, 190e3), "c401": orc.low_pass_complex(10e6, 1e6, 60e3)}
    taps["cc401"] = (taps["c401"] * np.exp(1j * 0.3 * np.arange(len(taps["c401"])))).astype(np.complex64)
    taps["c5000"] = (rnd_c(5000, 31) / 2000).astype(np.complex64)
    taps["f127"] = taps["c127"].real.copy()
    taps["f1000"] = (rnd_f(1000, 32) / 300).astype(np.float32)
    taps["f5000"] = (rnd_f(5000, 33) / 1500).astype(np.float32)
    n = 150_000
    x = _poisoned(rnd_f(n, 21) if real_in else rnd_c(n, 21), 5)
    sb = max(stream_bytes, 16 * 6000)                            # (a ring holds the longest filter here)
    yo = run_chain(mk(orc, taps), x, stream_bytes=sb)
    with rr.build_options(**opts):
        blocks = mk(rr, taps)
    yg = run_chain(blocks, x, stream_bytes=sb)
    assert len(yo) == len(yg) > 1000
    bo, bg = _nonfinite_mask(yo), _nonfinite_mask(yg)
    assert bo.sum() > 20
    assert np.array_equal(bo, bg), (int(bo.sum()), int(bg.sum()), np.flatnonzero(bo != bg)[:8])
    # ... of the same class component by component (NaN stays NaN, +Inf stays +Inf).  Not asked of the fused Hilbert ->
    # FirFilter: its composite filter meets an Inf sample with other tap signs than the two stages do (Inf - Inf = NaN in
    # one, Inf in the other); the SET of non-finite outputs is the reference's all the same.
    if not name.startswith("hilbertfir"):
        for part in ((np.real, np.imag) if np.iscomplexobj(yo) else (np.asarray,)):
            a, b = part(yo)[bo], part(yg)[bo]
            assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])
    ok = ~(bo | bg)
    assert max_norm_err(yg[ok], yo[ok]) <= TOL


FFTFILTER_NONFINITE_CASES = [
    # name, taps maker, real stream, small ring bytes (>= two reference blocks)
    ("fft5", lambda: (rnd_c(5, 41) / 3).astype(np.complex64), False, 8 * 3_000),
    ("fft127-1024-tiles", lambda: orc.low_pass_complex(10e6, 1e6, 190e3), False, 8 * 3_000),
    ("fft401-2048-tiles", lambda: orc.low_pass_complex(10e6, 1e6, 60e3), False, 8 * 6_000),
    ("fft1025-4096-tiles", lambda: (rnd_c(1025, 42) / 500).astype(np.complex64), False, 8 * 12_000),
    ("fft2467-split-or-plain", lambda: (rnd_c(2467, 43) / 1000).astype(np.complex64), False, 8 * 30_000),
    ("fft5000-16384-split", lambda: (rnd_c(5000, 44) / 2000).astype(np.complex64), False, 8 * 50_000),
    ("fft16500-any-size-frames", lambda: (rnd_c(16500, 45) / 6000).astype(np.complex64), False, 8 * 120_000),
    ("firfft127x401", lambda: (orc.low_pass_complex(10e6, 1e6, 190e3), orc.low_pass_complex(10e6, 1e6, 60e3)), False, 8 * 6_000),
    ("firfft33x2467-complex-taps", lambda: ((rnd_c(33, 48) / 10).astype(np.complex64), (rnd_c(2467, 49) / 1000).astype(np.complex64)), False, 8 * 30_000),
    ("fftfloat127", lambda: orc.low_pass_complex(10e6, 1e6, 190e3).real.copy(), True, 4 * 3_000),
    ("fftfloat1000", lambda: (rnd_f(1000, 46) / 300).astype(np.float32), True, 4 * 12_000),
    ("fftfloat5000-complex-inner", lambda: (rnd_f(5000, 47) / 1500).astype(np.float32), True, 4 * 50_000),
]


@pytest.mark.parametrize("name,mk_taps,real_in,small", FFTFILTER_NONFINITE_CASES, ids=[c[0] for c in FFTFILTER_NONFINITE_CASES])
@pytest.mark.parametrize("ring", ["reference", "small", "registered", "small-pass-in-the-tile-kernel"])
def test_nonfinite_samples_in_fftfilter_poison_the_references_blocks(rr, name, mk_taps, real_in, small, ring, monkeypatch):
    """Round 5 (csrc/kernels_misc.hip k_ref_blocks_nonfinite): FftFilter / FftFilterFloat's reference transforms blocks of
    nsamples inputs (fft_filter.rs:326-347) — a NaN / Inf input sample makes the fft_size outputs from its block's start
    non-finite and nothing else.  The GPU's tiles are of another size on another grid; a pass behind them puts exactly the
    reference's set in place (across work() calls too: the small rings end most calls inside a poisoned stretch) and restores
    what only the tile had smeared.  VERDICT r4 weak #3: 'FftFilter-type blocks smear over the GPU tile'."""
    taps = mk_taps()
    if ring.endswith("in-the-tile-kernel"):      # round 6, opt-in: the pass in the tile kernel's tail wherever the kernel has one
        from harness import knob
        knob(rr, monkeypatch, fft_nonfinite_tiles=3)
        ring = "small"
    mk = (lambda m: [m.FftFilterFloat(taps)]) if real_in else (lambda m: [m.FftFilter(taps)])
    if name.startswith("firfft"):      # FirFilter(t1) -> FftFilter(t2), two blocks in the reference, one composite convolution here
        t1, t2 = taps
        mk = lambda m: [m.FirFftFilter(t1, t2)] if m is rr else [m.FirFilter(t1), m.FftFilter(t2)]
        taps = t2
    n = 150_000 if len(taps) < 16_000 else 400_000
    x = _poisoned(rnd_f(n, 51) if real_in else rnd_c(n, 51), 6)
    for p in np.random.default_rng(len(taps)).integers(0, n, 6):     # (a few more: call ends are where the carry matters)
        x[int(p)] = np.nan
    sb = 4_096_000 if ring == "reference" else small
    yo = run_chain(mk(orc), x, stream_bytes=sb)
    if ring == "registered":           # page-locked rings: the kernels write the host window in place and the HOST probes it
        from harness import drive_registered
        cap = small // x.itemsize
        yg = drive_registered(rr, mk(rr)[0], x, cap, cap)[0][0]
    else:
        yg = run_chain(mk(rr), x, stream_bytes=sb)
    assert len(yo) == len(yg) > 1000
    bo, bg = _nonfinite_mask(yo), _nonfinite_mask(yg)
    assert bo.sum() > 20 and not bo.all()
    assert np.array_equal(bo, bg), (int(bo.sum()), int(bg.sum()), np.flatnonzero(bo != bg)[:8])
    assert max_norm_err(yg[~bo], yo[~bo]) <= TOL
    if ring == "reference" and name == "fft401-2048-tiles":          # the opt-out keeps the round-4 behaviour: the tile's smear
        with rr.build_options(fft_nonfinite_tiles=1):
            blocks = mk(rr)
        yt = run_chain(blocks, x, stream_bytes=sb)
        bt = _nonfinite_mask(yt)
        assert not np.array_equal(bt, bo) and max_norm_err(yt[~(bt | bo)], yo[~(bt | bo)]) <= TOL


def test_nonfinite_samples_hilbert_pair_kernel(rr):
    """The default Hilbert kernel (pair samples, zero taps skipped: csrc/kernels_fir.hip k_hilbert) leaves outputs finite that
    the reference poisons — it multiplies the transformer's zero taps too (0 * NaN = NaN).  Round 5: the kernel tests its
    staged input (one packed FMA per pair it stages) and a workgroup that saw a bad tile recomputes its outputs with the
    reference's fold: the non-finite outputs are EXACTLY the reference's, real and imaginary parts, of the same class
    (NaN / +Inf / -Inf), isolated samples, clusters, the first and last positions of a tile and of the stream.  301 and 1001
    taps run on the real-stream transform tiles (k_fftfilt_real<.., HILB>, no hooks: it sits at its register limit) with a
    refold pass behind them (kernels_misc.hip k_hilbert_refold_nonfinite) — found missing by test_fuzz_nonfinite_sets."""
    n = 100_000
    for L in (31, 65, 129, 301, 1001):
        for seed, extra in ((9, []), (10, [0, 1, 4095, 4096, 4097, 8191, 8192, n - 1, n - 2])):
            x = _poisoned(rnd_f(n, 4), seed)
            for k, p in enumerate(extra):                 # tile edges of the kernel (4096 outputs per tile) and stream ends
                x[p] = [np.nan, np.inf, -np.inf][k % 3]
            yo = run_chain([orc.Hilbert(L)], x)
            yg = run_chain([rr.Hilbert(L)], x)
            assert len(yo) == len(yg) == n
            for part in (np.real, np.imag):
                a, b = part(yo), part(yg)
                bad_o, bad_g = ~np.isfinite(a), ~np.isfinite(b)
                assert np.array_equal(bad_o, bad_g), (L, seed, part.__name__, int(bad_o.sum()), int(bad_g.sum()),
                                                      np.flatnonzero(bad_o != bad_g)[:8])
                assert np.array_equal(np.isnan(a[bad_o]), np.isnan(b[bad_o])) and np.array_equal(a[bad_o & ~np.isnan(a)], b[bad_o & ~np.isnan(a)])
            ok = ~_nonfinite_mask(yo)
            assert max_norm_err(yg[ok], yo[ok]) <= TOL


@pytest.mark.parametrize("kind", ["chain-1:6", "chain-1:10-two-wave-kernel", "multi-12", "multi-8", "chain-full-rate-2:3"])
def test_nonfinite_samples_in_the_fused_fm_chains(rr, kind):
    """Round 5: the decimate-first chains demodulate a tile without atan2's inf / NaN fix-ups when tile_tame() finds every value
    of the tile finite (csrc/kernels_poly.hip), and with them otherwise.  A poisoned input (NaN / +-Inf, whole samples and
    single components) must leave every output the reference keeps finite within tolerance unless it shares a GPU tile with a
    poisoned one (FftFilter's known tile-sized smear, DESIGN "known deviations"), and every output the reference poisons
    non-finite."""
    from tests.harness import angle_parity
    taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
    n = 400_000
    # an FM-ish carrier (|r| stays away from 0: the plain bound applies almost everywhere)
    ph = np.cumsum(0.3 * np.sin(2 * np.pi * 1e-3 * np.arange(n)))
    clean = (np.exp(1j * ph) + 0.02 * rnd_c(n, 3)).astype(np.complex64)
    x = _poisoned(clean, 13)
    I, D = (2, 3) if kind.endswith("2:3") else (1, 10) if "1:10" in kind else (1, 6)
    if kind.startswith("multi"):
        t3 = np.stack([taps, np.conj(taps), (taps * np.exp(1j * 0.1 * np.arange(len(taps)))).astype(np.complex64)])
        with rr.build_options(fm_poly=12 if kind == "multi-12" else 8):
            blk = rr.FmMulti(t3, I, D, 1.0)
        st, c_, p_, need, yg = blk.work(x, 1_024_000)                 # one window holds the whole stream (N outputs: (N, produced))
        assert yg.shape == (3, p_) and p_ > 60_000
        chans = [(t3[c], yg[c]) for c in range(3)]
    else:
        with rr.build_options(**({"fm_poly": 1} if "1:10" in kind else {})):      # (1:10 at 463 taps: decimate-first tiles forced)
            blk = rr.FmChain(taps, I, D, 1.0)
        chans = [(taps, run_chain([blk], x))]
    for tc, got in chans:
        want = run_chain([orc.FftFilter(tc), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x)
        ro = run_chain([orc.FftFilter(tc), orc.RationalResampler(I, D)], x)
        assert len(want) >= len(got) > 1000 and (len(got) == len(want) or kind.startswith("multi"))
        want, ro = want[:len(got)], ro[:len(got) + 1]                  # (one work() call: the block's last partial tile is not out yet)
        bo, bg = ~np.isfinite(want), ~np.isfinite(got)
        # every output whose filter window holds a poisoned sample is poisoned (the data dependence itself: y[n] sees
        # x[n - L + 1 .. n], r[m] = y[floor(m D / I)], the angle m sees r[m] and r[m + 1]); the reference's FFT blocks and the
        # GPU's tiles both smear further, each over its own block
        xm = _nonfinite_mask(x).astype(np.float64)
        dep_y = np.convolve(xm, np.ones(len(tc)))[:len(x)] > 0
        idx = (np.arange(len(want) + 1) * D) // I
        dep_r = dep_y[np.minimum(idx, len(dep_y) - 1)]
        must = dep_r[:-1] | dep_r[1:]
        assert bo.sum() > 20 and must.sum() > 20 and not np.any(must & ~bg) and not np.any(must & ~bo)
        # outputs farther than a GPU tile (1024 outputs on the decimate-first tiles, <= 8192 input samples on the others) from
        # every poisoned output of the reference: untouched
        near = np.convolve(bo.astype(np.float64), np.ones(2 * (max(1024, 8192 * I // D) + 2) + 1), mode="same") > 0
        assert not np.any(bg & ~near), int(np.sum(bg & ~near))
        ok = ~near
        ro_ok = np.where(np.isfinite(ro.real) & np.isfinite(ro.imag), ro, 1.0)
        par = angle_parity(np.where(ok, got, 0.0), np.where(ok, np.nan_to_num(want), 0.0), ro_ok)
        assert par["used"] <= 1.0, (kind, par)


def test_a_window_of_nans_through_a_long_filter_is_not_a_throughput_collapse(rr):
    """the non-finite repair (csrc/nan_fix.hpp) recomputes every non-finite output with the reference's own fold, one
    thread per output; a fold that has become NaN stops (it stays NaN), so a window of NaNs through a 16383-tap FirFilter costs
    about what a clean window does — not L dependent loads per output."""
    import time
    taps = (rnd_c(16383, 3) / 4000).astype(np.complex64)
    n = 400_000
    clean = rnd_c(n, 4)
    bad = np.full(n, np.nan + 0j, np.complex64)
    blk = rr.FirFilter(taps)
    blk.work(clean, n)                                       # (first call: tables, allocations)
    t0 = time.perf_counter(); st, c, p, need, y0 = blk.work(clean, n); t_clean = time.perf_counter() - t0
    t_bad = float("inf")
    for _ in range(3):          # (the best of three: one call in a few hundred stalls for tens of ms on this pool whatever it runs —
        #  seen once in 27 suite runs, 81 ms against the usual 0.45; a collapse would be every call)
        t0 = time.perf_counter(); st, c, p, need, y1 = blk.work(bad, n); t_bad = min(t_bad, time.perf_counter() - t0)
    assert p == len(y1) > 300_000 and np.all(np.isnan(y1.real)) and np.all(np.isnan(y1.imag))
    assert t_bad < 20 * t_clean + 0.05, (t_clean, t_bad)


@pytest.mark.parametrize("L,deci", [(16380, 2), (15400, 3), (16383, 8)])
def test_decimating_fir_near_the_top_of_the_tile_range(rr, L, deci):
    """a decimating FirFilter of 15293..16383 taps used to stay on 16384-point tiles, which keep 16385 - L samples each
    (16380 taps: 5 — seconds per 1e7 samples).  It now runs the any-size overlap-save frames at the full rate and keeps every
    deci-th output with a strided copy: parity with the oracle, the reference's protocol, and a throughput bound."""
    import time
    import torch
    taps = (rnd_c(L, L) / 4000).astype(np.complex64)
    x = rnd_c(60_000, L + deci)
    yo = run_chain([orc.FirFilter(taps, deci)], x)
    logs = []
    yg = run_chain([rr.FirFilter(taps, deci=deci)], x, log=logs)
    logo = []
    run_chain([orc.FirFilter(taps, deci)], x, log=logo)
    assert logs == logo and len(yg) == len(yo) > 0
    assert max_norm_err(yg, yo) <= TOL
    n = 10_000_000
    dx = torch.rand(2 * n, device="cuda") * 2 - 1
    dy = torch.empty(2 * (n // deci + 8), device="cuda")
    blk = rr.FirFilter(taps, deci=deci)
    s = torch.cuda.current_stream().cuda_stream
    blk.work_dev(dx.data_ptr(), n, dy.data_ptr(), n // deci + 8, s); torch.cuda.synchronize()
    dt = float("inf")
    for _ in range(3):          # (best of three, as above)
        t0 = time.perf_counter()
        st, c, p, need = blk.work_dev(dx.data_ptr(), n, dy.data_ptr(), n // deci + 8, s); torch.cuda.synchronize()
        dt = min(dt, time.perf_counter() - t0)
    assert p == (n - L + 1) // deci and c == p * deci
    assert dt < 0.05, dt          # (any-size frames: a few ms per 1e7 samples; the 16384-point tiles took 0.3 s at 16380 taps)
    # spot check at full size against f64
    k0 = 1_234_567
    seg = dx[2 * k0 * deci:2 * (k0 * deci + 63 * deci + L)].cpu().numpy().view(np.complex64).astype(np.complex128)
    ref = np.convolve(seg, taps.astype(np.complex128))[L - 1::deci][:64]
    got = dy[2 * k0:2 * (k0 + 64)].cpu().numpy().view(np.complex64)
    assert max_norm_err(got, ref) <= TOL


def test_fftfilter_pass_inside_the_tile_kernel(rr):
    """Round 6: the pass that keeps FftFilter's outputs on non-finite input the reference's
    (fft_filter.rs:326-347) can live in the tile kernel's tail (rr_build_opts.fft_nonfinite_tiles = 3): a work() on a device
    window is then ONE launch, clean or not, and the non-finite call still gives the reference's set.  It is opt-in: the
    per-workgroup release it needs costs more than the launch it saves (csrc/blocks.cpp ref_blocks_on has the numbers); the
    default stays two launches."""
    import torch
    taps = orc.low_pass_complex(10e6, 1e6, 60e3)
    s = torch.cuda.current_stream().cuda_stream
    launches = rr.lib().rr_debug_kernel_launches
    for knob, want in ((3, 1), (0, 2)):
        for n in (512_000, 10_000_000):
            x = rnd_c(n, 77)
            dx = torch.from_numpy(x.view(np.float32)).cuda()
            dy = torch.empty(2 * (n + 1024), device="cuda")
            with rr.build_options(fft_nonfinite_tiles=knob):
                blk = rr.FftFilter(taps)
            blk.work_dev(dx.data_ptr(), n, dy.data_ptr(), n + 1024, s); torch.cuda.synchronize()
            for _ in range(3):
                c0 = launches()
                st, c, p, need = blk.work_dev(dx.data_ptr(), n, dy.data_ptr(), n + 1024, s)
                assert launches() - c0 == want and p > 0
            torch.cuda.synchronize()
    # a NaN in the window: still one launch, and exactly the reference's outputs are NaN; over several calls, with a NaN in
    # the last block of one call (its tail lies at the head of the next)
    n = 512_000
    S = 623
    x = rnd_c(3 * n, 78)
    x[123_456] = np.nan
    x[(n // S) * S - 5] = np.inf                            # the last block the first call processes
    x[2 * n + 77] = np.nan
    with rr.build_options(fft_nonfinite_tiles=3):
        blk = rr.FftFilter(taps)
    ob = orc.FftFilter(taps)
    ring_g = ring_o = np.zeros(0, np.complex64)
    yg_all, yo_all = [], []
    for k in range(3):
        ring_g = np.concatenate([ring_g, x[k * n:(k + 1) * n]])
        dx = torch.from_numpy(ring_g.view(np.float32).copy()).cuda()
        dy = torch.empty(2 * (len(ring_g) + 1024), device="cuda")
        c0 = launches()
        st, c, p, need = blk.work_dev(dx.data_ptr(), len(ring_g), dy.data_ptr(), len(ring_g) + 1024, s); torch.cuda.synchronize()
        assert launches() - c0 == 1
        st2, c2, p2, need2, yo = ob.work(ring_g, len(ring_g) + 1024)
        assert (st, c, p, need) == (st2, c2, p2, need2)
        yg_all.append(dy[:2 * p].cpu().numpy().view(np.complex64)); yo_all.append(yo)
        ring_g = ring_g[c:]
    yg, yo = np.concatenate(yg_all), np.concatenate(yo_all)
    bad_o = ~(np.isfinite(yo.real) & np.isfinite(yo.imag))
    bad_g = ~(np.isfinite(yg.real) & np.isfinite(yg.imag))
    assert 3 * 1023 <= bad_o.sum() <= 3 * 1023 + 1023 and np.array_equal(bad_o, bad_g)
    ok = ~bad_o
    assert max_norm_err(yg[ok], yo[ok]) <= TOL


def _nan_poisoned(x, seed, extra=()):
    """NaN only (whole samples and single components), isolated, in a cluster, at both ends of the stream and wherever `extra` says"""
    rng = np.random.default_rng(seed)
    x = x.copy()
    n = len(x)
    pos = sorted(set([0, 3, n // 7, n // 7 + 1, n // 3, n // 2 + 5, n - 9, n - 1] + [int(p) for p in rng.integers(0, n, 5)] + list(extra)))
    for k, p in enumerate(pos):
        if np.iscomplexobj(x):
            x[p] = [complex(np.nan, 0.25), complex(-0.5, np.nan), complex(np.nan, np.nan)][k % 3]
        else:
            x[p] = np.nan
    return x


CHAIN_NAN_KINDS = ["chain-1:6-decimate-first", "chain-1:6-small-windows", "chain-1:10-two-wave-kernel", "chain-2:3-full-rate",
                   "chain-25:128-2467-taps", "firfm-127x401-1:4", "multi-12", "multi-8", "multi-4096-point-tiles", "multi-small-windows",
                   "audio-6:25", "audio-1:5-small-windows"]


@pytest.mark.parametrize("kind", CHAIN_NAN_KINDS)
def test_nan_sets_of_the_fused_chains_are_the_references(rr, kind):
    """Round 6: the fused FM / audio chains put the REFERENCE's NaN set in place — FftFilter poisons the
    blocks [b S, (b + 1) S + ntaps) of its output (fft_filter.rs:326-347), the resampler picks (rational_resampler.rs:183-198),
    the demodulator pairs (quadrature_demod.rs:65-109): an output is NaN iff a filtered sample it reads lies in such a stretch;
    every other output is finite and within tolerance, whatever GPU tile it shared with a NaN (csrc/kernels_misc.hip
    k_chain_blocks_nonfinite behind the chain kernel).  Also across work() calls: the small windows end most calls inside a
    poisoned stretch, so the carried verdicts (last block, second-to-last, the carried r) are used.  NaN only: what an Inf
    turns into inside rustfft is not defined by anything this repository holds."""
    from harness import angle_parity
    small = "small" in kind
    if kind.startswith("audio"):
        I, D = (6, 25) if "6:25" in kind else (1, 5)
        taps = orc.low_pass(48e3, 6e3, 1.2e3)                     # real taps, FftFilterFloat
        n = 300_000
        x = _nan_poisoned(rnd_f(n, 61), 62)
        sb = 4 * 9_000 if small else 4_096_000
        yo = run_chain([orc.FftFilterFloat(taps), orc.RationalResampler(I, D, np.float32), orc.MultiplyConst(0.35)], x, stream_bytes=sb)
        yg = run_chain([rr.AudioChain(taps, I, D, 0.35)], x, stream_bytes=sb)
        assert len(yg) == len(yo) > 10_000
        bo, bg = ~np.isfinite(yo), ~np.isfinite(yg)
        assert 20 < bo.sum() < len(yo) and np.array_equal(bo, bg), (int(bo.sum()), int(bg.sum()), np.flatnonzero(bo != bg)[:8])
        assert max_norm_err(yg[~bo], yo[~bo]) <= TOL
        return
    n = 400_000
    ph = np.cumsum(0.3 * np.sin(2 * np.pi * 1e-3 * np.arange(n)))
    clean = (np.exp(1j * ph) + 0.02 * rnd_c(n, 3)).astype(np.complex64)      # |r| stays away from 0
    taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
    S = 561
    x = _nan_poisoned(clean, 13, extra=[(n // 2 // S) * S - 3, (n // 2 // S) * S + 2])        # (either side of a block boundary)
    I, D, opts, front = 1, 6, {}, None
    if "1:10" in kind:
        I, D, opts = 1, 10, {"fm_poly": 1}
    elif "2:3" in kind:
        I, D = 2, 3
    elif "25:128" in kind:
        I, D = 25, 128
        taps = orc.low_pass_complex(1.024e6, 100e3, 1e3)
        assert len(taps) == 2467
    elif kind.startswith("firfm"):
        I, D = 1, 4
        front, taps = orc.low_pass_complex(10e6, 1e6, 190e3), orc.low_pass_complex(10e6, 1e6, 60e3)
    sb = 8 * (30_000 if len(taps) > 2000 else 7_000) if small else 4_096_000
    if kind.startswith("multi"):
        t3 = np.stack([taps, np.conj(taps), (taps * np.exp(1j * 0.1 * np.arange(len(taps)))).astype(np.complex64)])
        opts = {"fm_poly": 12} if kind == "multi-12" else {"fm_poly": 8} if kind == "multi-8" else {"fm_poly": -1} if "4096" in kind else {}
        with rr.build_options(**opts):
            blk = rr.FmMulti(t3, I, D, 1.0)
        if small:
            from harness import drive_pageable
            yg3, _ = drive_pageable(blk, x, 7_000, 7_000)
        else:
            st, c_, p_, need, yg3 = blk.work(x, 1_024_000)
        chans = [(t3[c], yg3[c]) for c in range(3)]
        assert yg3.shape[1] > 60_000
    else:
        with rr.build_options(**opts):
            blk = rr.FmChain(taps, I, D, 1.0) if front is None else rr.FirFmChain(front, taps, I, D, 1.0)
        chans = [(taps, run_chain([blk], x, stream_bytes=sb))]
    for tc, got in chans:
        pre = [] if front is None else [orc.FirFilter(front)]
        want = run_chain(pre + [orc.FftFilter(tc), orc.RationalResampler(I, D), orc.QuadratureDemod(1.0)], x, stream_bytes=sb)
        ro = run_chain(pre + [orc.FftFilter(tc), orc.RationalResampler(I, D)], x, stream_bytes=sb)
        assert len(want) >= len(got) > 1000 and (len(got) == len(want) or kind.startswith("multi"))
        want, ro = want[:len(got)], ro[:len(got) + 1]
        bo, bg = ~np.isfinite(want), ~np.isfinite(got)
        assert 20 < bo.sum() < len(want)
        assert np.array_equal(bo, bg), (kind, int(bo.sum()), int(bg.sum()), np.flatnonzero(bo != bg)[:8])
        ro_ok = np.where(np.isfinite(ro.real) & np.isfinite(ro.imag), ro, 1.0)
        par = angle_parity(np.where(bo, 0.0, got), np.where(bo, 0.0, want), ro_ok)
        assert par["used"] <= 1.0, (kind, par)
