"""GPU: the fused chains make progress on ANY output window the three reference blocks would.

The reference chain FftFilter -> RationalResampler -> QuadratureDemod never needs more than ONE free slot in its final
output stream: FftFilter writes into its own inner 4 MB stream, the resampler emits what fits and carries a `pending` sample
across a full buffer (src/rational_resampler.rs:162-173,190-196), the demodulator takes min(len - 1, room)
(src/quadrature_demod.rs:49-57).  A fused kernel emits whole filter blocks; a window smaller than one block's outputs now
gets that block through a device-side tail (blocks.hpp OutTail) instead of WAIT_DST for ever."""
import numpy as np
import pytest

from harness import AGAIN, WAIT_DST, WAIT_SRC, angle_parity, max_norm_err, run_chain
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def rr():
    import rustradio_amd
    return rustradio_amd


def _fm(n, seed):
    rng = np.random.default_rng(seed)
    ph = np.cumsum(0.25 * np.sin(2 * np.pi * 2e-3 * np.arange(n)))
    return (np.exp(1j * ph) + 0.02 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)


def _drive(blk, x, in_cap, out_cap, max_calls=400_000):
    """Graph::run around ONE block with fixed ring sizes: push what fits, work(), drain the output ring completely.
    Returns (outputs [windows][n], log of (status, consumed, produced, need)); raises if the block stops making progress
    while input is left (the wait-for-ever this file is about)."""
    ring = np.zeros(0, blk.in_dtype)
    pos, outs, log = 0, [], []
    for _ in range(max_calls):
        take = min(in_cap - len(ring), len(x) - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, out_cap)
        log.append((st, c, p, need))
        ring = ring[c:]
        outs.append(np.atleast_2d(out))
        if take == 0 and c == 0 and p == 0:
            break
    else:
        raise AssertionError("no termination")
    return np.concatenate(outs, axis=1), log


SHAPES = [
    # ntaps-ish filter, interp, deci, input samples, input ring (samples), output window (elements)
    pytest.param((2.4e6, 100e3, 2.4e3), 5, 1, 40_000, 9_000, 18_000, id="soak-2467taps-5:1-72kB-rings"),
    pytest.param((2.4e6, 100e3, 100e3), 1, 1, 2_000, 512_000, 1, id="one-slot-output-window"),
    pytest.param((2.4e6, 100e3, 12.5e3), 1, 6, 30_000, 512_000, 7, id="463taps-1:6-window-7"),
    pytest.param((2.4e6, 100e3, 12.5e3), 3, 7, 20_000, 512_000, 100, id="463taps-3:7-window-100"),
    pytest.param((2.4e6, 100e3, 12.5e3), 1, 6, 60_000, 1_000, 93, id="463taps-1:6-window-just-below-a-block"),
]


@pytest.mark.parametrize("design,interp,deci,n,in_cap,out_cap", SHAPES)
def test_fm_chain_trickles_like_the_three_blocks(rr, design, interp, deci, n, in_cap, out_cap):
    taps = orc.low_pass_complex(*design)
    x = _fm(n, 11)
    blk = rr.FmChain(taps, interp, deci, 1.0)
    assert "unfused" not in blk.name                       # the fused kernel, not the composition
    yg, log = _drive(blk, x, max(in_cap, 2 * len(taps)), out_cap)
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(interp, deci), orc.QuadratureDemod(1.0)], x)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(interp, deci)], x)      # (fresh blocks: they carry state)
    assert yg.shape[1] == len(yo) > 0, (yg.shape, len(yo))
    assert angle_parity(yg[0], yo, ro)["used"] <= 1.0
    # protocol of the tail: while outputs are pending the block consumes nothing and asks for one free slot
    sts = {st for st, *_ in log}
    assert WAIT_DST in sts
    for (st, c, p, need), (st2, c2, p2, need2) in zip(log, log[1:]):
        if st == WAIT_DST and need == 1 and p == out_cap and p2 == out_cap and st2 == WAIT_DST:
            assert c2 == 0 or need2 == 1
    assert all(p <= out_cap for _, _, p, _ in log)
    assert blk.eof(True)


def test_fm_chain_u8_and_fir_front_trickle(rr):
    taps = orc.low_pass_complex(2.4e6, 100e3, 12.5e3)
    rng = np.random.default_rng(3)
    xb = rng.integers(0, 256, 2 * 25_000 + 1, dtype=np.uint8)
    blk = rr.FmChainU8(taps, 1, 6, 1.0)
    yg, log = _drive(blk, xb, 200_000, 5)
    yo = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)], xb)
    ro = run_chain([orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(1, 6)], xb)
    assert yg.shape[1] == len(yo) > 1000
    assert angle_parity(yg[0], yo, ro)["used"] <= 1.0
    # FirFilter in front (the metric's four-block chain)
    fir = orc.low_pass_complex(2.4e6, 400e3, 50e3)
    x = _fm(30_000, 5)
    blk = rr.FirFmChain(fir, taps, 1, 4, 1.0)
    yg, log = _drive(blk, x, 200_000, 9)
    yo = run_chain([orc.FirFilter(fir), orc.FftFilter(taps), orc.RationalResampler(1, 4), orc.QuadratureDemod(1.0)], x)
    ro = run_chain([orc.FirFilter(fir), orc.FftFilter(taps), orc.RationalResampler(1, 4)], x)
    assert yg.shape[1] == len(yo) > 1000
    assert angle_parity(yg[0], yo, ro)["used"] <= 1.0


@pytest.mark.parametrize("interp,deci,out_cap", [(1, 6, 3), (1, 6, 50), (2, 5, 64)])
def test_fm_multi_trickles(rr, interp, deci, out_cap):
    fs, nch, n = 2.4e6, 3, 20_000
    proto = orc.low_pass_complex(fs, 100e3, 12.5e3)
    k = np.arange(len(proto), dtype=np.float64)
    taps = np.stack([(proto.astype(np.complex128) * np.exp(2j * np.pi * ((c - 1) * 9e3) * k / fs)).astype(np.complex64)
                     for c in range(nch)])
    x = _fm(n, 21)
    blk = rr.FmMulti(taps, interp, deci, 1.0)
    yg, log = _drive(blk, x, 512_000, out_cap)
    for c in range(nch):
        yo = run_chain([orc.FftFilter(taps[c]), orc.RationalResampler(interp, deci), orc.QuadratureDemod(1.0)], x)
        ro = run_chain([orc.FftFilter(taps[c]), orc.RationalResampler(interp, deci)], x)
        assert yg.shape[1] == len(yo) > 1000
        assert angle_parity(yg[c], yo, ro)["used"] <= 1.0
    assert any(st == WAIT_DST and need == 1 for st, _, _, need in log)


@pytest.mark.parametrize("out_cap", [1, 37, 260])
def test_audio_chain_trickles(rr, out_cap):
    taps = orc.low_pass(200_000.0, 44_100.0, 500.0)           # 963 taps, 6 : 25 (examples/rtl_fm.rs:398-418)
    x = np.random.default_rng(9).uniform(-1, 1, 12_000).astype(np.float32)
    blk = rr.AudioChain(taps, 48000, 200000, 0.5)
    yg, log = _drive(blk, x, 1_024_000, out_cap)
    yo = run_chain([orc.FftFilterFloat(taps), orc.RationalResampler(48000, 200000, np.float32), orc.MultiplyConst(0.5)], x)
    assert yg.shape[1] == len(yo) > 1000
    assert max_norm_err(yg[0], yo) <= TOL


def test_trickle_between_device_rings(rr):
    """rr_block_work_streams: the fused chain into an HBM ring of 4096 B (1024 f32), one block = 1145 outputs."""
    taps = orc.low_pass_complex(2.4e6, 100e3, 2.4e3)              # 2467 taps: nsamples 5725, 1:5 -> 1145 outputs per block
    x = _fm(60_000, 2)
    blk = rr.FmChain(taps, 1, 5, 1.0)
    src, dst = rr.DeviceStream(np.complex64, 4_096_000), rr.DeviceStream(np.float32, 4096)
    pos, got = 0, []
    for _ in range(100_000):
        pos += src.push(x[pos:])
        st, c, p, need = blk.work_streams(src, dst)
        y = dst.pop()
        got.append(y)
        if c == 0 and p == 0 and len(y) == 0 and pos == len(x):
            break
    else:
        raise AssertionError("no termination")
    yg = np.concatenate(got)
    yo = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 5), orc.QuadratureDemod(1.0)], x)
    ro = run_chain([orc.FftFilter(taps), orc.RationalResampler(1, 5)], x)
    assert len(yg) == len(yo) > 5000
    assert angle_parity(yg, yo, ro)["used"] <= 1.0


def test_rotator_mode_switch_replay_model_replay(rr):
    """REPLAY -> MODEL (>= 1e5 outputs, the chain is not generated meanwhile) -> REPLAY must resume the
    reference's recurrence at the right phase without writing outside the phase ring (65536 phases at these windows)."""
    fs, f = 1.0e6, 123_456.7
    one = np.ones(1, np.complex64)                                # a 1-tap filter: the output IS the rotator
    n1, n2, n3 = 50_000, 400_000, 300_000
    x = np.ones(n1 + n2 + n3, np.complex64)
    yo = run_chain([orc.FirFilter(one, translate=(fs, f))], x)
    b = rr.FirFilter(one, translate=(fs, f))
    outs = []

    def feed(lo, hi, step=20_000):
        for a in range(lo, hi, step):
            st, c, p, need, out = b.work(x[a:min(a + step, hi)], step)
            assert c == p == min(step, hi - a)
            outs.append(out)

    feed(0, n1)
    b.set_rotator_mode(rr.ROT_MODEL)
    feed(n1, n1 + n2)
    b.set_rotator_mode(rr.ROT_REPLAY)
    feed(n1 + n2, n1 + n2 + n3)
    y = np.concatenate(outs)
    assert len(y) == len(yo)
    assert np.array_equal(y[:n1], yo[:n1])                        # replay: bit for bit
    assert np.array_equal(y[n1 + n2:], yo[n1 + n2:])              # ... and again after the model interlude
    assert max_norm_err(y[n1:n1 + n2], yo[n1:n1 + n2]) < 0.1      # the model drifts (test_rotator_drift_vs_length), no garbage
    # a second block created afterwards still works: nothing was corrupted on the device
    b2 = rr.FirFilter(one, translate=(fs, f))
    assert np.array_equal(b2.work(x[:1000], 1000)[4], yo[:1000])


def test_unfused_composition_reports_the_starved_blocks_need(rr):
    """behind rr_fm_chain_u8_create / rr_fir_fm_chain_create the UNFUSED composition (filters beyond the fused
    tiles) must report the same WAIT_SRC `need` as the fused block of the same chain — what the starved FftFilter wants,
    carried back through RtlSdrDecode (bytes) / the front FirFilter (its ntaps - 1 kept samples) — not the one byte pair /
    ntaps samples block 0 alone asks for, which would wake a scheduler for nothing."""
    rng = np.random.default_rng(5)
    L = 20_000                                                  # > 16383: composed (nsamples = 2 * 32768 - L = 45536)
    taps = ((rng.standard_normal(L) + 1j * rng.standard_normal(L)) / L).astype(np.complex64)
    b = rr.FmChainU8(taps, 1, 6, 1.0)
    assert "unfused" in b.name
    xb = rng.integers(0, 256, 1000, dtype=np.uint8)
    st, c, p, need, _ = b.work(xb, 100_000)
    assert (st, c, p, need) == (WAIT_SRC, 1000, 0, 2 * (45536 - 500))
    small = ((rng.standard_normal(463) + 1j * rng.standard_normal(463)) / 463).astype(np.complex64)
    f = rr.FmChainU8(small, 1, 6, 1.0)                          # the fused block of the same chain shape: same rule
    assert "unfused" not in f.name
    st, c, p, need, _ = f.work(xb, 100_000)
    assert (st, c, p, need) == (WAIT_SRC, 1000, 0, 2 * (561 - 500))
    fir = np.ones(5, np.complex64) / 5
    g = rr.FirFmChain(fir, taps, 1, 6, 1.0)
    assert "unfused" in g.name
    x = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000)).astype(np.complex64)
    st, c, p, need, _ = g.work(x, 100_000)
    # the FirFilter keeps ntaps - 1 = 4 samples in the window; FftFilter has 2996 of its 45536 pending
    assert (st, c, p) == (WAIT_SRC, 2996, 0) and need == 45536 - 2996 + 4
