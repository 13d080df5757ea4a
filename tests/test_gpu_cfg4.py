"""BASELINE configs[3] at its own workload: 32 FM channels per GPU (256 over 8 GPUs) on one shared IQ source,
with exactly the taps bench.py's `fm_multi` workload builds (rustradio_amd.multi.cfg4_taps: the configs[2]
low-pass shifted to f_c = (c - 128) * 8 kHz).  Every channel of the fused multi-channel block must equal its own
oracle chain FftFilter(taps_c) -> RationalResampler(1, 6) -> QuadratureDemod (examples/rtl_fm.rs:381-419 wiring);
the 8-rank sharding (multi.shard_channels(256, 8, r)) run rank by rank on one GPU must cover the 256-channel
oracle set exactly once."""
import numpy as np
import pytest

from harness import angle_parity, knob, run_chain
from oracle import pyoracle as orc
from rustradio_amd import multi

pytestmark = pytest.mark.gpu
TOL = 1e-5
FS = multi.CFG4_FS


@pytest.fixture(scope="module")
def rr():
    import rustradio_amd
    return rustradio_amd


def stations(n, seed, offsets_hz):
    """a 2.4 Msps band with FM stations (75 kHz deviation, 1 kHz tone) at the given offsets + sigma = 0.01 noise"""
    t = np.arange(n, dtype=np.float64)
    r = np.random.default_rng(seed)
    x = 0.01 * (r.standard_normal(n) + 1j * r.standard_normal(n))
    for i, f in enumerate(offsets_hz):
        phi = 2 * np.pi * np.cumsum(f + 75e3 * np.sin(2 * np.pi * (1e3 + 37.0 * i) * t / FS)) / FS
        x += np.exp(1j * phi) / len(offsets_hz)
    return x.astype(np.complex64)


def drive_multi(blk, x, nch, cap_in, cap_out):
    """the reference's window protocol by hand for the multi-output block -> [nch] output streams"""
    outs = [[] for _ in range(nch)]
    pos, ring = 0, np.zeros(0, x.dtype)
    while True:
        take = min(cap_in - len(ring), len(x) - pos)
        ring = np.concatenate([ring, x[pos:pos + take]]); pos += take
        st, c, p, need, out = blk.work(ring, cap_out)
        ring = ring[c:]
        out = out.reshape(nch, -1)
        if p:
            for ch in range(nch):
                outs[ch].append(out[ch])
        if take == 0 and c == 0 and p == 0:
            break
    return [np.concatenate(o) if o else np.zeros(0, np.float32) for o in outs]


def check_channels(yg_all, taps, x, stream_bytes, centred=()):
    """every channel against its own oracle chain; prints how much of the propagated allowance is used and the share of
    samples above the PLAIN 1e-5 pi; channels in `centred` carry a station at their centre frequency (|r| stays large
    after the filter's start-up transient): there the plain bound must hold for every sample"""
    worst, worst_plain = 0.0, 0.0
    skip = taps.shape[1] // 6 + 2                                 # start-up transient of the filter
    for ch in range(len(taps)):
        yo = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)], x, stream_bytes=stream_bytes)
        ro = run_chain([orc.FftFilter(taps[ch]), orc.RationalResampler(1, 6)], x, stream_bytes=stream_bytes)
        yg = yg_all[ch]
        assert len(yg) == len(yo) > 0, (ch, len(yg), len(yo))
        r = angle_parity(yg, yo, ro, TOL, skip)
        assert r["used"] <= 1.0, (ch, r)
        if ch in centred:
            assert r["above_plain"] == 0.0, (ch, r)
        worst, worst_plain = max(worst, r["used"]), max(worst_plain, r["above_plain"])
    print(f"chain parity over {len(taps)} channels: at most {worst:.3f} of the propagated allowance used; "
          f"at most {100 * worst_plain:.3f} % of a channel's samples above the plain 1e-5 pi")
    return worst


@pytest.mark.parametrize("kernel", ["auto", "w8", "w12", "half", "full"])
@pytest.mark.parametrize("stream_bytes", [4_096_000, 8 * 37_003])
def test_cfg4_32_channels_per_gpu(rr, monkeypatch, kernel, stream_bytes):
    """bench.py's fm_multi block at N = 1: channels 0..31 of the 256-channel bank, 600,000 samples, reference-sized
    and small windows, every channel against its own oracle chain; auto = the kernel the block picks by itself,
    half / full = the folded 1024-point and the full-size inverse kernels."""
    if kernel == "full":
        knob(rr, monkeypatch, fm_full=1, fm_poly=-1)
    elif kernel == "half":
        knob(rr, monkeypatch, fm_poly=-1)
    elif kernel in ("w8", "w12"):          # the decimate-first kernel with 8 / 12 waves per workgroup at every window size
        knob(rr, monkeypatch, fm_poly=int(kernel[1:]))
    proto = orc.low_pass_complex(FS, 100e3, 12.5e3)
    assert len(proto) == 463
    chans = list(multi.shard_channels(32, 1, 0))
    taps = multi.cfg4_taps(proto, chans)
    assert taps.shape == (32, 463)
    # stations inside the band of channels 0..31 (f_c = -1024 .. -776 kHz) and outside it
    x = stations(600_000, 41, [-1000e3, -900e3, -800e3, 0.0, 400e3])
    blk = rr.FmMulti(taps, 1, 6, 1.0)
    yg = drive_multi(blk, x, 32, stream_bytes // 8, stream_bytes // 4)
    check_channels(yg, taps, x, stream_bytes)


def test_cfg4_centred_station_meets_the_plain_bound(rr):
    """for a station centred in its channel (here: one station at -1000 kHz = the centre of channel 3, alone
    in the band) |r| stays large after the filter's start-up transient, nothing amplifies the stage error, and the PLAIN
    1e-5 pi bound must hold for every sample of that channel — not only the propagated one."""
    proto = orc.low_pass_complex(FS, 100e3, 12.5e3)
    taps = multi.cfg4_taps(proto, list(multi.shard_channels(32, 1, 0)))
    x = stations(400_000, 53, [-1000e3])
    yg = drive_multi(rr.FmMulti(taps, 1, 6, 1.0), x, 32, 512_000, 1_024_000)
    check_channels(yg, taps, x, 4_096_000, centred=(3,))


def test_cfg4_u8_32_channels(rr):
    """the same 32 channels fed by the RTL-SDR byte stream (rr.FmMultiU8)"""
    proto = orc.low_pass_complex(FS, 100e3, 12.5e3)
    taps = multi.cfg4_taps(proto, range(32))
    z = stations(300_000, 43, [-1000e3, -850e3, 100e3])
    b = np.empty(2 * len(z), np.uint8)
    b[0::2] = np.clip(np.round(z.real / 0.008 * 0.5 + 127), 0, 255).astype(np.uint8)
    b[1::2] = np.clip(np.round(z.imag / 0.008 * 0.5 + 127), 0, 255).astype(np.uint8)
    yg = drive_multi(rr.FmMultiU8(taps, 1, 6, 1.0), b, 32, 4_096_000, 4_096_000 // 4)
    x = run_chain([orc.RtlSdrDecode()], b)
    check_channels(yg, taps, x, 4_096_000)


def test_cfg4_256_channels_sharded_over_8_ranks(rr):
    """all 8 ranks of the 8-GPU job, one after the other on this GPU: rank r builds the block for
    multi.shard_channels(256, 8, r); the union of their outputs is the 256-channel oracle set, each channel once."""
    proto = orc.low_pass_complex(FS, 100e3, 12.5e3)
    x = stations(200_000, 47, [-1000e3, -600e3, -250e3, 0.0, 130e3, 520e3, 910e3])
    seen = []
    for r in range(8):
        chans = list(multi.shard_channels(256, 8, r))
        assert len(chans) == 32
        taps = multi.cfg4_taps(proto, chans)
        yg = drive_multi(rr.FmMulti(taps, 1, 6, 1.0), x, 32, 512_000, 1_024_000)
        check_channels(yg, taps, x, 4_096_000)
        seen += chans
    assert seen == list(range(256))


@pytest.mark.parametrize("waves", [0, 8, 12])
@pytest.mark.parametrize("nch,n", [(32, 2_400_000), (9, 3_100_000), (17, 1_460_000)])
def test_cfg4_windows_of_more_tiles_than_cus(rr, monkeypatch, nch, n, waves):
    """bench.py's fm_multi step itself — ONE window of 2,400,000 samples = 423 tiles on 256 CUs — and two more shapes: there
    the launch hands every workgroup a contiguous run of channel rounds (8 channels of a tile each; launch_multi_poly_d)
    that starts and ends INSIDE tiles, so a tile's channels are computed by up to two workgroups; 9 and 17 channels leave a
    partial last round.  Every channel against its own oracle chain, as for the ring-sized windows above.  waves = 8 / 12
    forces the kernel variant (rr_build_opts.fm_poly), 0 is the library's own choice by predicted cost."""
    if waves:
        knob(rr, monkeypatch, fm_poly=waves)
    proto = orc.low_pass_complex(FS, 100e3, 12.5e3)
    taps = multi.cfg4_taps(proto, list(range(nch)))
    x = stations(n, 59 + nch, [-1000e3, -960e3, -900e3, 0.0])
    yg = drive_multi(rr.FmMulti(taps, 1, 6, 1.0), x, nch, n, n // 6 + 1024)
    check_channels(yg, taps, x, 4_096_000)
