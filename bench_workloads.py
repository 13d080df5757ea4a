"""bench_workloads.py — the workloads bench.py times (BASELINE.json configs and their variants): synthetic inputs generated on
the GPU, the block chain of each workload (through the C ABI: rustradio_amd), its algorithmic bytes and flops per sample
 and the f64 stage list (`ref`) bench_verify.py evaluates to check what was timed.  No measurement code here.

    fftfilter    configs[1]: FftFilter 401 taps (low_pass_complex(10e6, 1e6, 60e3) => reference fft_size 1024, nsamples 623),
                 10 Msps synthetic Complex<f32>, 10 s = 100,000,000 samples per step                       [the N = 1 headline]
    fir          configs[0]: FirFilter<Complex> 127 real taps, 1,000,000 samples (deci 1, > 40 taps: overlap-save FFT tiles)
    fm_chain     configs[2]: FftFilter(463) -> RationalResampler(1:6) -> QuadratureDemod, 2.4 Msps x 10 s, fused (rr.FmChain)
    fm_multi     configs[3]: 32 such channels per GPU on one shared IQ source (256 channels on 8 GPUs) [the N > 1 headline]
    channelizer  configs[4]: Hilbert(65) -> FirFilter(255 taps, deci 8), 100 Msps x 1 s (f32 in), fused into one composite
                 decimating FIR (rr.HilbertFir); channelizer_unfused = the two blocks; channelizer_translate = what one rank
                 of the 8-GPU variant runs (.translate(), replay rotator); channelizer_model = the opt-in f64 rotator
    fir_1e8 / fir_direct / fir_float   the configs[0] filter at 1e8 samples: FFT tiles / forced direct form / FirFilter<Float>
    fir_fft_chain[_unfused]   configs[0] taps -> configs[1] filter (the north star's ">= 100x CPU" pair)
    full_chain[_fused]        the metric's words: FirFilter(127) -> FftFilter(401) -> RationalResampler(1:4) -> QuadratureDemod
    rtl_fm_chain / rtl_fm_example / fm_multi_u8   the chains fed by the RTL-SDR byte stream (u8 I/Q in)
"""
from __future__ import annotations

import math

import numpy as np
import torch

import rustradio_amd as rr
from rustradio_amd import multi

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3       # same guide: peak vector FP32 (no MFMA on this path: vector contractions)
METRIC = "Msamples/s through FIR+FftFilter+Resampler+QuadDemod chain; % HBM roofline"


# ---- synthetic inputs (generated on the GPU; torch is plumbing only) -----------------------
def synth_complex(n, fs, tones_hz, seed, device, chunk=8_000_000):
    """uniform[-1,1) noise per component + unit tones, Complex<f32> interleaved -> float32[2n]."""
    out = torch.empty(2 * n, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        v = torch.rand(m, 2, generator=g, device=device, dtype=torch.float32) * 2 - 1
        t = torch.arange(s, s + m, device=device, dtype=torch.float64)
        for f in tones_hz:
            ph = (2 * math.pi * f / fs) * t
            v[:, 0] += torch.cos(ph).float() * 0.25
            v[:, 1] += torch.sin(ph).float() * 0.25
        out[2 * s:2 * (s + m)] = v.reshape(-1)
    return out


def synth_real(n, fs, tones_hz, seed, device, chunk=16_000_000):
    out = torch.empty(n, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        v = torch.rand(m, generator=g, device=device, dtype=torch.float32) * 2 - 1
        t = torch.arange(s, s + m, device=device, dtype=torch.float64)
        for f in tones_hz:
            v += torch.cos((2 * math.pi * f / fs) * t).float() * 0.25
        out[s:s + m] = v
    return out


def synth_fm(n, fs, device, seed, chunk=4_000_000):
    """Broadcast-FM-like station centred in the channel: 75 kHz deviation, 1 kHz tone, sigma=0.01 noise."""
    out = torch.empty(2 * n, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        t = torch.arange(s, s + m, device=device, dtype=torch.float64)
        # phase = integral of 2 pi * 75e3 * sin(2 pi 1e3 t): closed form
        ph = -(75e3 / 1e3) * torch.cos(2 * math.pi * 1e3 * t / fs)
        v = torch.stack([torch.cos(ph), torch.sin(ph)], dim=1).float()
        v += 0.01 * torch.randn(m, 2, generator=g, device=device, dtype=torch.float32)
        out[2 * s:2 * (s + m)] = v.reshape(-1)
    return out


def to_rtlsdr_bytes(f32):
    """what the dongle delivers for this signal: round((v / 0.008) + 127) clamped to a byte (rtlsdr_decode.rs:9-47 inverted)"""
    return torch.clamp(torch.round(f32 / 0.008 + 127.0), 0, 255).to(torch.uint8)


# ---- workloads ------------------------------------------------------------------------------
class Workload:
    """blocks = device-resident chain; bufs[i] feeds blocks[i]; bufs[-1] is the sink.  `make_blocks()` builds fresh handles
    of the same constructors (bench_verify runs ONE step on fresh handles: zero history, so the f64 evaluation of `ref`
    lines up with the output buffer)."""
    key = ""
    name = ""                      # <= 110 characters (the driver's record keeps 120 per string)
    desc = ""                      # the long form, for gpurun_out/bench_detail.json
    dtype = "f32"
    alg_bytes_per_sample = 0.0     # SURVEY §8d compulsory traffic per INPUT sample of the chain
    dominant = 0                   # index of the block whose kernel the roofline object describes
    dominant_bytes_per_unit = 0.0  # algorithmic bytes of that kernel per sample it consumes
    in_mult = 1                    # stream elements of the first block per input sample (2 for u8 I/Q bytes)
    units_per_sample = 1           # channel-samples per input sample (the N-channel block)
    dominant_flops_per_unit = 0.0  # NOMINAL flops per sample it consumes: the REFERENCE's algorithm (5 N log2 N per N-point transform)
    dominant_flops_exec_per_unit = None   # flops the GPU kernel actually EXECUTES per sample (None: the nominal count)
    kernel = ""                    # name of the dominant kernel (rocprofv3 --kernel-trace shows it)
    bound = None                   # forced label ("sequential_rotator"); None: the larger of the HBM and FP32 fractions
    bound_note = None              # when neither roofline is what the kernel is short of: what is, and the evidence
    rotator = None
    ref = None                     # f64 stage list of the whole chain (bench_verify.py); per channel: ref_of(c)
    n_windows = 1                  # output windows of the last block (channels)
    last_p = 0                     # elements the last block produced in the most recent step (per window)

    def build(self):
        self.blocks = self.make_blocks()
        return self

    def step(self, stream, src_ptr=None):
        """one pass over the resident batch (or the broadcast tile at src_ptr); returns units: input samples consumed by the
        first block x units_per_sample"""
        n_in = self.n * self.in_mult
        for i, b in enumerate(self.blocks):
            cap = self.caps[i]
            inp = src_ptr if (i == 0 and src_ptr is not None) else self.bufs[i].data_ptr()
            st, c, p, need = b.work_dev(inp, n_in, self.bufs[i + 1].data_ptr(), cap, stream)
            if i == 0:
                c //= self.in_mult
                consumed0 = c
            if i == self.dominant:
                self.dom_units += c
            n_in = p
        self.last_p = p
        return consumed0 * self.units_per_sample

    def exec_flops_per_unit(self):
        return self.dominant_flops_per_unit if self.dominant_flops_exec_per_unit is None else self.dominant_flops_exec_per_unit


chan_taps = multi.channel_taps
CHAIN_BOUND_NOTE = ("neither roofline binds: fed RTL-SDR bytes (1/3 of the input traffic) it is only 12 % faster, VALU 37 % busy, package "
                    "1182 W of 1400 at full clock; three waves per SIMD are bound by instruction ISSUE (profiles/TUNING_LOG.md 4.1d)")


def fft_flops(n):
    """nominal flop count of one n-point complex transform"""
    return 5.0 * n * math.log2(n)


def poly_exec_flops_per_sample(ntaps, deci, nch=1, demod=True):
    """flops the decimate-first tile kernels (kernels_poly.hip) EXECUTE per input sample: per tile of 1024 - ceil(L / D)
    outputs = D (1024 - Ls) inputs: D forward transforms of 1024 points (shared by all channels), and per channel D x 1024
    complex multiply-adds (8 flop), one inverse transform and the demodulation (conj-multiply 6 + polynomial atan2 27 + gain 1)"""
    ls = -(-ntaps // deci)
    sa = 1024 - ls
    per_ch = deci * 1024 * 8 + fft_flops(1024) + (sa * 34 if demod else 0)
    return (deci * fft_flops(1024) + nch * per_ch) / (deci * sa)


def _lp(fs, cutoff, tw, n_expected=None):
    t = rr.low_pass_complex(fs, cutoff, tw)
    assert n_expected is None or len(t) == n_expected, len(t)
    return t


def make_fftfilter(dev, rank, world, shared_src):
    w = Workload()
    w.name = "configs[1]: FftFilter 401 taps (ref fft_size 1024), 10 Msps Complex<f32>, 100,000,000 samples/step"
    fs, n = 10e6, 100_000_000
    taps = _lp(fs, 1e6, 60e3, 401)
    f_c = 0.0 if world == 1 else multi.channel_frequency(rank, world, 250e3)
    ct = chan_taps(taps, fs, f_c)
    w.make_blocks = lambda: [rr.FftFilter(ct)]
    w.build()
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev), 2 * n, torch.float32),
              torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
    w.caps = [n + 1024]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    gf = rr.fftfilter_dims(w.blocks[0])[2]
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / 623           # the reference: 1024-point blocks of 623 samples
    w.dominant_flops_exec_per_unit = (2 * fft_flops(gf) + 6 * gf) / (gf - 400)    # the GPU's tile
    w.kernel = "k_fftfilt_os"
    w.cpu = ("FftFilter", taps)
    w.ref = [("fft", ct)]
    return w


def _make_fir(dev, shared_src, n, label):
    w = Workload()
    fs = 10e6
    taps = _lp(fs, 1e6, 190e3, 127)
    w.name = f"{label}: FirFilter<Complex> 127 real taps, {n:,} samples/step (overlap-save tiles)"
    w.make_blocks = lambda: [rr.FirFilter(taps)]
    w.build()
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0001, dev), 2 * n, torch.float32),
              torch.empty(2 * n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    w.dominant_flops_per_unit = 8.0 * 127                                          # the reference: a 127-tap fold per sample
    w.dominant_flops_exec_per_unit = (2 * fft_flops(1024) + 6 * 1024) / (1024 - 126)
    w.kernel = "k_fftfilt_os"
    w.cpu = ("FirFilter", taps)
    w.ref = [("fir", taps, 1)]
    return w


def make_fir(dev, rank, world, shared_src):
    return _make_fir(dev, shared_src, 1_000_000, "configs[0]")


def make_fir_1e8(dev, rank, world, shared_src):
    """configs[0]'s filter at a steady-state size (1e6 samples is one launch of ~10 us: launch-bound)"""
    return _make_fir(dev, shared_src, 100_000_000, "configs[0] filter, steady-state size")


def make_fir_direct(dev, rank, world, shared_src):
    """configs[0]'s filter forced onto the DIRECT-FORM kernel (the north star's LDS-staged tap window + register-blocked
    dot products): vector-FP32-bound, 4 flop per real tap and sample (31.75 flop/B > the 19.7 flop/B ridge)"""
    with rr.build_options(fir_path="direct"):
        w = _make_fir(dev, shared_src, 100_000_000, "configs[0] filter, forced direct-form k_fir")
        mk = w.make_blocks

        def fresh():
            with rr.build_options(fir_path="direct"):
                return mk()
        w.make_blocks = fresh
    w.name = w.name.replace(" (overlap-save tiles)", "")
    w.dominant_flops_per_unit = w.dominant_flops_exec_per_unit = 4.0 * 127      # real taps: 2 FMA per tap and sample
    w.kernel = "k_fir"
    return w


def make_fir_float(dev, rank, world, shared_src):
    """Fir<Float> with the configs[0] taps on a real stream: two overlap-save segments per Complex tile"""
    w = Workload()
    w.name = "FirFilter<Float> 127 taps, 100,000,000 f32 samples/step (real-stream overlap-save tiles)"
    fs, n = 10e6, 100_000_000
    taps = rr.low_pass(fs, 1e6, 190e3)
    assert len(taps) == 127
    w.make_blocks = lambda: [rr.FirFilter(taps)]
    w.build()
    w.n = n
    w.bufs = [shared_src(lambda: synth_real(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0006, dev), n, torch.float32),
              torch.empty(n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 8.0
    w.dominant, w.dominant_bytes_per_unit = 0, 8.0
    w.dominant_flops_per_unit = 2.0 * 127
    w.dominant_flops_exec_per_unit = (2 * fft_flops(1024) + 6 * 1024) / (2 * (1024 - 126))
    w.kernel = "k_fftfilt_real"
    w.cpu = ("FirFilterFloat", taps)
    w.ref = [("fir", taps, 1)]
    return w


def make_fir_fft_chain(dev, rank, world, shared_src, fused=True):
    """the north star's ">= 100x the CPU reference" pair: 127-tap FirFilter -> FftFilter(401 taps, ref 1024-pt)
    on the configs[1] input.  fused: ONE convolution with the composite taps t1 (*) t2 (rr.FirFftFilter); unfused: two
    blocks with a device-resident intermediate."""
    w = Workload()
    fs, n = 10e6, 100_000_000
    t1 = _lp(fs, 1e6, 190e3, 127)
    t2 = _lp(fs, 1e6, 60e3, 401)
    src = shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev), 2 * n, torch.float32)
    w.n = n
    if fused:
        w.name = "FirFilter(127)->FftFilter(401) as ONE 527-tap convolution (rr.FirFftFilter), 10 Msps, 1e8 samples/step"
        w.make_blocks = lambda: [rr.FirFftFilter(t1, t2)]
        w.bufs = [src, torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
        w.caps = [n + 1024]
        w.dominant = 0
    else:
        w.name = "FirFilter(127)->FftFilter(401), two blocks, device-resident intermediate, 10 Msps, 1e8 samples/step"
        w.make_blocks = lambda: [rr.FirFilter(t1), rr.FftFilter(t2)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev), torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
        w.caps = [n, n + 1024]
        w.dominant = 1
    w.build()
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 16.0
    w.dominant_flops_per_unit = 8.0 * 127 + (2 * fft_flops(1024) + 6 * 1024) / 623
    w.dominant_flops_exec_per_unit = ((2 * fft_flops(2048) + 6 * 2048) / (2048 - 526) if fused else
                                      (2 * fft_flops(2048) + 6 * 2048) / (2048 - 400))
    w.kernel = "k_fftfilt_os"
    w.cpu = ("fir_fft_chain", (t1, t2))
    w.ref = [("fir", t1, 1), ("fft", t2)]
    return w


def make_fir_fft_chain_unfused(dev, rank, world, shared_src):
    return make_fir_fft_chain(dev, rank, world, shared_src, fused=False)


def make_full_chain(dev, rank, world, shared_src, fused=False):
    """BASELINE.json's metric in its own words: FIR + FftFilter + Resampler + QuadDemod as ONE chain —
    FirFilter(127 real taps) -> FftFilter(401 taps) -> RationalResampler(1:4) -> QuadratureDemod on the configs[1] input
    (10 Msps; 1 MHz low-pass => 2.5 Msps after 1:4).  unfused: four blocks, device-resident intermediates; fused: the
    composite 527-tap filter, the resampler and the demodulator in one kernel (rr.FirFmChain)."""
    w = Workload()
    fs, n = 10e6, 100_000_000
    t1 = _lp(fs, 1e6, 190e3, 127)
    t2 = _lp(fs, 1e6, 60e3, 401)
    src = shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev), 2 * n, torch.float32)
    w.n = n
    w.alg_bytes_per_sample = 8.0 + 4.0 / 4.0
    oc = n // 4 + 1024
    nominal = 8.0 * 127 + (2 * fft_flops(1024) + 6 * 1024) / 623 + 40.0 / 4.0
    if fused:
        w.name = "FirFilter(127)->FftFilter(401)->RationalResampler(1:4)->QuadratureDemod, ONE kernel (rr.FirFmChain), 1e8 samples/step"
        w.make_blocks = lambda: [rr.FirFmChain(t1, t2, 1, 4, 1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(oc, dtype=torch.float32, device=dev)]
        w.caps = [oc]
        w.dominant, w.dominant_bytes_per_unit = 0, 9.0
        w.kernel = "k_fm_chain*"
        w.bound_note = CHAIN_BOUND_NOTE
        w.dominant_flops_exec_per_unit = poly_exec_flops_per_sample(527, 4)
    else:
        w.name = "FirFilter(127)->FftFilter(401)->RationalResampler(1:4)->QuadratureDemod, four blocks, 1e8 samples/step"
        w.make_blocks = lambda: [rr.FirFilter(t1), rr.FftFilter(t2), rr.RationalResampler(1, 4, np.complex64),
                                 rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev), torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev),
                  torch.empty(2 * oc, dtype=torch.float32, device=dev), torch.empty(oc, dtype=torch.float32, device=dev)]
        w.caps = [n, n + 1024, oc, oc]
        w.dominant, w.dominant_bytes_per_unit = 1, 16.0
        w.kernel = "k_fftfilt_os"
        w.dominant_flops_exec_per_unit = (2 * fft_flops(2048) + 6 * 2048) / (2048 - 400)
    w.build()
    w.dominant_flops_per_unit = nominal
    w.cpu = ("full_chain", (t1, t2))
    w.ref = [("fir", t1, 1), ("fft", t2), ("rs", 1, 4), ("demod", 1.0)]
    return w


def make_full_chain_fused(dev, rank, world, shared_src):
    return make_full_chain(dev, rank, world, shared_src, fused=True)


def make_fm_chain(dev, rank, world, shared_src, fused=True):
    w = Workload()
    w.name = ("configs[2]: FftFilter(463)->RationalResampler(1:6)->QuadratureDemod, 2.4 Msps x 10 s = 24,000,000 samples/step, "
              + ("fused" if fused else "3 blocks"))
    fs, n = 2.4e6, 24_000_000
    taps = _lp(fs, 100e3, 12.5e3, 463)
    src = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0003), 2 * n, torch.float32)
    if fused:
        w.make_blocks = lambda: [rr.FmChain(taps, 1, 6, 1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
        w.caps = [n // 6 + 1024]
        w.dominant_bytes_per_unit = 8.0 + 4.0 / 6.0
        w.kernel = "k_fm_chain*"
        w.bound_note = CHAIN_BOUND_NOTE
        w.dominant_flops_exec_per_unit = poly_exec_flops_per_sample(463, 6)
    else:
        w.make_blocks = lambda: [rr.FftFilter(taps), rr.RationalResampler(1, 6, np.complex64), rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)]
        w.bufs = [src,
                  torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev),
                  torch.empty(2 * (n // 6 + 1024), dtype=torch.float32, device=dev),
                  torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
        w.caps = [n + 1024, n // 6 + 1024, n // 6 + 1024]
        w.dominant_bytes_per_unit = 16.0
        w.kernel = "k_fftfilt_os"
        w.dominant_flops_exec_per_unit = (2 * fft_flops(2048) + 6 * 2048) / (2048 - 462)
    w.build()
    w.n = n
    w.alg_bytes_per_sample = 8.0 + 4.0 / 6.0
    w.dominant = 0
    # nominal work of the chain as the reference runs it per input sample: two 1024-point transforms + the product per
    # 561 samples (fft_filter.rs:172-176) + conj-multiply and atan2 per output
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / 561 + 40.0 / 6.0
    w.cpu = ("fm_chain", taps)
    w.ref = [("fft", taps), ("rs", 1, 6), ("demod", 1.0)]
    return w


def make_rtl_fm_chain(dev, rank, world, shared_src):
    """configs[2] from the RTL-SDR wire format (examples/rtl_fm.rs:328-419): u8 I/Q pairs in, f32 out."""
    w = Workload()
    w.name = "RtlSdrDecode->FftFilter(463)->RationalResampler(1:6)->QuadratureDemod fused (rr.FmChainU8), u8 I/Q, 24e6 samples/step"
    fs, n = 2.4e6, 24_000_000
    taps = _lp(fs, 100e3, 12.5e3, 463)
    src = to_rtlsdr_bytes(synth_fm(n, fs, dev, 0x5EED0003))
    w.make_blocks = lambda: [rr.FmChainU8(taps, 1, 6, 1.0, rr.ATAN2_EXACT)]
    w.build()
    w.bufs = [src, torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
    w.caps = [n // 6 + 1024]
    w.in_mult = 2
    w.dtype = "u8->f32"
    w.n = n
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 2.0 + 4.0 / 6.0
    w.dominant = 0
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / 561 + 40.0 / 6.0
    w.dominant_flops_exec_per_unit = poly_exec_flops_per_sample(463, 6)
    w.kernel = "k_fm_chain*"
    w.bound_note = CHAIN_BOUND_NOTE
    w.cpu = ("rtl_fm_chain", taps)
    w.ref = [("u8",), ("fft", taps), ("rs", 1, 6), ("demod", 1.0)]
    return w


def make_rtl_fm_example(dev, rank, world, shared_src):
    """examples/rtl_fm.rs:328-419 with its own numbers: 1.024 Msps RTL-SDR bytes, low_pass_complex(fs, 100 kHz, 1 kHz)
    = 2467 taps (reference fft_size 8192), resampled 1,024,000 -> 200,000 (25:128), quadrature demod."""
    w = Workload()
    fs, n = 1.024e6, 24_000_000
    taps = _lp(fs, 100e3, 1e3)
    L = len(taps)
    w.name = f"examples/rtl_fm.rs front end: u8 I/Q->FftFilter({L})->RationalResampler(25:128)->QuadratureDemod fused, 24e6 samples/step"
    src = to_rtlsdr_bytes(synth_fm(n, fs, dev, 0x5EED0006))
    w.make_blocks = lambda: [rr.FmChainU8(taps, 200000, 1024000, 1.0, rr.ATAN2_EXACT)]
    w.build()
    cap = n * 25 // 128 + 4096
    w.bufs = [src, torch.empty(cap, dtype=torch.float32, device=dev)]
    w.caps = [cap]
    w.in_mult = 2
    w.dtype = "u8->f32"
    w.n = n
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 2.0 + 4.0 * 25 / 128
    w.dominant = 0
    w.dominant_flops_per_unit = (2 * fft_flops(8192) + 6 * 8192) / (8192 - L + 1) + 40.0 * 25 / 128
    # k_fm_chain_split<2>: an 8192-point tile as two 4096-point halves split in frequency — forward, product, inverse per
    # 8192 - L + 1 inputs, the demodulation (34 flop) on 25/128 of them
    w.dominant_flops_exec_per_unit = (2 * fft_flops(8192) + 6 * 8192) / (8192 - L + 1) + 34.0 * 25 / 128
    w.kernel = "k_fm_chain_split"
    w.bound_note = ("2467 taps: 8192-point tiles keep 5726 of 8192 points; the kernel is on the FP32 / issue side, not the HBM side "
                    "(2.78 B per sample): executed_fp32_frac is the figure to read")
    w.cpu = ("rtl_fm_example", taps)
    w.ref = [("u8",), ("fft", taps), ("rs", 25, 128), ("demod", 1.0)]
    return w


def make_fm_chain_unfused(dev, rank, world, shared_src):
    return make_fm_chain(dev, rank, world, shared_src, fused=False)


def make_fm_multi(dev, rank, world, shared_src, per_gpu=32, u8=False):
    """BASELINE configs[3]: 256 FM channels of configs[2] on one shared IQ source, 32 per GPU (weak scaling: N GPUs run
    the first 32 N channels of the bank; 8 GPUs = all 256).  Channel c uses the configs[2] low-pass shifted to
    f_c = (c - 128) * 8 kHz (complex band-pass, multi.cfg4_taps); rank r owns channels r*32 .. r*32+31
    (multi.shard_channels).  `value` counts channel-samples: input samples x channels processed."""
    w = Workload()
    fs, n = 2.4e6, 2_400_000
    taps = _lp(fs, 100e3, 12.5e3, 463)
    total = multi.CFG4_CHANNELS
    chans = list(multi.shard_channels(per_gpu * world, world, rank))
    w.name = (f"configs[3]: {len(chans)} FM channels/GPU (configs[2] chain, fused, rr.FmMulti{'U8' if u8 else ''}) on one shared 2.4 Msps "
              f"{'u8 I/Q' if u8 else 'IQ'} source, {n:,} samples/step/ch")
    w.desc = w.name + (f"; this rank: channels {chans[0]}..{chans[-1]} of the {total}-channel bank, {per_gpu * world} in the job" if world > 1 else "")
    if u8:      # the RTL-SDR wire format as the fan-out format: 2 B instead of 8 B per sample over xGMI, decoded in the kernel
        src = shared_src(lambda: to_rtlsdr_bytes(synth_fm(n, fs, dev, 0x5EED0004)), 2 * n, torch.uint8)
        w.in_mult, w.dtype = 2, "u8->f32"
    else:
        src = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0004), 2 * n, torch.float32)
    taps_all = multi.cfg4_taps(taps, chans, total)
    w.make_blocks = lambda: [(rr.FmMultiU8 if u8 else rr.FmMulti)(taps_all, 1, 6, 1.0, rr.ATAN2_EXACT)]   # one kernel: forward FFT shared by all channels
    w.build()
    w.n = n
    cap = n // 6 + 1024
    nch = len(chans)
    w.bufs = [src, torch.empty(nch * cap, dtype=torch.float32, device=dev)]
    w.caps = [cap]
    w.n_windows = nch
    w.units_per_sample = nch
    bin_ = 2.0 if u8 else 8.0
    w.alg_bytes_per_sample = bin_ / nch + 4.0 / 6.0       # shared read: 8/N (2/N) B in + 0.67 B out per channel-sample
    w.dominant, w.dominant_bytes_per_unit = 0, (bin_ / nch + 4.0 / 6.0) * nch
    # per channel and input sample the reference's filter work (two 1024-point transforms + product per 561 samples) + demod,
    # the forward transform shared by the channels of one GPU
    w.dominant_flops_per_unit = fft_flops(1024) / 561 + nch * ((fft_flops(1024) + 6 * 1024) / 561 + 40.0 / 6.0)
    # ... and what k_fm_multi_poly<6> executes: 6 shared phase transforms + per channel 6 x 1024 multiply-adds, ONE inverse
    # transform and the demodulation per 946 outputs = 5676 inputs
    w.dominant_flops_exec_per_unit = poly_exec_flops_per_sample(463, 6, nch)
    w.kernel = "k_fm_multi*"
    w.bound_note = ("the per-channel response stream comes from L2 (49 KB per channel-tile, 12 TB/s of the L2's 34.5): the waves wait "
                    "for it 41 % of the time (profiles/r05_fm_multi_stall_counters.txt); HBM sees the shared input once")
    w.cpu = ("fm_chain", taps)
    pre = [("u8",)] if u8 else []
    w.ref_of = lambda c: pre + [("fft", taps_all[c]), ("rs", 1, 6), ("demod", 1.0)]
    return w


def make_fm_multi_u8(dev, rank, world, shared_src):
    """configs[3] with the fan-out in the RTL-SDR wire format: the broadcast moves 2 B per sample"""
    return make_fm_multi(dev, rank, world, shared_src, u8=True)


def make_channelizer(dev, rank, world, shared_src, fused=True, rotator=None, as_rank=None):
    """BASELINE configs[4]; on N > 1 GPUs rank g runs channel offset f_g through FirFilter::translate(100e6, f_g)
    (src/fir.rs:476-486, SURVEY §8d cfg5: multi.cfg5_translate_hz) — with the library's DEFAULT rotator, the reference's own
    f32 recurrence replayed bit for bit (ROT_REPLAY: on parity for any stream length, one sequential chain per block,
    walked ahead of the filter — by one device lane at 10 ns per output while the block's calls leave it time, by a host
    thread at ~2.6 ns per output once they do not; back-to-back bench steps do not); `channelizer_model` is the same with
    the opt-in f64 closed form (parallel, but outside the 1e-5 parity bar beyond ~1e5 outputs of a stream)."""
    w = Workload()
    if as_rank is not None:                      # (N = 1 line: the workload ONE rank of the 8-GPU variant runs)
        rank, world = as_rank
    f_g = multi.cfg5_translate_hz(rank, world)
    rotator = rr.ROT_REPLAY if rotator is None else rotator
    w.name = ("configs[4]: Hilbert(65)->FirFilter(255 taps, deci 8), 100 Msps f32, 1e8 samples/step, "
              + ("fused (rr.HilbertFir)" if fused else "two blocks")
              + (f", .translate({f_g / 1e6:.3f} MHz) rotator={'replay' if rotator == rr.ROT_REPLAY else 'model (OFF parity)'}" if world > 1 else ""))
    w.rotator = None if world == 1 else ("replay" if rotator == rr.ROT_REPLAY else "model")
    fs, n = 100e6, 100_000_000
    taps = _lp(fs, 5e6, 943e3, 255)
    src = shared_src(lambda: synth_real(n, fs, (3e6, 12e6, 37e6), 0x5EED0005, dev), n, torch.float32)
    tr = (fs, f_g) if world > 1 else None
    w.n = n
    w.alg_bytes_per_sample = 5.0
    w.dominant_flops_per_unit = 2.0 * 65 + 8.0 * 255 / 8               # the reference: a 65-tap fold per sample + a 255-tap Complex fold per output
    if fused:
        w.make_blocks = lambda: [rr.HilbertFir(65, taps, 8, translate=tr, rotator=rotator)]
        w.bufs = [src, torch.empty(2 * (n // 8 + 8), dtype=torch.float32, device=dev)]
        w.caps = [n // 8 + 8]
        w.dominant, w.dominant_bytes_per_unit = 0, 5.0
        w.kernel = "k_fftfilt_prune"
        # two real segments per 2048-point Complex tile: one forward transform, the two products, two inverses pruned to 256
        w.dominant_flops_exec_per_unit = (fft_flops(2048) + 2 * 6 * 2048 + 2 * fft_flops(256)) / (2 * (2048 - 318))
    else:
        w.make_blocks = lambda: [rr.Hilbert(65), rr.FirFilter(taps, deci=8, translate=tr, rotator=rotator)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev),
                  torch.empty(2 * (n // 8 + 8), dtype=torch.float32, device=dev)]
        w.caps = [n, n // 8 + 8]
        w.dominant, w.dominant_bytes_per_unit = 0, 12.0
        w.kernel = "k_hilbert"
        w.dominant_flops_per_unit = 2.0 * 65                              # (the Hilbert block alone)
        w.dominant_flops_exec_per_unit = 2.0 * 33 + 2                     # k_hilbert skips the transformer's zero taps
    w.build()
    w.cpu = ("channelizer", taps)
    w.ref = [("hilbert", 65), ("fir", taps, 8) if tr is None else ("fir_translate", taps, 8, fs, f_g, rotator == rr.ROT_REPLAY)]
    return w


def make_channelizer_unfused(dev, rank, world, shared_src):
    return make_channelizer(dev, rank, world, shared_src, fused=False)


def make_channelizer_translate(dev, rank, world, shared_src):
    """what ONE rank of configs[4]'s 8-GPU variant runs (rank 1 of 8: .translate(100e6, f_1), default on-parity rotator), on
    the N = 1 run so that the driver sees it: bound by the sequential rotator chain, not by a roofline"""
    w = make_channelizer(dev, rank, world, shared_src, as_rank=(1, 8))
    w.bound = "sequential_rotator"
    w.bound_note = ("FirFilter::translate's rotator (src/fir.rs:464-473) is an un-renormalised f32 recurrence, one dependent chain of "
                    "12.5 M steps per step here; replayed bit for bit it runs at 2.6 ns per output on a host core (10 ns on a device "
                    "lane), whatever the filter kernel does (0.14 ms)")
    return w


def make_channelizer_model(dev, rank, world, shared_src):
    """configs[4]'s N > 1 variant with the OPT-IN model rotator (labelled off-parity; see make_channelizer)"""
    return make_channelizer(dev, rank, world, shared_src, rotator=rr.ROT_MODEL)


WORKLOADS = {"fftfilter": make_fftfilter, "fir": make_fir, "fm_chain": make_fm_chain,
             "fm_chain_unfused": make_fm_chain_unfused, "fm_multi": make_fm_multi, "fm_multi_u8": make_fm_multi_u8,
             "channelizer": make_channelizer, "channelizer_model": make_channelizer_model,
             "channelizer_translate": make_channelizer_translate,
             "rtl_fm_chain": make_rtl_fm_chain, "channelizer_unfused": make_channelizer_unfused,
             "fir_fft_chain": make_fir_fft_chain, "fir_fft_chain_unfused": make_fir_fft_chain_unfused,
             "full_chain": make_full_chain, "full_chain_fused": make_full_chain_fused,
             "rtl_fm_example": make_rtl_fm_example,
             "fir_1e8": make_fir_1e8, "fir_direct": make_fir_direct, "fir_float": make_fir_float}


def make(name, dev, rank, world, shared_src):
    w = WORKLOADS[name](dev, rank, world, shared_src)
    w.key = name
    if not w.desc:
        w.desc = w.name
    return w
