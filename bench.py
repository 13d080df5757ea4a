#!/usr/bin/env python3
"""bench.py — throughput of the rustradio hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  Default workload = BASELINE.json configs[1]:
    FftFilter, 401 taps (low_pass_complex(10e6, 1e6, 60e3) => reference fft_size 1024,
    nsamples 623), 10 Msps synthetic Complex<f32>, 10 s = 100,000,000 samples per step.
Other workloads (--workload, also summarised under "others" in the JSON line):
    fir          configs[0]: FirFilter<Complex> 127 real taps, 1,000,000 samples (deci 1 and > 40 taps: the block runs
                 on overlap-save FFT tiles, the FftFilter kernel with no history)
    fm_chain     configs[2]: FftFilter(463) -> RationalResampler(1:6) -> QuadratureDemod, 2.4 Msps x 10 s
    fm_multi     configs[3]: 32 such channels per GPU on one shared IQ source (256 channels on 8 GPUs)
    channelizer  configs[4]: Hilbert(65) -> FirFilter(255 taps, deci 8), 100 Msps x 1 s (f32 in), fused into one
                 composite decimating FIR (rr.HilbertFir: real-stream overlap-save tiles, inverse transform pruned to 1/8);
                 channelizer_unfused = the two blocks
    fir_1e8      the configs[0] filter on 100,000,000 samples (steady state; 1e6 samples is a single ~15 us launch)
    fir_float    FirFilter<Float>, the same 127 taps on 100,000,000 f32 samples (real-stream tiles, 8 B/sample)
    fir_fft_chain  configs[0] taps -> configs[1] filter as one chain (the north star's ">= 100x CPU" pair)
    rtl_fm_example examples/rtl_fm.rs with its own parameters (1.024 Msps, 2467 taps, 25:128), fused
    rtl_fm_chain configs[2] fed by the RTL-SDR byte stream: RtlSdrDecode fused in front (u8 in)

    full_chain   the metric's own words: FirFilter(127) -> FftFilter(401) -> RationalResampler(1:4) -> QuadratureDemod,
                 four blocks with device-resident intermediates; full_chain_fused = the best fused form
    dropin_*     (under "others" only) the DROP-IN path: rr_block_work on reference-sized 4,096,000-byte HOST windows
                 exactly as the Rust shim calls it, and the device-resident graph with reference-sized rings

Multi-GPU (`--gpus N`, one process per GPU, weak scaling): when WORLD_SIZE is not set, this process — before it
touches the GPU — starts N ranks of itself with torch.distributed.run and exits with their code.  The path shards
by channel: default N > 1 workload = configs[3] (fm_multi: rank r owns channels 32 r .. 32 r + 31 of the 256-channel
bank).  The only collective is the fan-out of the shared IQ source: rank 0 produces tile t+1 and RCCL-broadcasts it on a
communication stream while every rank runs tile t (double buffer, events) — INSIDE the timed region.

Prints ONE JSON line (rank 0).  `value` counts input samples entering the first block (x channels for the
multi-channel block), summed over ranks, per second of max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import statistics
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import rustradio_amd as rr  # noqa: E402
from rustradio_amd import multi  # noqa: E402

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


# ---- workloads ------------------------------------------------------------------------------
class Workload:
    """blocks = device-resident chain; bufs[i] feeds blocks[i]; bufs[-1] is the sink."""
    name = ""
    dtype = "f32"
    alg_bytes_per_sample = 0.0     # SURVEY §8d compulsory traffic per INPUT sample of the chain
    dominant = 0                   # index of the block whose kernel the roofline object describes
    dominant_bytes_per_unit = 0.0  # algorithmic bytes of that kernel per sample it consumes
    in_mult = 1                    # stream elements of the first block per input sample (2 for u8 I/Q bytes)
    bound = "hbm"                  # roofline that bounds the dominant kernel: "hbm" | "vector_fp32"
    dominant_flops_per_unit = 0.0  # NOMINAL flops per sample it consumes: the REFERENCE's algorithm (5 N log2 N per N-point transform)
    dominant_flops_exec_per_unit = None   # flops of the algorithm the GPU kernel actually EXECUTES per sample (None: the same count)
    kernel = ""                    # name of the dominant kernel (rocprofv3 --kernel-trace shows it)
    bound_note = None              # when neither roofline is what the kernel is short of: what is, and the evidence

    def step(self, stream, src_ptr=None):
        """one pass over the resident batch (or the broadcast tile at src_ptr); returns input samples consumed by the first block"""
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
        return consumed0


chan_taps = multi.channel_taps
CHAIN_BOUND_NOTE = ("neither roofline binds this kernel: fed RTL-SDR bytes (a third of the input traffic) it is only 12 % faster, its VALU is 37 % "
                    "busy and the package draws 1182 W of 1400 at full clock; three waves per SIMD are bound by instruction ISSUE (giving its "
                    "waiting waves redundant work made it 11 % slower, profiles/TUNING_LOG.md 4.1d round 4).  Both fractions are reported: "
                    "dominant_kernel_hbm_frac and dominant_kernel_executed_fp32_frac")


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


def make_fftfilter(dev, rank, world, shared_src):
    w = Workload()
    w.name = "configs[1]: FftFilter 401 taps (ref fft_size 1024, nsamples 623), 10 Msps Complex<f32>, 100,000,000 samples/step"
    fs, n = 10e6, 100_000_000
    taps = rr.low_pass_complex(fs, 1e6, 60e3)
    assert len(taps) == 401
    f_c = 0.0 if world == 1 else multi.channel_frequency(rank, world, 250e3)
    w.blocks = [rr.FftFilter(chan_taps(taps, fs, f_c))]
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev), 2 * n, torch.float32),
              torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
    w.caps = [n + 1024]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    gf = rr.fftfilter_dims(w.blocks[0])[2]
    w.dominant_flops_per_unit = (2 * fft_flops(gf) + 6 * gf) / (gf - 400)
    w.kernel = "k_fftfilt_os"
    w.cpu = ("FftFilter", taps)
    return w


def _make_fir(dev, shared_src, n, label):
    w = Workload()
    fs = 10e6
    taps = rr.low_pass_complex(fs, 1e6, 190e3)
    assert len(taps) == 127
    w.name = f"{label}: FirFilter<Complex> 127 real taps, {n:,} samples/step (deci 1, > 40 taps: overlap-save tiles)"
    w.blocks = [rr.FirFilter(taps)]
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0001, dev), 2 * n, torch.float32),
              torch.empty(2 * n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / (1024 - 126)
    w.kernel = "k_fftfilt_os"
    w.cpu = ("FirFilter", taps)
    return w


def make_fir(dev, rank, world, shared_src):
    return _make_fir(dev, shared_src, 1_000_000, "configs[0]")


def make_fir_1e8(dev, rank, world, shared_src):
    """configs[0]'s filter at a steady-state size (1e6 samples is one launch of ~15 us: launch-bound)"""
    return _make_fir(dev, shared_src, 100_000_000, "configs[0] filter at steady-state size")


def make_fir_direct(dev, rank, world, shared_src):
    """configs[0]'s filter forced onto the DIRECT-FORM kernel (the north star's LDS-staged tap window + register-blocked
    dot products): vector-FP32-bound, 4 flop per real tap and sample (31.75 flop/B > the 19.7 flop/B ridge)"""
    with rr.build_options(fir_path="direct"):
        w = _make_fir(dev, shared_src, 100_000_000, "configs[0] filter, direct form")
    w.name = w.name.replace("(deci 1, > 40 taps: overlap-save tiles)", "(forced direct-form k_fir)")
    w.bound, w.dominant_flops_per_unit, w.kernel = "vector_fp32", 4.0 * 127, "k_fir"
    return w


def make_fir_float(dev, rank, world, shared_src):
    """Fir<Float> with the configs[0] taps on a real stream: two overlap-save segments per Complex tile"""
    w = Workload()
    w.name = "FirFilter<Float> 127 taps, 100,000,000 f32 samples/step (real-stream overlap-save tiles)"
    fs, n = 10e6, 100_000_000
    taps = rr.low_pass(fs, 1e6, 190e3)
    assert len(taps) == 127
    w.blocks = [rr.FirFilter(taps)]
    w.n = n
    w.bufs = [shared_src(lambda: synth_real(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0006, dev), n, torch.float32),
              torch.empty(n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 8.0
    w.dominant, w.dominant_bytes_per_unit = 0, 8.0
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / (2 * (1024 - 126))
    w.kernel = "k_fftfilt_real"
    w.cpu = ("FirFilterFloat", taps)
    return w


def make_fir_fft_chain(dev, rank, world, shared_src, fused=True):
    """the north star's ">= 100x the CPU reference" pair: 127-tap FirFilter -> FftFilter(401 taps, ref 1024-pt)
    on the configs[1] input.  fused: ONE convolution with the composite taps t1 (*) t2 (rr.FirFftFilter); unfused: two
    blocks with a device-resident intermediate."""
    w = Workload()
    fs, n = 10e6, 100_000_000
    t1 = rr.low_pass_complex(fs, 1e6, 190e3)
    t2 = rr.low_pass_complex(fs, 1e6, 60e3)
    assert len(t1) == 127 and len(t2) == 401
    src = shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev), 2 * n, torch.float32)
    w.n = n
    if fused:
        w.name = ("FirFilter<Complex>(127 real taps) -> FftFilter(401 taps, ref fft_size 1024) fused into one 527-tap "
                  "convolution (rr.FirFftFilter), 10 Msps Complex<f32>, 100,000,000 samples/step")
        w.blocks = [rr.FirFftFilter(t1, t2)]
        w.bufs = [src, torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
        w.caps = [n + 1024]
        w.dominant = 0
    else:
        w.name = ("FirFilter<Complex>(127 real taps) -> FftFilter(401 taps, ref fft_size 1024), two blocks, 10 Msps "
                  "Complex<f32>, 100,000,000 samples/step")
        w.blocks = [rr.FirFilter(t1), rr.FftFilter(t2)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev), torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
        w.caps = [n, n + 1024]
        w.dominant = 1
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 16.0
    w.dominant_flops_per_unit = (2 * fft_flops(2048) + 6 * 2048) / (2048 - 526)
    w.kernel = "k_fftfilt_os"
    w.cpu = ("fir_fft_chain", (t1, t2))
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
    t1 = rr.low_pass_complex(fs, 1e6, 190e3)
    t2 = rr.low_pass_complex(fs, 1e6, 60e3)
    src = shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev), 2 * n, torch.float32)
    w.n = n
    w.alg_bytes_per_sample = 8.0 + 4.0 / 4.0
    oc = n // 4 + 1024
    if fused:
        w.name = ("full chain FirFilter(127)->FftFilter(401)->RationalResampler(1:4)->QuadratureDemod fused into one kernel "
                  "(rr.FirFmChain: composite 527-tap filter), 10 Msps Complex<f32>, 100,000,000 samples/step")
        w.blocks = [rr.FirFmChain(t1, t2, 1, 4, 1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(oc, dtype=torch.float32, device=dev)]
        w.caps = [oc]
        w.dominant, w.dominant_bytes_per_unit = 0, 9.0
        w.kernel = "k_fm_chain*"
        w.bound_note = CHAIN_BOUND_NOTE
    else:
        w.name = ("full chain FirFilter(127)->FftFilter(401)->RationalResampler(1:4)->QuadratureDemod, four blocks with "
                  "device-resident intermediates, 10 Msps Complex<f32>, 100,000,000 samples/step")
        w.blocks = [rr.FirFilter(t1), rr.FftFilter(t2), rr.RationalResampler(1, 4, np.complex64), rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev), torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev),
                  torch.empty(2 * oc, dtype=torch.float32, device=dev), torch.empty(oc, dtype=torch.float32, device=dev)]
        w.caps = [n, n + 1024, oc, oc]
        w.dominant, w.dominant_bytes_per_unit = 1, 16.0
        w.kernel = "k_fftfilt_os"
    w.dominant_flops_per_unit = (2 * fft_flops(2048) + 6 * 2048) / (2048 - 526)
    w.cpu = ("full_chain", (t1, t2))
    return w


def make_full_chain_fused(dev, rank, world, shared_src):
    return make_full_chain(dev, rank, world, shared_src, fused=True)


def make_fm_chain(dev, rank, world, shared_src, fused=True):
    w = Workload()
    how = "fused into one kernel (rr.FmChain)" if fused else "three blocks, device-resident intermediates"
    w.name = ("configs[2]: FftFilter(463 taps)->RationalResampler(1:6)->QuadratureDemod(exact atan2), 2.4 Msps x 10 s = "
              "24,000,000 samples/step, " + how)
    fs, n = 2.4e6, 24_000_000
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    assert len(taps) == 463
    src = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0003), 2 * n, torch.float32)
    if fused:
        w.blocks = [rr.FmChain(taps, 1, 6, 1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
        w.caps = [n // 6 + 1024]
        w.dominant_bytes_per_unit = 8.0 + 4.0 / 6.0
        w.kernel = "k_fm_chain*"
        w.bound_note = CHAIN_BOUND_NOTE
    else:
        w.blocks = [rr.FftFilter(taps), rr.RationalResampler(1, 6, np.complex64), rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)]
        w.bufs = [src,
                  torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev),
                  torch.empty(2 * (n // 6 + 1024), dtype=torch.float32, device=dev),
                  torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
        w.caps = [n + 1024, n // 6 + 1024, n // 6 + 1024]
        w.dominant_bytes_per_unit = 16.0
        w.kernel = "k_fftfilt_os"
    w.n = n
    w.alg_bytes_per_sample = 8.0 + 4.0 / 6.0
    w.dominant = 0
    # nominal work of the chain as the reference runs it per input sample: two 1024-point transforms + the product per
    # 561 samples (fft_filter.rs:172-176) + conj-multiply and atan2 per output
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / 561 + 40.0 / 6.0
    if fused:
        w.dominant_flops_exec_per_unit = poly_exec_flops_per_sample(463, 6)
    w.cpu = ("fm_chain", taps)
    return w


def make_rtl_fm_chain(dev, rank, world, shared_src):
    """configs[2] from the RTL-SDR wire format (examples/rtl_fm.rs:328-419): u8 I/Q pairs in, f32 out."""
    w = Workload()
    w.name = ("RtlSdrDecode->FftFilter(463 taps)->RationalResampler(1:6)->QuadratureDemod(exact atan2) fused into one "
              "kernel (rr.FmChainU8), 2.4 Msps x 10 s = 24,000,000 samples/step, u8 I/Q input")
    fs, n = 2.4e6, 24_000_000
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    f32 = synth_fm(n, fs, dev, 0x5EED0003)
    src = torch.clamp(torch.round(f32 / 0.008 + 127.0), 0, 255).to(torch.uint8)      # what the dongle delivers
    w.blocks = [rr.FmChainU8(taps, 1, 6, 1.0, rr.ATAN2_EXACT)]
    w.bufs = [src, torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
    w.caps = [n // 6 + 1024]
    w.in_mult = 2
    w.dtype = "u8->f32"
    w.n = n
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 2.0 + 4.0 / 6.0
    w.dominant = 0
    w.dominant_flops_per_unit = (2 * fft_flops(1024) + 6 * 1024) / 561 + 40.0 / 6.0
    w.kernel = "k_fm_chain*"
    w.bound_note = CHAIN_BOUND_NOTE
    w.cpu = ("rtl_fm_chain", taps)
    return w


def make_rtl_fm_example(dev, rank, world, shared_src):
    """examples/rtl_fm.rs:328-419 with its own numbers: 1.024 Msps RTL-SDR bytes, low_pass_complex(fs, 100 kHz, 1 kHz)
    = 2467 taps (reference fft_size 8192), resampled 1,024,000 -> 200,000 (25:128), quadrature demod."""
    w = Workload()
    fs, n = 1.024e6, 24_000_000
    taps = rr.low_pass_complex(fs, 100e3, 1e3)
    w.name = (f"examples/rtl_fm.rs front end: RtlSdrDecode->FftFilter({len(taps)} taps)->RationalResampler(25:128)->QuadratureDemod "
              "fused (rr.FmChainU8), 1.024 Msps u8 I/Q, 24,000,000 samples/step")
    f32 = synth_fm(n, fs, dev, 0x5EED0006)
    src = torch.clamp(torch.round(f32 / 0.008 + 127.0), 0, 255).to(torch.uint8)
    w.blocks = [rr.FmChainU8(taps, 200000, 1024000, 1.0, rr.ATAN2_EXACT)]
    cap = n * 25 // 128 + 4096
    w.bufs = [src, torch.empty(cap, dtype=torch.float32, device=dev)]
    w.caps = [cap]
    w.in_mult = 2
    w.dtype = "u8->f32"
    w.n = n
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 2.0 + 4.0 * 25 / 128
    w.dominant = 0
    w.dominant_flops_per_unit = (2 * fft_flops(8192) + 6 * 8192) / 5725 + 40.0 * 25 / 128
    w.kernel = "k_fm_chain_split"
    w.cpu = ("rtl_fm_example", taps)
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
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    total = multi.CFG4_CHANNELS
    chans = list(multi.shard_channels(per_gpu * world, world, rank))
    w.name = (f"configs[3]: {len(chans)} FM channels/GPU (FftFilter 463 taps->RationalResampler 1:6->QuadratureDemod, fused, "
              f"rr.FmMulti) on one shared 2.4 Msps IQ source, {n:,} samples/step/channel"
              + (f"; this rank: channels {chans[0]}..{chans[-1]} of the {total}-channel bank, {per_gpu * world} in the job" if world > 1 else ""))
    if u8:      # the RTL-SDR wire format as the fan-out format: 2 B instead of 8 B per sample over xGMI, decoded in the kernel
        src = shared_src(lambda: torch.clamp(torch.round(synth_fm(n, fs, dev, 0x5EED0004) / 0.008 + 127.0), 0, 255).to(torch.uint8),
                         2 * n, torch.uint8)
        w.in_mult, w.dtype = 2, "u8->f32"
        w.name = w.name.replace("one shared 2.4 Msps IQ source", "one shared 2.4 Msps RTL-SDR byte stream (rr.FmMultiU8)")
    else:
        src = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0004), 2 * n, torch.float32)
    taps_all = multi.cfg4_taps(taps, chans, total)
    blk = (rr.FmMultiU8 if u8 else rr.FmMulti)(taps_all, 1, 6, 1.0, rr.ATAN2_EXACT)     # one kernel: forward FFT shared by all channels
    w.blocks = [blk]
    w.n = n
    cap = n // 6 + 1024
    w.outs = torch.empty(len(chans) * cap, dtype=torch.float32, device=dev)
    nch = len(chans)
    w.units_per_sample = nch
    bin_ = 2.0 if u8 else 8.0
    w.alg_bytes_per_sample = bin_ / nch + 4.0 / 6.0       # shared read: 8/N (2/N) B in + 0.67 B out per channel-sample
    w.dominant, w.dominant_bytes_per_unit = 0, (bin_ / nch + 4.0 / 6.0) * nch
    # vector-FP32-bound: per channel and input sample the reference's filter work (two 1024-point transforms + product
    # per 561 samples) + demod, the forward transform shared by the channels of one GPU
    w.bound = "vector_fp32"
    w.dominant_flops_per_unit = fft_flops(1024) / 561 + nch * ((fft_flops(1024) + 6 * 1024) / 561 + 40.0 / 6.0)
    # ... and what k_fm_multi_poly<6> executes: 6 shared phase transforms + per channel 6 x 1024 multiply-adds, ONE inverse
    # transform and the demodulation per 946 outputs = 5676 inputs (~4x less than the nominal count)
    w.dominant_flops_exec_per_unit = poly_exec_flops_per_sample(463, 6, nch)
    w.kernel = "k_fm_multi*"
    w.cpu = ("fm_chain", taps)
    w.bufs = [src]

    def step(stream, src_ptr=None):
        st, c, p, need = blk.work_dev(src.data_ptr() if src_ptr is None else src_ptr, n * w.in_mult, w.outs.data_ptr(), cap, stream)
        c //= w.in_mult
        w.dom_units += c
        return c * nch
    w.step = step
    return w


def make_fm_multi_u8(dev, rank, world, shared_src):
    """configs[3] with the fan-out in the RTL-SDR wire format: the broadcast moves 2 B per sample"""
    return make_fm_multi(dev, rank, world, shared_src, u8=True)


def make_channelizer(dev, rank, world, shared_src, fused=True, rotator=None, as_rank=None):
    """BASELINE configs[4]; on N > 1 GPUs rank g runs channel offset f_g through FirFilter::translate(100e6, f_g)
    (src/fir.rs:476-486, SURVEY §8d cfg5: multi.cfg5_translate_hz) — with the library's DEFAULT rotator, the reference's own
    f32 recurrence replayed bit for bit (RR_ROT_REPLAY: on parity for any stream length, one sequential chain per block,
    walked ahead of the filter — by one device lane at 10 ns per output while the block's calls leave it time, by a host
    thread at ~2.6 ns per output once they do not (round 5; back-to-back bench steps do not)); `channelizer_model` is the same with
    the opt-in f64 closed form (parallel, but outside the 1e-5 parity bar beyond ~1e5 outputs of a stream)."""
    w = Workload()
    if as_rank is not None:                      # (N = 1 line: the workload ONE rank of the 8-GPU variant runs)
        rank, world = as_rank
    f_g = multi.cfg5_translate_hz(rank, world)
    rotator = rr.ROT_REPLAY if rotator is None else rotator
    rot_txt = ("rotator=replay: the reference's f32 recurrence bit for bit, the library default, ON parity; back-to-back steps are "
               "bound by that sequential chain, not by the filter" if rotator == rr.ROT_REPLAY else
               "rotator=model: opt-in f64 closed form, parallel, OFF parity beyond ~1e5 outputs of a stream")
    how = ("fused into one composite decimating FIR (rr.HilbertFir)" if fused
           else "two blocks, device-resident analytic stream")
    w.name = ("configs[4]: Hilbert(65)->FirFilter<Complex>(255 real taps, deci 8), 100 Msps f32 x 1 s = 100,000,000 samples/step, "
              + how + (f", .translate(100e6, {f_g / 1e6:.3f} MHz) on this rank ({rot_txt})" if world > 1 else ""))
    w.rotator = None if world == 1 else ("replay" if rotator == rr.ROT_REPLAY else "model")
    fs, n = 100e6, 100_000_000
    taps = rr.low_pass_complex(fs, 5e6, 943e3)
    assert len(taps) == 255
    src = shared_src(lambda: synth_real(n, fs, (3e6, 12e6, 37e6), 0x5EED0005, dev), n, torch.float32)
    tr = (fs, f_g) if world > 1 else None
    w.n = n
    w.alg_bytes_per_sample = 5.0
    if fused:
        w.blocks = [rr.HilbertFir(65, taps, 8, translate=tr, rotator=rotator)]
        w.bufs = [src, torch.empty(2 * (n // 8 + 8), dtype=torch.float32, device=dev)]
        w.caps = [n // 8 + 8]
        w.dominant, w.dominant_bytes_per_unit = 0, 5.0
        w.kernel = "k_fftfilt_prune"
        w.dominant_flops_per_unit = (fft_flops(2048) + 2 * 6 * 2048 + 2 * fft_flops(256)) / (2 * (2048 - 318))
    else:
        w.blocks = [rr.Hilbert(65), rr.FirFilter(taps, deci=8, translate=tr, rotator=rotator)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev),
                  torch.empty(2 * (n // 8 + 8), dtype=torch.float32, device=dev)]
        w.caps = [n, n // 8 + 8]
        w.dominant, w.dominant_bytes_per_unit = 0, 12.0
        w.kernel = "k_hilbert"
        w.dominant_flops_per_unit = 2.0 * 33 + 2
    w.cpu = ("channelizer", taps)
    return w


def make_channelizer_unfused(dev, rank, world, shared_src):
    return make_channelizer(dev, rank, world, shared_src, fused=False)


def make_channelizer_translate(dev, rank, world, shared_src):
    """what ONE rank of configs[4]'s 8-GPU variant runs (rank 1 of 8: .translate(100e6, f_1), default on-parity rotator), on
    the N = 1 line so that the driver sees it: bound by the sequential rotator chain, not by a roofline"""
    w = make_channelizer(dev, rank, world, shared_src, as_rank=(1, 8))
    w.bound = "sequential_rotator"
    w.bound_note = ("FirFilter::translate's rotator (src/fir.rs:464-473) is an un-renormalised f32 recurrence, one dependent chain of 12.5 M "
                    "steps per step here; replayed bit for bit it runs at 2.6 ns per output on a host core (10 ns on a device lane), "
                    "whatever the filter kernel does (0.14 ms).  rotator_ns_per_output is this step's time per output")
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


# ---- measurement ------------------------------------------------------------------------------
def run_timed(w, steps, warmup, dist, stream, fan=None, src_ptr=None, settle_ms=0.0):
    """W untimed warm-up steps, then EXACTLY `steps` timed steps bracketed by barrier + synchronize on both sides.
    With `fan` (multi.TileFanout) every step's input is the tile rank 0 produced and broadcast during the previous
    step.  -> units, wall seconds, dominant-kernel ms, launches, dominant units, per-step times (ms, HIP events)

    The timed region carries NO instrumentation at all (round 2 kept the dominant block's two
    HIP-event brackets per launch inside it, ~8 us of stream time per step — 10 % of the 0.07 ms fm_chain step, charged
    to `value`).  Two more passes of the same `steps` steps follow it back to back, at the same sustained clocks:
    pass 2 = the library's launch brackets around the dominant kernel on its launch stream (rr_block_set_profiling;
    `roofline.achieved` = algorithmic bytes per launch / that kernel's mean duration), pass 3 = one HIP-event pair per
    step for `ms_per_step_median`."""
    for b in w.blocks:
        b.set_profiling(False)
    w.dom_units = 0
    cs = stream.cuda_stream
    t = 0

    k = getattr(fan, "tile_steps", 1)           # steps of source per fanned-out tile
    step_bytes = getattr(fan, "step_bytes", 0)

    def one(t):
        if fan is None:
            return w.step(cs, src_ptr)           # (src_ptr: a tile already resident on this rank instead of bufs[0])
        T, sub = divmod(t, k)
        if sub == 0:
            fan.prefetch(T + 1)                  # tile T+1 travels while the k steps of tile T are computed
        x = fan.acquire(T, stream)               # (idempotent; the compute stream waits for the tile's fan-out)
        u = w.step(cs, x.data_ptr() + sub * step_bytes)
        if sub == k - 1:
            fan.release(T, stream)
        return u

    if fan is not None:
        fan.prefetch(0)
    # Settle (untimed, before the W warm-up steps): the same step back to back for ~settle_ms of GPU time.  From idle the
    # first few passes run at boost clocks, the power controller then clamps hard and relaxes to its equilibrium over the
    # next ~40 ms (tools/step_series.py, FftFilter, ms per step: 0.33 0.33 0.34 | 0.40 0.42 0.45 0.47 ... | 0.38 by step 30,
    # 0.357 by step 50, 0.342 from step 100 on for as long as the load lasts).  A streaming graph runs for hours: the
    # sustained rate is the one to report, and W = 5 warm-up steps end in the middle of the dip.
    n_settle = 0
    w.cold_ms_per_step = None
    if settle_ms > 0 and fan is None and dist is None and getattr(w, "report_cold", False):
        # for the record: the same W + K steps straight from idle, i.e. what the line would say without the settle phase
        for _ in range(warmup):
            one(t); t += 1
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(steps):
            one(t); t += 1
        torch.cuda.synchronize()
        w.cold_ms_per_step = (time.perf_counter() - c0) / max(steps, 1) * 1e3
    if settle_ms > 0:
        for _ in range(2):                       # (the very first launch of a kernel pays its one-time set-up)
            one(t); t += 1
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(4):
            one(t); t += 1
        e1.record(stream)
        torch.cuda.synchronize()
        est = max(e0.elapsed_time(e1) / 4, 1e-3)
        n_settle = int(min(4000, max(40, settle_ms / est)))
        if dist is not None:                     # every rank runs the same number of steps (the fan-out is collective)
            ns = torch.tensor([n_settle], dtype=torch.int64, device=stream.device)
            dist.all_reduce(ns, op=dist.ReduceOp.MAX)
            n_settle = int(ns.item())
        for _ in range(n_settle):
            one(t); t += 1
        n_settle += 6
    w.settle_steps = n_settle
    for _ in range(warmup):
        one(t); t += 1
    torch.cuda.synchronize()
    if fan is not None:
        fan.reset_timing()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    units = 0
    for i in range(steps):
        units += one(t); t += 1
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # second pass: the dominant kernel alone (the library's HIP-event brackets on the launch stream)
    w.blocks[w.dominant].set_profiling(True)
    w.dom_units = 0
    for i in range(steps):
        one(t); t += 1
    torch.cuda.synchronize()
    kms, launches = w.blocks[w.dominant].profile(reset=True)
    w.blocks[w.dominant].set_profiling(False)
    dom_units = w.dom_units
    # third pass: per-step durations for the median
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for i in range(steps):
        evs[i][0].record(stream)
        one(t); t += 1
        evs[i][1].record(stream)
    torch.cuda.synchronize()
    step_ms = [a.elapsed_time(b) for a, b in evs]
    w.dom_units = dom_units
    return units, dt, kms, launches, dom_units, step_ms


# ---- CPU baseline (the oracle; test infrastructure used here as the reported baseline only) ----------------------
def _cpu_chain(kind, taps):
    from oracle import cpu_worker
    return cpu_worker.chain_for(kind, taps)


def _cpu_graph_1thread(chain, host, win, in_mult, seconds):
    from oracle import cpu_worker
    return cpu_worker.graph_1thread(chain, host, win, in_mult, seconds)


def _cpu_mtgraph(chain, host, win, in_mult, seconds):
    """MTGraph (src/mtgraph.rs:77-120): one OS thread per block, bounded rings between them (the oracle's C work()
    runs outside the GIL) -> (samples fed, seconds)"""
    import queue
    nwin = len(host) // win
    qs = [queue.Queue(maxsize=2) for _ in chain]
    stop = threading.Event()

    def stage(j):
        b, ring = chain[j], np.zeros(0, chain[j].in_dtype)
        while not stop.is_set():
            try:
                ring = np.concatenate([ring, qs[j].get(timeout=0.1)])
            except queue.Empty:
                continue
            while True:
                st, c, p, need, out = b.work(ring, 4_096_000 // b.out_dtype.itemsize)
                ring = ring[c:]
                if j + 1 < len(chain) and len(out):
                    while not stop.is_set():
                        try:
                            qs[j + 1].put(out, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                if st == 1 or (c == 0 and p == 0):
                    break

    th = [threading.Thread(target=stage, args=(j,), daemon=True) for j in range(len(chain))]
    for t in th:
        t.start()
    t0 = time.perf_counter()
    fed = i = 0
    while time.perf_counter() - t0 < seconds:
        chunk = host[(i % nwin) * win:(i % nwin + 1) * win]
        i += 1
        while True:
            try:
                qs[0].put(chunk, timeout=0.1)
                break
            except queue.Full:
                if time.perf_counter() - t0 >= seconds:
                    break
        fed += len(chunk) // in_mult
    dt = time.perf_counter() - t0
    stop.set()
    for t in th:
        t.join(timeout=5)
    return fed, dt


def native_oracle():
    """SURVEY §8d / VERDICT r2 #2e: the CPU baseline runs the oracle built FOR THIS HOST — `gcc -O3 -march=native
    -ffp-contract=off` (the reference's docs recommend `-Ctarget-cpu=native`), compiled here at bench time into
    oracle/_native/ under a name that carries the host's CPU identity (a build from another machine is never loaded).  The
    portable -O2 build (oracle/liboracle.so) stays what the parity tests use.  -> (library path or None, flags text)"""
    import platform
    try:
        ident = platform.machine()
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith(("model name", "flags")):
                    ident += line
                if line.startswith("flags"):
                    break
        tag = hashlib.sha256(ident.encode()).hexdigest()[:12]
        odir = os.path.join(ROOT, "oracle", "_native")
        os.makedirs(odir, exist_ok=True)
        lib = os.path.join(odir, f"liboracle_{tag}.so")
        src = os.path.join(ROOT, "oracle", "rr_oracle.c")
        flags = "-O3 -march=native -std=c11 -fPIC -ffp-contract=off -fno-fast-math -fno-unsafe-math-optimizations"
        if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
            subprocess.run(["gcc"] + flags.split() + ["-shared", "-o", lib, src, "-lm"], check=True,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        return lib, "gcc " + flags
    except Exception as e:                      # no compiler on the host: the portable build, and the line says so
        print(f"bench.py: native oracle build failed ({e}); timing the portable -O2 build", file=sys.stderr)
        return None, "gcc -O2 -ffp-contract=off (portable build; the native build failed)"


def cpu_baseline(w, seconds=8.0):
    """The oracle (strict-order C restatement of the reference blocks, oracle/rr_oracle.c) timed on the host over
    512,000-sample work() windows (src/stream.rs:105) of the same synthetic input.  `value` = the whole chain on ONE
    thread (the reference's Graph); `modes` adds BASELINE.md §3's other two: one thread per block (MTGraph) and one
    independent chain per core on all cores."""
    lib_path, flags = native_oracle()
    if lib_path:
        os.environ["RR_ORACLE_LIB"] = lib_path  # read by oracle/pyoracle.py at its first use (this process and the workers)
    kind, taps = w.cpu
    win = 512_000
    if kind in ("channelizer", "FirFilterFloat"):
        host = w.bufs[0][:win * 2 * 16].cpu().numpy()
        win = 1_024_000
    elif kind in ("rtl_fm_example", "rtl_fm_chain"):
        win = 4_096_000                                       # a full u8 ring (src/stream.rs:105)
        host = w.bufs[0][:win * 4].cpu().numpy()
    else:
        host = w.bufs[0][:2 * win * 16].cpu().numpy().view(np.complex64)
    fed, dt = _cpu_graph_1thread(_cpu_chain(kind, taps), host, win, w.in_mult, seconds)
    base = fed / dt / 1e6
    chain = _cpu_chain(kind, taps)
    modes = {"graph_1_thread": {"msamples_per_s": round(base, 3), "threads": 1}}
    if len(chain) > 1:
        f2, d2 = _cpu_mtgraph(chain, host, win, w.in_mult, seconds / 2)
        modes["mtgraph_thread_per_block"] = {"msamples_per_s": round(f2 / d2 / 1e6, 3), "threads": len(chain) + 1}
    # one independent chain per core on all cores: child processes (they never touch the GPU), each with 2 windows of
    # the same input
    cores = os.cpu_count() or 1
    import tempfile
    tmp = tempfile.NamedTemporaryFile(suffix=".npz", dir="/dev/shm" if os.path.isdir("/dev/shm") else None, delete=False)
    tmp.close()
    try:
        arrs = {"host": host[:2 * win], "kind": np.array(kind), "win": np.array(win), "in_mult": np.array(w.in_mult)}
        if isinstance(taps, tuple):
            arrs["taps0"], arrs["taps1"] = taps
        else:
            arrs["taps0"] = taps
        np.savez(tmp.name, **arrs)
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "oracle", "cpu_worker.py"), tmp.name, str(seconds / 2)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env) for _ in range(cores)]
        tot = 0.0
        ok = 0
        for pr in procs:
            o, _ = pr.communicate(timeout=seconds * 10 + 120)
            try:
                f_, d_ = o.decode().split()
                tot += float(f_) / float(d_)
                ok += 1
            except Exception:
                pass
        modes["one_chain_per_core_all_cores"] = {"msamples_per_s": round(tot / 1e6, 3), "processes": ok}
    finally:
        os.unlink(tmp.name)
    return {"value": round(base, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{fed} input samples of the same synthetic stream in {win}-sample work() windows, {dt:.1f} s, 1 thread, "
                      f"{flags} built on this host at bench time, strict f32 (own scalar radix-4 FFT, not rustfft's SIMD kernels)",
            "build": flags, "host_cores": cores, "modes": modes}


def cpu_1thread(w, seconds):
    """the oracle chain of workload `w` on ONE host thread over 512,000-sample work() windows of the same synthetic input
    (the first leg of cpu_baseline, on its own) -> {value, unit, cores, kind, sample}"""
    lib_path, flags = native_oracle()
    if lib_path:
        os.environ["RR_ORACLE_LIB"] = lib_path
    kind, taps = w.cpu
    win = 512_000
    host = w.bufs[0][:2 * win * 8].cpu().numpy().view(np.complex64)
    fed, dt = _cpu_graph_1thread(_cpu_chain(kind, taps), host, win, w.in_mult, seconds)
    return {"value": round(fed / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{fed} input samples of the same synthetic stream in {win}-sample work() windows, {dt:.1f} s, 1 thread, {flags}"}


def _sources_hash():
    """sha256 over the kernel sources: profiles/traffic.json is only valid for the kernels it was collected on"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rustradio_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # kernels, their headers and the block logic that picks between them (not the ABI / fan-out / ring plumbing)
        if (f.endswith((".hip", ".hpp")) and f != "dstream.hpp") or f == "blocks.cpp":
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/traffic.json,
    written by tools/pmc_traffic.py with the kernel-source hash it was collected at) -> (bytes or None, note)"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            d = json.load(f)
    except Exception:
        return None, "profiles/traffic.json missing"
    e = d.get(workload)
    if not e:
        return None, "not collected for this workload"
    if e.get("sources_sha16") != _sources_hash():
        return None, f"stale: collected at kernel sources {e.get('sources_sha16')}, tree is {_sources_hash()} (re-run tools/pmc_run.sh)"
    return e.get("hbm_bytes_per_launch"), f"rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE (separate passes), kernel sources {e['sources_sha16']}"


def parity_report():
    """the line's `parity` object: the tolerance the parity tests hold the chains to, and how much of it is used — from
    profiles/parity_allowance.json (tests/parity_allowance.py on the GPU box, against the oracle), valid for the kernel
    sources it was measured on"""
    base = {"tol": 1e-5, "per_block": "max|y - y_ref| / max|y_ref| <= tol for every block alone (tests/test_gpu_parity.py)",
            "chain_bound": "propagated"}
    try:
        with open(os.path.join(ROOT, "profiles", "parity_allowance.json")) as f:
            d = json.load(f)
    except Exception:
        return dict(base, above_plain_share=None, note="profiles/parity_allowance.json missing")
    if d.get("kernel_sources") != _sources_hash():
        return dict(base, above_plain_share=None,
                    note=f"stale: measured at kernel sources {d.get('kernel_sources')}, tree is {_sources_hash()} (python -m tests.parity_allowance)")
    sm = d["summary"]
    return dict(base, chain_bound_formula=d["chain_bound"], used_max=sm["used_max"],
                above_plain_share={"cfg3_centred_station": sm["above_plain_share_cfg3_centred"],
                                   "cfg3_station_150kHz_off_centre": sm["above_plain_share_cfg3_off_centre"],
                                   "cfg4_worst_of_32_channels": sm["above_plain_share_cfg4_worst_channel"]},
                kernel_sources=d["kernel_sources"], source="profiles/parity_allowance.json")


# ---- the drop-in path (others.dropin_*): rr_block_work on HOST windows, as the Rust shim calls it -------------------
_RINGS = {}


def _registered_ring(which, like):
    """a 4,096,000-byte page-locked ring per direction, registered once; returned as a view of `like`'s dtype and length"""
    a = _RINGS.get(which)
    if a is None:
        a = rr.host_ring(4_096_000)             # page-aligned whole pages: what the library grants zero-copy windows on
        rr.host_register(a)
        _RINGS[which] = a
    return a[:like.nbytes].view(like.dtype)



def dropin_host_windows(kind, registered, seconds=1.5):
    """`rr_block_work` on reference-sized 4,096,000-byte host windows (src/stream.rs:105,208-217,301-310): the shim hands
    read_buf()/write_buf() windows of the reference's rings; `registered` = the ring mappings page-locked once with
    rr_host_register.  -> Msamples/s (input samples of the first block, wall clock incl. PCIe)"""
    rng = np.random.default_rng(7)
    if kind == "copy":          # the path's own ceiling: a block that only moves the window (x * 1.0), same bytes each way
        blk = rr.MultiplyConst(1.0)
        n_in = 4_096_000 // 4
        x = rng.uniform(-1, 1, n_in).astype(np.float32)
        mult = 1
    elif kind == "fftfilter":
        taps = rr.low_pass_complex(10e6, 1e6, 60e3)
        blk = rr.FftFilter(taps)
        n_in = 4_096_000 // 8
        x = (rng.uniform(-1, 1, n_in) + 1j * rng.uniform(-1, 1, n_in)).astype(np.complex64)
        mult = 1
    else:                       # examples/rtl_fm.rs front end from the RTL-SDR byte ring, fused
        taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
        blk = rr.FmChainU8(taps, 1, 6, 1.0, rr.ATAN2_EXACT)
        n_in = 4_096_000
        x = rng.integers(0, 256, n_in, dtype=np.uint8)
        mult = 2
    out_cap = 4_096_000 // blk.out_dtype.itemsize
    out = np.zeros(out_cap, blk.out_dtype)
    if registered:
        # the two rings of a stream pair, page-locked ONCE per process like the shim's (an address range registered a
        # second time is retired from zero-copy by the library: csrc/blocks.cpp "RETIRED addresses")
        xin, out = _registered_ring("in", x), _registered_ring("out", out)
        xin[:] = x
        x = xin
    try:
        fed, t0 = 0, None
        i = 0
        while True:
            st, c, p, need = blk.work_into(x, out, out_cap)
            if i == 3:
                t0, fed = time.perf_counter(), 0
            fed += c // mult
            i += 1
            if t0 is not None and time.perf_counter() - t0 > seconds:
                break
        dt = time.perf_counter() - t0
    finally:
        pass
    return round(fed / dt / 1e6, 1)


def devgraph_ref_rings(fused, seconds=1.5):
    """the configs[2] graph device-resident with reference-sized 4,096,000-byte HBM rings (rr_dstream): a host source
    pushes windows in, blocks run ring to ring (rr_block_work_streams), a NullSink consumes.  Python drives it (ctypes)."""
    fs = 2.4e6
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    rng = np.random.default_rng(9)
    x = rr.host_ring(4_096_000).view(np.complex64)         # a page-aligned source ring, as the shims' (copy_in reads it in place)
    x[:] = (rng.uniform(-1, 1, 512_000) + 1j * rng.uniform(-1, 1, 512_000)).astype(np.complex64)
    rr.host_register(x)
    try:
        blocks = ([rr.FmChain(taps, 1, 6, 1.0, rr.ATAN2_EXACT)] if fused else
                  [rr.FftFilter(taps), rr.RationalResampler(1, 6, np.complex64), rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)])
        rings = [rr.DeviceStream(blocks[0].in_dtype)] + [rr.DeviceStream(b.out_dtype) for b in blocks]
        fed, t0, rounds = 0, None, 0
        while True:
            fed += rings[0].push(x)
            for i, b in enumerate(blocks):
                b.work_streams(rings[i], rings[i + 1])
            rings[-1].discard()                      # NullSink: consume without copying (null_sink.rs:15-25)
            rounds += 1
            if rounds == 20:
                torch.cuda.synchronize(); t0, fed = time.perf_counter(), 0
            if t0 is not None and rounds % 50 == 0:
                torch.cuda.synchronize()
                if time.perf_counter() - t0 > seconds:
                    break
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        rr.host_unregister(x)
    return round(fed / dt / 1e6, 1)


def dropin_report():
    out = {}
    # the ceiling of the path itself: 4,096,000 bytes down and 4,096,000 up per call through a kernel that does nothing else
    cps = dropin_host_windows("copy", True) * 1e6 / (4_096_000 // 4)           # calls per second
    out["dropin_ceiling"] = {
        "what": "rr_block_work on a block that only copies (MultiplyConst(1.0), f32): 4,096,000-byte registered HOST windows, "
                "in place over PCIe both ways, one call at a time (launch + completion wait included)",
        "us_per_call": round(1e6 / cps, 1), "gbs_each_way": round(4_096_000 * cps / 1e9, 2),
        "pcie_gen5_x16_gbs_each_way_spec": 63.0,
        "link_note": "a bare copy kernel moves such a window at 55 GB/s one way and at 32 GB/s EACH way when both directions run at "
                     "once (64 GB/s combined: profiles/r05_pcie_inplace.txt, tools/micro/pcie_inplace.hip) — 128 us per window pair "
                     "before any launch or wait"}
    LINK_COMBINED_GBS = 64.2                       # profiles/r05_pcie_inplace.txt: host -> host, both ways at once
    for kind in ("fftfilter", "rtl_fm"):
        ms_reg = dropin_host_windows(kind, True)
        n_in = 4_096_000 // 8 if kind == "fftfilter" else 4_096_000 // 2          # input samples per call
        b_in, b_out = 4_096_000, (4_096_000 if kind == "fftfilter" else 4_096_000 // 2 // 6 * 4)
        us_call = n_in / ms_reg
        out[f"dropin_{kind}"] = {
            "what": ("rr_block_work, FftFilter 401 taps" if kind == "fftfilter" else
                     "rr_block_work, RtlSdrDecode>FftFilter(463)>RationalResampler(1:6)>QuadratureDemod fused, u8 in")
                    + ", 4,096,000-byte HOST windows in and out, wall clock incl. PCIe",
            "msamples_per_s_registered_rings": ms_reg,
            "us_per_call": round(us_call, 1),
            "bytes_per_call_in_out": [b_in, b_out],
            "link_floor_us": round((b_in + b_out) / LINK_COMBINED_GBS / 1e3, 1),
            "frac_of_link_floor": round((b_in + b_out) / LINK_COMBINED_GBS / 1e3 / us_call, 3),
            "frac_of_copy_block": round((1e6 / cps) * (b_in + b_out) / 8_192_000 / us_call, 3),
            "msamples_per_s_pageable": dropin_host_windows(kind, False)}
    out["devgraph_ref_rings"] = {
        "what": "configs[2] graph over 4,096,000-byte HBM rings (rr_dstream), registered host source -> NullSink, Python driver",
        "msamples_per_s_three_blocks": devgraph_ref_rings(False),
        "msamples_per_s_fused": devgraph_ref_rings(True)}
    return out


# ---- launch ----------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


_STAGE = {"name": "start", "t0": time.monotonic()}


def stage(name):
    """N > 1: where this rank is — the watchdog names it if a collective never returns"""
    _STAGE["name"], _STAGE["t0"] = name, time.monotonic()


def start_watchdog(rank, limit_s):
    """N > 1 only.  A collective that one rank never enters blocks every other rank for ever and the job dies without a word
    when its launcher's own limit expires; this thread says WHICH stage of WHICH rank stood still (all thread stacks on
    stderr) and ends the rank, so the launcher tears the job down at once.  No number is invented: there is no JSON line."""
    import faulthandler

    def watch():
        while True:
            time.sleep(min(5.0, max(0.05, limit_s / 4)))
            held = time.monotonic() - _STAGE["t0"]
            if held > limit_s:
                print(f"bench.py: rank {rank} has been in stage {_STAGE['name']!r} for {held:.0f} s (limit {limit_s:.0f} s, "
                      "--stage-timeout): giving up, no result line", file=sys.stderr, flush=True)
                faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                os._exit(4)
    threading.Thread(target=watch, name="bench-watchdog", daemon=True).start()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script with torch.distributed.run —
    from this still GPU-free process (nothing here has initialised HIP) — and exit with their code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: fftfilter (configs[1]) on one GPU, fm_multi (configs[3]) on N > 1")
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other workloads")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-dropin", action="store_true", help="skip the host-window (drop-in path) measurements")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--fanout", choices=("torch", "abi"), default=None,
                    help="N > 1: the source fan-out through the library's own rr_fanout_* entry points (RCCL bound by the C ABI, what "
                         "a Rust graph would call: the default whenever every rank has its own GPU) or through torch.distributed "
                         "(the default on the gloo fallback of a box with fewer GPUs than ranks)")
    ap.add_argument("--fanout-algo", choices=("auto", "bcast", "scatter_allgather"), default="auto",
                    help="N > 1, --fanout torch: one broadcast per tile, scatter + all-gather over the xGMI mesh, or (default) "
                         "whichever is faster on this job's fabric, timed before the run")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed passes of the workload before the warm-up steps, until the power controller's start-up "
                         "transient is over (tools/step_series.py); 0 = none")
    ap.add_argument("--stage-timeout", type=float, default=600.0,
                    help="N > 1: seconds one stage of a rank (group set-up, fan-out self-check, calibration, a timed run) may take "
                         "before the rank reports where it stands and exits (a collective one rank never entered hangs the others)")
    ap.add_argument("--tile-steps", type=int, default=4,
                    help="N > 1: steps of source per fanned-out tile (the fan-out of a tile costs the host ~0.1 ms through "
                         "torch.distributed, as much as one 0.09 ms step: tools/fanout_overhead.py)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="rr_build_opts override for every block built (e.g. fft_log2f=11, fir_path=direct)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    # (ranks started by a launcher of the caller's: the host driver of this pool only supports dmabuf IPC, and RCCL's
    #  buffer exchange between processes fails without this — nothing has touched the GPU yet)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()          # (does not initialise the GPU on this image)
    if ndev == 0 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (rustradio_amd has no CPU path)")
    dev_idx = local_rank % ndev
    torch.cuda.set_device(dev_idx)
    rr.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    opts = {}
    for kv in args.opt:
        k, v = kv.split("=", 1)
        opts[k] = v if k == "fir_path" else int(v)
    dist, backend = None, None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        start_watchdog(rank, args.stage_timeout)
        stage("init_process_group")
        # RCCL needs one GPU per rank; with fewer visible devices (the 1-GPU box) the ranks share devices and the
        # collective runs over gloo — same code path, not a performance configuration (stated in the JSON line)
        backend = "nccl" if ndev >= world else "gloo"
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        dist = dist_mod
    if args.fanout is None:
        args.fanout = "abi" if backend == "nccl" else "torch"

    def shared_src(gen, numel, dtype):
        """the shared IQ source lives on rank 0 (it produces the tiles); the other ranks only receive broadcasts"""
        return gen() if rank == 0 else torch.empty(0, dtype=dtype, device=dev)

    wname = args.workload or ("fftfilter" if world == 1 else "fm_multi")
    stream = torch.cuda.current_stream()
    stage("build workload")
    with rr.build_options(**opts):
        w = WORKLOADS[wname](dev, rank, world, shared_src)
    abi_check = {}

    def make_fan(wl):
        """the streaming fan-out of wl's shared source (resident on rank 0): one tile = --tile-steps steps of input"""
        if world == 1:
            return None
        store = wl.bufs[0]
        meta = torch.tensor([store.numel() if rank == 0 else 0], dtype=torch.int64, device=dev)
        dist.broadcast(meta, src=0)
        sdtype = torch.uint8 if wl.in_mult == 2 else torch.float32

        k = max(1, args.tile_steps)
        step_elems = int(meta.item())

        def produce(t, out):                  # rank 0: the source block writes the k steps of tile t into the ring half
            out.view(k, step_elems).copy_(store.unsqueeze(0).expand(k, step_elems), non_blocking=True)

        def tag(f):
            f.tile_steps, f.step_bytes = k, step_elems * (1 if sdtype == torch.uint8 else 4)
            if k >= 2:
                f.TIME_EVERY = 1              # a fan-out every k steps: timing each one costs the stream little
            return f
        if args.fanout == "abi":
            if backend != "nccl":
                raise SystemExit("bench.py --fanout abi: rr_fanout_* binds RCCL, which needs one GPU per rank")
            if "ok" not in abi_check:            # once per job: known tiles through both algorithms, checksums on every rank
                stage("rr_fanout_* self-check (RCCL communicator of the C ABI)")
                abi_check["ok"], abi_check["why"] = multi.verify_abi_fanout(rr, dist, rank, dev)
                if not abi_check["ok"] and rank == 0:
                    print(f"bench.py: rr_fanout_* failed its self-check on this group ({abi_check['why']}); "
                          "the fan-out runs through torch.distributed instead", file=sys.stderr)
            if abi_check["ok"]:
                algo, cal = args.fanout_algo, None
                if algo == "auto":               # timed on this group's fabric at this tile size, like the torch.distributed form
                    stage("rr_fanout_* calibration (bcast against scatter + all-gather)")
                    algo, cal = multi.calibrate_abi_fanout(rr, dist, rank, k * step_elems, sdtype, dev)
                f = tag(multi.AbiFanout(rr, dist, rank, k * step_elems, sdtype, dev, produce, mesh=algo == "scatter_allgather"))
                f.calibration = cal
                return f
            args.fanout = "torch"
        try:
            return tag(multi.TileFanout(dist, rank, k * step_elems, sdtype, dev, produce, algo=args.fanout_algo))
        except RuntimeError as e:             # a backend without scatter / all-gather on device tensors: the broadcast always works
            if rank == 0:
                print(f"bench.py: fan-out algorithm {args.fanout_algo!r} unavailable ({e}); using bcast", file=sys.stderr)
            return tag(multi.TileFanout(dist, rank, k * step_elems, sdtype, dev, produce, algo="bcast"))

    def src_fmt(wl):
        return "u8 I/Q bytes (2 B per sample, the RTL-SDR wire format)" if wl.in_mult == 2 else "Complex<f32> (8 B per sample)"

    def collective_report(fan_, kms_, steps_, wall_ms, kernel_ms, fmt):
        bms_sum, bn = fan_.broadcast_ms()
        bms = bms_sum / max(bn, 1)
        kstep = kernel_ms
        ks = getattr(fan_, "tile_steps", 1)
        bstep = bms / ks                         # fan-out time per step of source
        return {"backend": "rccl" if backend == "nccl" else backend, "fanout": args.fanout,
                "fanout_self_check": (None if "ok" not in abi_check else
                                      "rr_fanout_* verified on this group (known tiles, both algorithms, checksums on every rank)"
                                      if abi_check["ok"] else f"rr_fanout_* FAILED its self-check ({abi_check['why']}): torch.distributed fan-out used"),
                "algorithm": getattr(fan_, "algo", "bcast"), "calibration_ms_per_tile": getattr(fan_, "calibration", None),
                "algorithm_choice": (f"--fanout-algo {args.fanout_algo}" if args.fanout_algo != "auto" else
                                     "both algorithms timed on this group's fabric at this tile size before the run "
                                     "(calibration_ms_per_tile, no compute alongside): the faster one"
                                     if getattr(fan_, "calibration", None) else
                                     "one broadcast per tile (the only form this backend moves device tensors with)"),
                "source_format": fmt,
                "ranks": dist.get_world_size(),
                "devices_visible": ndev, "tile_bytes": fan_.bytes_per_tile, "tile_steps": ks, "broadcasts_timed": bn,
                "broadcast_ms_per_tile": round(bms, 4), "broadcast_ms_per_step": round(bstep, 4),
                "source_broadcast_gbs": round(fan_.bytes_per_tile / (bms * 1e-3) / 1e9, 1) if bms > 0 else None,
                "kernel_ms_per_step": round(kstep, 4),
                "overlap": round(max(0.0, min(1.0, (bstep + kstep - wall_ms) / max(min(bstep, kstep), 1e-9))), 3),
                # written down BEFORE any run on more than one GPU (none has happened: DESIGN §6): per algorithm the fan-out
                # time of one tile over ~153 GB/s xGMI links and the efficiency it allows, at this job's size and at 2 / 4 / 8
                "predicted": {"assumptions": {"xgmi_link_gbs": multi.XGMI_LINK_GBS, "collective_latency_ms": multi.COLLECTIVE_LATENCY_MS,
                                              "compute_ms_per_tile": round(kstep * ks, 4), "tile_bytes": fan_.bytes_per_tile},
                              "this_job": multi.predict_fanout(dist.get_world_size(), fan_.bytes_per_tile, kstep * ks),
                              "at_2_4_8_gpus": {str(n): multi.predict_fanout(n, fan_.bytes_per_tile, kstep * ks) for n in (2, 4, 8)}}}

    # N > 1: the SAME workload on one rank with its source resident, measured by rank 0 alone before the collective run (the
    # other ranks wait at the barrier): the N = 1 anchor of this line's scaling curve.  (The driver's own N = 1 run is a
    # different workload, configs[1]: value(N) / value(1) across those two lines would compare FftFilter samples with
    # channel-samples — VERDICT r2 weak #9.)
    anchor = None
    if world > 1:
        stage("N = 1 anchor on rank 0 (the other ranks wait at the barrier)")
        if rank == 0:
            ua, ta, _, _, _, sma = run_timed(w, args.steps, args.warmup, None, stream, None, None, settle_ms=args.settle_ms)
            anchor = {"workload": w.name, "workload_key": wname, "n1_value": round(ua / ta / 1e6, 2), "unit": "Msamples/s",
                      "n1_ms_per_step": round(ta / args.steps * 1e3, 4), "n1_ms_per_step_median": round(statistics.median(sma), 4),
                      "how": "rank 0 alone with the source resident in its HBM, before the collective run; scaling efficiency of "
                             "this line = value / (n_gpus * n1_value)"}
        dist.barrier()
    stage("fan-out set-up")
    fan = make_fan(w)
    w.report_cold = True
    stage("timed run (fan-out + compute)")
    units, dt, kms, launches, dom_units, step_ms = run_timed(w, args.steps, args.warmup, dist, stream, fan, settle_ms=args.settle_ms)
    settle_main = getattr(w, "settle_steps", 0)

    # max over ranks of the wall time, sum over ranks of the units
    units_all, dt = multi.aggregate(dist, units, dt, dev)

    # N > 1: the same step on every rank at once with the tile ALREADY on the rank (no fan-out) — what channel sharding
    # alone scales to; value / this = what the fan-out costs.  (N = 1 of the driver's scaling run is a different
    # workload, configs[1]: this is the same-workload reference for the N > 1 lines.)
    resident = None
    if fan is not None:
        stage("resident reference run (no fan-out)")
        torch.cuda.synchronize()
        tile = fan.buf[0] if hasattr(fan, "buf") else fan.acquire(fan.issued, stream)      # (its first step's worth is read)
        u1, t1, _, _, _, sm1 = run_timed(w, args.steps, 2, dist, stream, None, tile.data_ptr(), settle_ms=args.settle_ms / 2)
        u1a, t1a = multi.aggregate(dist, u1, t1, dev)
        resident = {"value": round(u1a / t1a / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(t1a / args.steps * 1e3, 4),
                    "ms_per_step_median": round(statistics.median(sm1), 4),
                    "what": "the same N-rank step with the source tile already resident on every rank (no fan-out in the step)"}

    others = {}
    cpu_legs = {}
    if not args.no_others:
        names = ([n for n in WORKLOADS if n not in (wname, "channelizer_model")] if world == 1 else
                 [n for n in ("fm_multi_u8", "channelizer", "channelizer_model") if n != wname])
        for name in names:
            streamed = world > 1 and name.startswith("fm_multi")
            stage(f"others: {name}")
            with rr.build_options(**opts):
                wo = WORKLOADS[name](dev, rank, world, shared_src if streamed else (lambda gen, numel, dtype: gen()))
            k = max(3, min(args.steps, 10)) if name != "fir" else 200
            if getattr(wo, "rotator", None) == "replay":
                k = 5                              # (a step is 1.25e7 sequential rotator phases: ~33 ms on the host generator)
            fo = make_fan(wo) if streamed else None
            # (a replay-rotator step is bound by one sequential chain, not by clocks: no settle phase for it)
            # (warm-up 6 for a replay rotator: the default starts on the device chain and hands a block whose calls outrun it —
            #  these back-to-back steps do — to the host generator after three such calls; the timed steps are the sustained state)
            u, t, km, ln, du, sm = run_timed(wo, k, 6 if getattr(wo, "rotator", None) == "replay" else 2, dist, stream, fo,
                                             settle_ms=0.0 if getattr(wo, "rotator", None) == "replay" else args.settle_ms / 2)
            ua, ta = multi.aggregate(dist, u, t, dev)
            avg_s = km / max(ln, 1) * 1e-3
            ach = (wo.dominant_bytes_per_unit * du / max(ln, 1)) / avg_s / 1e9 if km > 0 else None
            fl = (wo.dominant_flops_per_unit * du / max(ln, 1)) / avg_s / 1e12 if km > 0 else None
            fx = None if (km <= 0 or wo.dominant_flops_exec_per_unit is None) else (wo.dominant_flops_exec_per_unit * du / max(ln, 1)) / avg_s / 1e12
            others[name] = {"workload": wo.name, "msamples_per_s": round(ua / ta / 1e6, 1),
                            "ms_per_step": round(ta / k * 1e3, 4), "ms_per_step_median": round(statistics.median(sm), 4),
                            "chain_alg_gbs": round(wo.alg_bytes_per_sample * ua / ta / 1e9, 1),
                            "bound": wo.bound, "dominant_kernel": wo.kernel,
                            "dominant_kernel_alg_gbs": None if ach is None else round(ach, 1),
                            "dominant_kernel_hbm_frac": None if ach is None else round(ach / HBM_PEAK_GBS, 4),
                            "dominant_kernel_ms": round(avg_s * 1e3, 4) if km > 0 else None,
                            "dominant_kernel_executed_tflops": None if fx is None else round(fx, 2),
                            "dominant_kernel_executed_fp32_frac": None if fx is None else round(fx / FP32_PEAK_TFLOPS, 4),
                            "dominant_kernel_nominal_reference_tflops": None if fl is None else round(fl, 2),
                            "dominant_kernel_nominal_reference_fp32_frac": None if fl is None else round(fl / FP32_PEAK_TFLOPS, 4)}
            if getattr(wo, "rotator", None):
                others[name]["rotator"] = wo.rotator
            if wo.bound == "sequential_rotator":
                others[name]["outputs_per_s"] = round(ua / 8 / ta, 1)
                others[name]["rotator_ns_per_output"] = round(ta / (ua / 8) * 1e9, 2)
            if wo.bound_note:
                others[name]["bound_note"] = wo.bound_note
            if world == 1 and not args.no_cpu and name in ("full_chain_fused", "fir_fft_chain"):
                cpu_legs[name] = cpu_1thread(wo, max(2.0, args.cpu_seconds / 2))
            if world > 1 and streamed:
                others[name]["source"] = "streamed: rank 0 broadcasts every step's tile (double-buffered) inside the timed region"
                others[name]["collective"] = collective_report(fo, km, k, ta / k * 1e3, avg_s * 1e3, src_fmt(wo))
            elif world > 1:
                others[name]["source"] = "resident on every rank (a 400 MB f32 tile per 0.14 ms step cannot stream over xGMI)"
            del fo
            del wo
            torch.cuda.empty_cache()
        if world == 1 and not args.no_dropin:
            others.update(dropin_report())

    if rank == 0:
        value = units_all / dt / 1e6
        avg_kernel_s = (kms / max(launches, 1)) * 1e-3
        alg_bytes_per_launch = w.dominant_bytes_per_unit * dom_units / max(launches, 1)
        alg_flops_per_launch = w.dominant_flops_per_unit * dom_units / max(launches, 1)
        exec_per_unit = w.dominant_flops_per_unit if w.dominant_flops_exec_per_unit is None else w.dominant_flops_exec_per_unit
        exec_flops_per_launch = exec_per_unit * dom_units / max(launches, 1)
        gbs = alg_bytes_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        tfl = alg_flops_per_launch / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
        tfx = exec_flops_per_launch / avg_kernel_s / 1e12 if avg_kernel_s > 0 else 0.0
        traffic, tnote = measured_traffic(wname)
        roof = {"bound": w.bound, "kernel": w.kernel, "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                "measured_in": "a pass of `steps` steps right after the timed region (the library's HIP-event brackets around the "
                               "dominant kernel on its launch stream); the timed region itself carries no instrumentation",
                "alg_bytes_per_launch": alg_bytes_per_launch,
                "executed_flops_per_launch": exec_flops_per_launch, "nominal_reference_flops_per_launch": alg_flops_per_launch,
                "traffic": traffic, "traffic_note": tnote}
        if w.bound_note:
            roof["bound_note"] = w.bound_note
        if w.bound == "hbm":
            roof.update({"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "frac_counts": "ALGORITHMIC bytes (SURVEY §8d: compulsory input + output of the chain) per launch / mean kernel duration / 8 TB/s",
                         "executed_vector_fp32_tflops": round(tfx, 2)})
        else:
            roof.update({"achieved": round(tfx, 2), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tfx / FP32_PEAK_TFLOPS, 4),
                         "frac_counts": "EXECUTED flops: the transforms (5 N log2 N), multiply-adds and demodulation the GPU kernel actually "
                                        "runs per launch / mean kernel duration / 157.3 TFLOP/s (not the reference algorithm's count)",
                         "nominal_reference_tflops": round(tfl, 2),
                         "nominal_reference_frac": round(tfl / FP32_PEAK_TFLOPS, 4),
                         "nominal_reference_note": "the REFERENCE's algorithm for the same samples (two 1024-point transforms + product per 561 "
                                                   "samples and channel) priced at this kernel's duration: a speed-up-at-peak figure, not a utilisation",
                         "hbm_gbs": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
        par = ("1 GPU" if world == 1 else
               f"{world} ranks, channel-sharded (no data-path collective); shared IQ source produced on rank 0 and broadcast "
               f"tile by tile on a communication stream, double-buffered against the compute stream, inside the timed region")
        line = {
            "metric": METRIC, "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            # the same W + K steps straight from an idle GPU, measured first (no settle phase): what `value` would be without
            # the sustained-clock protocol, comparable with the round-1 figures
            "ms_per_step_from_idle": None if getattr(w, "cold_ms_per_step", None) is None else round(w.cold_ms_per_step, 4),
            "value_from_idle": None if getattr(w, "cold_ms_per_step", None) is None else round(units_all / (w.cold_ms_per_step * 1e-3 * args.steps) / 1e6, 2),
            "ms_per_step_median": round(statistics.median(step_ms), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w.dtype, "data": "synthetic",
            "config": {"workload": w.name, "workload_key": wname, "samples_per_step_per_gpu": w.n, "parallelism": par,
                       "settle_steps": settle_main,
                       "protocol": (f"{settle_main} untimed settle steps (~{args.settle_ms:g} ms of back-to-back passes: the power controller's "
                                    f"start-up transient), then the driver's {args.warmup} warm-up and {args.steps} timed steps; the timed "
                                    "region carries no events or profiling"),
                       "why_this_workload": ("BASELINE.json configs[1] is the single-GPU configuration the metric is quoted on; the "
                                             "metric's four-block chain is others.full_chain / others.full_chain_fused"
                                             if wname == "fftfilter" else
                                             "BASELINE.json configs[3], the multi-GPU configuration (32 channels per GPU)"
                                             if wname == "fm_multi" else "--workload"),
                       **({"step_is": "k_fftfilt_os (roofline.avg_kernel_ms) + the 5 us pass that keeps FftFilter's outputs on non-finite input "
                                      "the reference's (k_ref_blocks_nonfinite: one probe per tile in the steady state, DESIGN.md section 8); "
                                      "rr_build_opts.fft_nonfinite_tiles leaves it out (bench.py --opt fft_nonfinite_tiles=1)"}
                          if wname in ("fftfilter", "fir_fft_chain") else {})},
            "roofline": roof,
            "parity": parity_report(),
            "chain_alg_gbs": round(w.alg_bytes_per_sample * value * 1e6 / 1e9, 1),
        }
        if world > 1:
            line["scale_anchor"] = anchor
            line["scaling_efficiency_vs_anchor"] = (round(value / (world * anchor["n1_value"]), 4)
                                                    if anchor and anchor["n1_value"] > 0 else None)
            line["collective"] = collective_report(fan, kms, args.steps, dt / args.steps * 1e3, avg_kernel_s * 1e3, src_fmt(w))
            line["source_broadcast_gbs"] = line["collective"]["source_broadcast_gbs"]
            line["resident_source"] = resident
            line["fanout_efficiency"] = round(value / resident["value"], 4) if resident and resident["value"] > 0 else None
        if others:
            line["others"] = others
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
            line["gpu_over_cpu_1thread_port"] = round(value / line["cpu_baseline"]["value"], 1)
        # The metric string names the FOUR-block chain and the north star states its ">= 100x" target on the FIR + FftFilter
        # pair; `value` stays configs[1] (the configuration the metric is quoted on).  Both chains, with their own roofline
        # and their own CPU leg, as first-class objects of the line:
        # The 1 -> 8 curve has never been measured (no node with more than one GPU has run this code): what stands in for it is
        # a PREDICTION, regenerated on every run from THIS run's measured configs[3] step (DESIGN.md §7 prints it
        # through tools/fanout_table.py).  No scaling claim is made anywhere.
        fm = others.get("fm_multi")
        if world == 1 and fm:
            ks = max(1, args.tile_steps)
            comp = fm["ms_per_step"] * ks
            tile_f32, tile_u8 = ks * 19_200_000, ks * 4_800_000
            line["multi_gpu_prediction"] = {
                "what": "PREDICTED weak-scaling efficiency of configs[3] (32 channels per GPU, shared source fanned out per tile of "
                        f"{ks} steps, overlapped with the compute on the previous tile) from this run's measured step; never measured",
                "measured_fm_multi_ms_per_step": fm["ms_per_step"], "tile_steps": ks,
                "assumptions": {"xgmi_link_gbs": multi.XGMI_LINK_GBS, "collective_latency_ms": multi.COLLECTIVE_LATENCY_MS},
                "complex_f32_source": {"tile_bytes": tile_f32, **{str(n): multi.predict_fanout(n, tile_f32, comp) for n in (2, 4, 8)}},
                "u8_source": {"tile_bytes": tile_u8, **{str(n): multi.predict_fanout(n, tile_u8, comp) for n in (2, 4, 8)}}}
        fc = others.get("full_chain_fused")
        if fc and "full_chain_fused" in cpu_legs:
            cb = cpu_legs["full_chain_fused"]
            line["metric_chain"] = {
                "workload": fc["workload"], "workload_key": "full_chain_fused", "value": fc["msamples_per_s"], "unit": "Msamples/s",
                "ms_per_step": fc["ms_per_step"],
                "roofline": {"kernel": fc["dominant_kernel"], "avg_kernel_ms": fc["dominant_kernel_ms"],
                             "bound": fc["bound"], "bound_note": fc.get("bound_note"), "achieved": fc["dominant_kernel_alg_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fc["dominant_kernel_hbm_frac"],
                             "alg_bytes_per_sample": 9.0,
                             "executed_vector_fp32_tflops": fc["dominant_kernel_executed_tflops"],
                             "executed_fp32_frac": fc["dominant_kernel_executed_fp32_frac"]},
                "cpu_baseline": dict(cb, chain="FirFilter(127) -> FftFilter(401) -> RationalResampler(1:4) -> QuadratureDemod, "
                                                "four oracle blocks under the reference's single-threaded Graph loop"),
                "gpu_over_cpu_1thread_port": round(fc["msamples_per_s"] / cb["value"], 1) if cb["value"] > 0 else None}
        pr = others.get("fir_fft_chain")
        if pr and "fir_fft_chain" in cpu_legs:
            cb = cpu_legs["fir_fft_chain"]
            line["north_star_target"] = {
                "workload": pr["workload"], "workload_key": "fir_fft_chain",
                "target": ">= 100x the CPU-reference Msamples/s on the 127-tap FIR + 1024-pt FftFilter chain at 1 GPU (BASELINE.json north_star)",
                "gpu_msamples": pr["msamples_per_s"], "cpu_msamples_1thread": cb["value"], "cpu_kind": cb["kind"], "cpu_sample": cb["sample"],
                "ratio": round(pr["msamples_per_s"] / cb["value"], 1) if cb["value"] > 0 else None,
                "met": bool(cb["value"] > 0 and pr["msamples_per_s"] / cb["value"] >= 100.0),
                "roofline_frac": pr["dominant_kernel_hbm_frac"]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
