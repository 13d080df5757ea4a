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

Multi-GPU (one process per GPU, weak scaling): the path shards by channel — every rank
filters its own channel of a shared IQ source (channel c uses the low-pass taps shifted to
f_c, i.e. complex taps, same kernel).  The only collective is the fan-out broadcast of the
source from rank 0 over RCCL, done before the timed region (inputs resident in HBM).

Prints ONE JSON line (rank 0).  `value` counts input samples entering the first block,
summed over ranks, per second of max-over-ranks wall time.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import rustradio_amd as rr  # noqa: E402
from rustradio_amd import multi  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
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

    def step(self, stream):
        """one pass over the resident batch; returns input samples consumed by the first block"""
        n_in = self.n * self.in_mult
        for i, b in enumerate(self.blocks):
            es_out = b.out_dtype.itemsize
            cap = self.caps[i]
            st, c, p, need = b.work_dev(self.bufs[i].data_ptr(), n_in, self.bufs[i + 1].data_ptr(), cap, stream)
            if i == 0:
                c //= self.in_mult
                consumed0 = c
            if i == self.dominant:
                self.dom_units += c
            n_in = p
        return consumed0


def chan_taps(taps, fs, f_c):
    """channel c of a shared source: the low-pass prototype shifted to f_c (complex band-pass)."""
    if f_c == 0.0:
        return taps
    k = np.arange(len(taps), dtype=np.float64)
    return (taps.astype(np.complex128) * np.exp(2j * np.pi * f_c * k / fs)).astype(np.complex64)


def make_fftfilter(dev, rank, world, shared_src):
    w = Workload()
    w.name = "FftFilter 401 taps (ref fft_size 1024, nsamples 623), 10 Msps Complex<f32>, 100,000,000 samples/step"
    w.dtype = "f32"
    fs, n = 10e6, 100_000_000
    taps = rr.low_pass_complex(fs, 1e6, 60e3)
    assert len(taps) == 401
    f_c = 0.0 if world == 1 else multi.channel_frequency(rank, world, 250e3)
    blk = rr.FftFilter(chan_taps(taps, fs, f_c))
    w.blocks = [blk]
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev)),
              torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
    w.caps = [n + 1024]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    w.fs = fs
    w.cpu = ("FftFilter", taps)
    return w


def make_fir(dev, rank, world, shared_src):
    w = Workload()
    w.name = "FirFilter<Complex> 127 real taps, 1,000,000 samples/step"
    fs, n = 10e6, 1_000_000
    taps = rr.low_pass_complex(fs, 1e6, 190e3)
    assert len(taps) == 127
    w.blocks = [rr.FirFilter(taps)]
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0001, dev)),
              torch.empty(2 * n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    w.cpu = ("FirFilter", taps)
    return w


def make_fir_1e8(dev, rank, world, shared_src):
    """configs[0]'s filter at a steady-state size (1e6 samples is one launch of ~15 us: launch-bound)"""
    w = Workload()
    w.name = "FirFilter<Complex> 127 real taps, 100,000,000 samples/step (overlap-save tiles)"
    fs, n = 10e6, 100_000_000
    taps = rr.low_pass_complex(fs, 1e6, 190e3)
    w.blocks = [rr.FirFilter(taps)]
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0001, dev)),
              torch.empty(2 * n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 0, 16.0
    w.cpu = ("FirFilter", taps)
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
    w.bufs = [shared_src(lambda: synth_real(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0006, dev)),
              torch.empty(n, dtype=torch.float32, device=dev)]
    w.caps = [n]
    w.alg_bytes_per_sample = 8.0
    w.dominant, w.dominant_bytes_per_unit = 0, 8.0
    w.cpu = ("FirFilterFloat", taps)
    return w


def make_fir_fft_chain(dev, rank, world, shared_src):
    """the north star's ">= 100x the CPU reference" pair: 127-tap FirFilter -> FftFilter(401 taps, ref 1024-pt)
    on the configs[1] input, device-resident intermediate"""
    w = Workload()
    w.name = "FirFilter<Complex>(127 real taps) -> FftFilter(401 taps, ref fft_size 1024), 10 Msps Complex<f32>, 100,000,000 samples/step"
    fs, n = 10e6, 100_000_000
    t1 = rr.low_pass_complex(fs, 1e6, 190e3)
    t2 = rr.low_pass_complex(fs, 1e6, 60e3)
    assert len(t1) == 127 and len(t2) == 401
    w.blocks = [rr.FirFilter(t1), rr.FftFilter(t2)]
    w.n = n
    w.bufs = [shared_src(lambda: synth_complex(n, fs, (0.3e6, 1.2e6, 3.7e6), 0x5EED0002, dev)),
              torch.empty(2 * n, dtype=torch.float32, device=dev), torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev)]
    w.caps = [n, n + 1024]
    w.alg_bytes_per_sample = 16.0
    w.dominant, w.dominant_bytes_per_unit = 1, 16.0
    w.cpu = ("fir_fft_chain", (t1, t2))
    return w


def make_fm_chain(dev, rank, world, shared_src, fused=True):
    w = Workload()
    how = "fused into one kernel (rr.FmChain)" if fused else "three blocks, device-resident intermediates"
    w.name = ("FftFilter(463 taps)->RationalResampler(1:6)->QuadratureDemod(exact atan2), 2.4 Msps x 10 s = "
              "24,000,000 samples/step, " + how)
    fs, n = 2.4e6, 24_000_000
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    assert len(taps) == 463
    src = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0003))
    if fused:
        w.blocks = [rr.FmChain(taps, 1, 6, 1.0, rr.ATAN2_EXACT)]
        w.bufs = [src, torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
        w.caps = [n // 6 + 1024]
        w.dominant_bytes_per_unit = 8.0 + 4.0 / 6.0
    else:
        w.blocks = [rr.FftFilter(taps), rr.RationalResampler(1, 6, np.complex64), rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)]
        w.bufs = [src,
                  torch.empty(2 * (n + 1024), dtype=torch.float32, device=dev),
                  torch.empty(2 * (n // 6 + 1024), dtype=torch.float32, device=dev),
                  torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
        w.caps = [n + 1024, n // 6 + 1024, n // 6 + 1024]
        w.dominant_bytes_per_unit = 16.0
    w.n = n
    w.alg_bytes_per_sample = 8.0 + 4.0 / 6.0
    w.dominant = 0
    w.cpu = ("fm_chain", taps)
    return w


def make_rtl_fm_chain(dev, rank, world, shared_src):
    """configs[2] from the RTL-SDR wire format (examples/rtl_fm.rs:328-419): u8 I/Q pairs in, f32 out."""
    w = Workload()
    w.name = ("RtlSdrDecode->FftFilter(463 taps)->RationalResampler(1:6)->QuadratureDemod(exact atan2) fused into one "
              "kernel (rr.FmChainU8), 2.4 Msps x 10 s = 24,000,000 samples/step, u8 I/Q input")
    fs, n = 2.4e6, 24_000_000
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    f32 = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0003))
    src = torch.clamp(torch.round(f32 / 0.008 + 127.0), 0, 255).to(torch.uint8)      # what the dongle delivers
    w.blocks = [rr.FmChainU8(taps, 1, 6, 1.0, rr.ATAN2_EXACT)]
    w.bufs = [src, torch.empty(n // 6 + 1024, dtype=torch.float32, device=dev)]
    w.caps = [n // 6 + 1024]
    w.in_mult = 2
    w.dtype = "u8->f32"
    w.n = n
    w.alg_bytes_per_sample = w.dominant_bytes_per_unit = 2.0 + 4.0 / 6.0
    w.dominant = 0
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
    f32 = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0006))
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
    w.cpu = ("rtl_fm_example", taps)
    return w


def make_fm_chain_unfused(dev, rank, world, shared_src):
    return make_fm_chain(dev, rank, world, shared_src, fused=False)


def make_fm_multi(dev, rank, world, shared_src, per_gpu=32, total=256):
    """BASELINE configs[3]: 256 FM channels of configs[2] on one shared IQ source, 32 per GPU.
    Channel c uses the configs[2] low-pass shifted to f_c = (c - 128) * 8 kHz (complex band-pass);
    rank r owns channels r*32 .. r*32+31 (multi.shard_channels).  `value` counts
    channel-samples: input samples x channels processed."""
    w = Workload()
    fs, n = 2.4e6, 2_400_000
    taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
    nch_total = total if world > 1 else per_gpu
    chans = list(multi.shard_channels(nch_total if world > 1 else per_gpu, world, rank))
    w.name = (f"{len(chans)} FM channels/GPU (FftFilter 463 taps->RationalResampler 1:6->QuadratureDemod, fused) "
              f"on one shared 2.4 Msps IQ source, {n:,} samples/step/channel")
    src = shared_src(lambda: synth_fm(n, fs, dev, 0x5EED0004))
    taps_all = np.stack([chan_taps(taps, fs, multi.channel_frequency(c, total, 8e3)) for c in chans])
    blk = rr.FmMulti(taps_all, 1, 6, 1.0, rr.ATAN2_EXACT)     # one kernel: forward FFT shared by all channels
    w.blocks = [blk]
    w.src, w.n = src, n
    cap = n // 6 + 1024
    w.outs = torch.empty(len(chans) * cap, dtype=torch.float32, device=dev)
    w.alg_bytes_per_sample = 8.0 / len(chans) + 4.0 / 6.0       # shared read: 8/N B in + 0.67 B out per channel-sample
    w.dominant, w.dominant_bytes_per_unit = 0, (8.0 / len(chans) + 4.0 / 6.0) * len(chans)
    w.cpu = ("fm_chain", taps)
    w.bufs = [src]
    nch = len(chans)

    def step(stream):
        st, c, p, need = blk.work_dev(src.data_ptr(), n, w.outs.data_ptr(), cap, stream)
        w.dom_units += c
        return c * nch
    w.step = step
    return w


def make_channelizer(dev, rank, world, shared_src, fused=True):
    w = Workload()
    how = ("fused into one composite decimating FIR (rr.HilbertFir)" if fused
           else "two blocks, device-resident analytic stream")
    w.name = ("Hilbert(65)->FirFilter<Complex>(255 real taps, deci 8), 100 Msps f32 x 1 s = 100,000,000 samples/step, "
              + how)
    fs, n = 100e6, 100_000_000
    taps = rr.low_pass_complex(fs, 5e6, 943e3)
    assert len(taps) == 255
    src = shared_src(lambda: synth_real(n, fs, (3e6, 12e6, 37e6), 0x5EED0005, dev))
    w.n = n
    w.alg_bytes_per_sample = 5.0
    if fused:
        w.blocks = [rr.HilbertFir(65, taps, 8)]
        w.bufs = [src, torch.empty(2 * (n // 8 + 8), dtype=torch.float32, device=dev)]
        w.caps = [n // 8 + 8]
        w.dominant, w.dominant_bytes_per_unit = 0, 5.0
    else:
        w.blocks = [rr.Hilbert(65), rr.FirFilter(taps, deci=8)]
        w.bufs = [src, torch.empty(2 * n, dtype=torch.float32, device=dev),
                  torch.empty(2 * (n // 8 + 8), dtype=torch.float32, device=dev)]
        w.caps = [n, n // 8 + 8]
        w.dominant, w.dominant_bytes_per_unit = 0, 12.0
    w.cpu = ("channelizer", taps)
    return w


def make_channelizer_unfused(dev, rank, world, shared_src):
    return make_channelizer(dev, rank, world, shared_src, fused=False)


WORKLOADS = {"fftfilter": make_fftfilter, "fir": make_fir, "fm_chain": make_fm_chain,
             "fm_chain_unfused": make_fm_chain_unfused, "fm_multi": make_fm_multi, "channelizer": make_channelizer,
             "rtl_fm_chain": make_rtl_fm_chain, "channelizer_unfused": make_channelizer_unfused,
             "fir_fft_chain": make_fir_fft_chain, "rtl_fm_example": make_rtl_fm_example,
             "fir_1e8": make_fir_1e8, "fir_float": make_fir_float}


# ---- measurement ------------------------------------------------------------------------------
def run_timed(w, steps, warmup, dist, stream):
    for b in w.blocks:
        b.set_profiling(False)
    w.dom_units = 0
    for _ in range(warmup):
        w.step(stream.cuda_stream)
    torch.cuda.synchronize()
    w.blocks[w.dominant].set_profiling(True)
    w.dom_units = 0
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    units = 0
    for _ in range(steps):
        units += w.step(stream.cuda_stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, launches = w.blocks[w.dominant].profile(reset=True)
    w.blocks[w.dominant].set_profiling(False)
    return units, dt, kms, launches, w.dom_units


def cpu_baseline(w, seconds=10.0):
    """The oracle (strict-order C restatement of the reference blocks, oracle/rr_oracle.c)
    timed on ONE host core over 512,000-sample work() windows (src/stream.rs:105) of the
    same synthetic input, for about `seconds` of CPU time."""
    from oracle import pyoracle as orc
    kind, taps = w.cpu
    win = 512_000
    nwin = 16
    if kind == "channelizer":
        host = w.bufs[0][:win * 2 * nwin].cpu().numpy()
        chain = [orc.Hilbert(65), orc.FirFilter(taps, deci=8)]
        win = 1_024_000
    elif kind == "FirFilterFloat":
        host = w.bufs[0][:win * 2 * nwin].cpu().numpy()
        chain = [orc.FirFilter(taps)]
        win = 1_024_000
    elif kind == "fir_fft_chain":
        host = w.bufs[0][:2 * win * nwin].cpu().numpy().view(np.complex64)
        chain = [orc.FirFilter(taps[0]), orc.FftFilter(taps[1])]
    elif kind == "rtl_fm_example":
        win = 4_096_000
        host = w.bufs[0][:win * 4].cpu().numpy()
        chain = [orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(200000, 1024000), orc.QuadratureDemod(1.0)]
    elif kind == "rtl_fm_chain":
        win = 4_096_000                                       # a full u8 ring (src/stream.rs:105)
        host = w.bufs[0][:win * 4].cpu().numpy()
        chain = [orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)]
    else:
        host = w.bufs[0][:2 * win * nwin].cpu().numpy().view(np.complex64)
        chain = {"FftFilter": lambda: [orc.FftFilter(taps)],
                 "FirFilter": lambda: [orc.FirFilter(taps)],
                 "fm_chain": lambda: [orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)]}[kind]()
    nwin = len(host) // win
    rings = [np.zeros(0, b.in_dtype) for b in chain]
    t0 = time.perf_counter()
    fed = 0
    i = 0
    while time.perf_counter() - t0 < seconds:
        chunk = host[(i % nwin) * win:(i % nwin + 1) * win]
        i += 1
        rings[0] = np.concatenate([rings[0], chunk])
        fed += len(chunk) // w.in_mult
        for j, b in enumerate(chain):
            while True:
                st, c, p, need, out = b.work(rings[j], 4_096_000 // b.out_dtype.itemsize)
                rings[j] = rings[j][c:]
                if j + 1 < len(chain):
                    rings[j + 1] = np.concatenate([rings[j + 1], out])
                if st == 1 or (c == 0 and p == 0):      # WAIT_SRC, or no progress (the output is drained every call)
                    break
    dt = time.perf_counter() - t0
    return {"value": round(fed / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": "port",
            "sample": f"{fed} input samples of the same synthetic stream in {win}-sample work() windows, {dt:.1f} s, 1 thread, gcc -O2 strict f32"}


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/pmc_traffic.py); None when not collected."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fftfilter", choices=sorted(WORKLOADS))
    ap.add_argument("--no-others", action="store_true", help="skip the short runs of the other workloads")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (rustradio_amd has no CPU path)")
    ndev = torch.cuda.device_count()
    dev_idx = local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_idx)
    rr.set_device(dev_idx)
    dev = torch.device("cuda", dev_idx)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RR_BENCH_BACKEND", "nccl")     # "gloo": exercise the N>1 path on a 1-GPU box
        if backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group(backend, rank=rank, world_size=world)
        dist = dist_mod

    def shared_src(gen):
        """fan-out of the shared IQ source: rank 0 synthesises it, RCCL broadcasts it."""
        t, gbs = multi.broadcast_source(dist, rank, gen, dev)
        if gbs is not None:
            shared_src.bcast_gbs = gbs
        return t
    shared_src.bcast_gbs = None

    stream = torch.cuda.current_stream()
    w = WORKLOADS[args.workload](dev, rank, world, shared_src)
    units, dt, kms, launches, dom_units = run_timed(w, args.steps, args.warmup, dist, stream)

    # max over ranks of the wall time, sum over ranks of the units
    units_all, dt = multi.aggregate(dist, units, dt, dev)

    others = {}
    if not args.no_others and world == 1:
        for name in WORKLOADS:
            if name == args.workload:
                continue
            wo = WORKLOADS[name](dev, rank, world, lambda gen: gen())
            k = max(3, min(args.steps, 10)) if name != "fir" else 200
            u, t, km, ln, du = run_timed(wo, k, 2, None, stream)
            ach = (wo.dominant_bytes_per_unit * du / max(ln, 1)) / (km / max(ln, 1) * 1e-3) / 1e9 if km > 0 else None
            others[name] = {"workload": wo.name, "msamples_per_s": round(u / t / 1e6, 1),
                            "ms_per_step": round(t / k * 1e3, 4),
                            "chain_alg_gbs": round(wo.alg_bytes_per_sample * u / t / 1e9, 1),
                            "dominant_kernel_alg_gbs": None if ach is None else round(ach, 1)}
            del wo
            torch.cuda.empty_cache()

    if rank == 0:
        value = units_all / dt / 1e6
        avg_kernel_s = (kms / max(launches, 1)) * 1e-3
        alg_bytes_per_launch = w.dominant_bytes_per_unit * dom_units / max(launches, 1)
        achieved = alg_bytes_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0
        line = {
            "metric": METRIC, "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w.dtype, "data": "synthetic",
            "config": {"workload": w.name, "samples_per_step_per_gpu": w.n,
                       "parallelism": f"{world} independent channel(s), one per GPU; shared IQ source broadcast before the timed region"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload),
                         "kernel": "k_fftfilt_real" if args.workload == "fir_float" else "k_fftfilt_os" if args.workload in ("fftfilter", "fm_chain_unfused", "fir_fft_chain", "fir", "fir_1e8") else "k_fm_chain_half" if args.workload in ("fm_chain", "rtl_fm_chain") else "k_fm_chain_split" if args.workload == "rtl_fm_example" else "k_fm_multi_half" if args.workload == "fm_multi" else ("k_fftfilt_prune" if args.workload == "channelizer" else "k_fir" if args.workload != "channelizer_unfused" else "k_hilbert"),
                         "avg_kernel_ms": round(avg_kernel_s * 1e3, 4), "launches": launches,
                         "alg_bytes_per_launch": alg_bytes_per_launch},
            "chain_alg_gbs": round(w.alg_bytes_per_sample * value * 1e6 / 1e9, 1),
        }
        if shared_src.bcast_gbs is not None:
            line["source_broadcast_gbs"] = round(shared_src.bcast_gbs, 1)
        if others:
            line["others"] = others
        if world == 1 and not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
            line["gpu_over_cpu"] = round(value / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
