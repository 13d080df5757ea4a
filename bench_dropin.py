"""bench_dropin.py — the DROP-IN path measurements of bench.py's detail file (never `value`): rr_block_work on
reference-sized 4,096,000-byte HOST windows exactly as the Rust shim calls it (PCIe-inclusive), and the device-resident graph
over reference-sized rr_dstream rings (src/stream.rs:105)."""
from __future__ import annotations

import time

import numpy as np
import torch

import rustradio_amd as rr

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


def devgraph_resident_source(kind, seconds=1.0):
    """a device-RESIDENT source in front of reference-sized 4,096,000-byte HBM rings (rr_dstream): nothing crosses PCIe, so
    what is timed is the blocks' own cost per work() at the window size every unchanged examples/ graph uses (stream.rs:105).
    kind: "fftfilter" (401 taps alone), "fm_chain_3" (configs[2], three blocks), "fm_chain_fused".  The rings are filled
    from the host for the first rounds (valid samples everywhere), then the source only moves the ring's counters.
    -> {"us_per_round", "us_per_call", "calls_per_round", "msamples_per_s", "samples_per_round"}"""
    rng = np.random.default_rng(9)
    x = (rng.uniform(-1, 1, 512_000) + 1j * rng.uniform(-1, 1, 512_000)).astype(np.complex64)
    if kind == "fftfilter":
        blocks = [rr.FftFilter(rr.low_pass_complex(10e6, 1e6, 60e3))]
    else:
        taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
        blocks = ([rr.FmChain(taps, 1, 6, 1.0, rr.ATAN2_EXACT)] if kind == "fm_chain_fused" else
                  [rr.FftFilter(taps), rr.RationalResampler(1, 6, np.complex64), rr.QuadratureDemod(1.0, rr.ATAN2_EXACT)])
    rings = [rr.DeviceStream(blocks[0].in_dtype)] + [rr.DeviceStream(b.out_dtype) for b in blocks]
    launches = rr.lib().rr_debug_kernel_launches
    fed, t0, rounds, calls = 0, None, 0, 0
    while True:
        fed += rings[0].push(x) if rounds < 4 else rings[0].produce_resident()
        for i, b in enumerate(blocks):
            b.work_streams(rings[i], rings[i + 1])
            calls += 1
        rings[-1].discard()
        rounds += 1
        if rounds == 24:
            torch.cuda.synchronize(); t0, fed, calls, r0, l0 = time.perf_counter(), 0, 0, rounds, launches()
        if t0 is not None and rounds % 200 == 0:
            torch.cuda.synchronize()
            if time.perf_counter() - t0 > seconds:
                break
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nr = rounds - r0
    nl = launches() - l0
    # the blocks' own time on the GPU: the library's HIP-event brackets around each block's kernels (a second, short pass)
    for b in blocks:
        b.set_profiling(True)
    for _ in range(50):
        rings[0].produce_resident()
        for i, b in enumerate(blocks):
            b.work_streams(rings[i], rings[i + 1])
        rings[-1].discard()
    torch.cuda.synchronize()
    kus = []
    for b in blocks:
        ms, n = b.profile(reset=True)
        b.set_profiling(False)
        kus.append(round(ms / max(n, 1) * 1e3, 2))
    return {"us_per_round_wall": round(dt / nr * 1e6, 2), "calls_per_round": len(blocks), "us_per_call_wall": round(dt / max(calls, 1) * 1e6, 2),
            "kernel_us_per_call": kus, "kernel_launches_per_round": round(nl / nr, 2),
            "samples_per_round": round(fed / nr), "msamples_per_s": round(fed / dt / 1e6, 1),
            "note": "wall = a Python (ctypes) driver, five calls per round; kernel_us_per_call = HIP events around each block's launches"}


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
    out["devgraph_resident_source"] = {
        "what": "device-RESIDENT source -> blocks over 4,096,000-byte HBM rings (rr_dstream, rr_block_work_streams) -> NullSink: no PCIe, "
                "the blocks' own cost per work() at the reference's window size (512,000 Complex samples), Python driver (ctypes)",
        "fftfilter_401_taps": devgraph_resident_source("fftfilter"),
        "configs2_three_blocks": devgraph_resident_source("fm_chain_3"),
        "configs2_fused": devgraph_resident_source("fm_chain_fused")}
    return out


