#!/usr/bin/env python3
"""bench.py — throughput of the rustradio hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM.  Default workload = BASELINE.json configs[1]:
    FftFilter, 401 taps (low_pass_complex(10e6, 1e6, 60e3) => reference fft_size 1024,
    nsamples 623), 10 Msps synthetic Complex<f32>, 10 s = 100,000,000 samples per step.
Other workloads (--workload; every one also runs short in the default run): bench_workloads.py lists them; the drop-in
path (rr_block_work on reference-sized HOST windows, PCIe-inclusive, never `value`) is bench_dropin.py.

Multi-GPU (`--gpus N`, one process per GPU, weak scaling): when WORLD_SIZE is not set, this process — before it
touches the GPU — starts N ranks of itself with torch.distributed.run and exits with their code.  The path shards
by channel: default N > 1 workload = configs[3] (fm_multi: rank r owns channels 32 r .. 32 r + 31 of the 256-channel
bank).  The only collective is the fan-out of the shared IQ source: rank 0 produces tile t+1 and RCCL-broadcasts it on a
communication stream while every rank runs tile t (double buffer, events) — INSIDE the timed region.

Prints ONE JSON line (rank 0), strict JSON under 4 KB: the contract keys, `config`, `roofline` (dominant kernel: mean launch
duration from HIP events on its launch stream, algorithmic bytes and executed flops per launch, both fractions, `bound` = the
larger), `cpu_baseline`, and compact companions (`metric_chain`, `north_star_target`, `parity`, `verified`, `others_brief`).
Everything else — every workload's full record, the drop-in path, the CPU modes, the multi-GPU prediction, the prose — goes
to gpurun_out/bench_detail.json (--detail-out) and to stderr BEFORE the line.  After each workload's timed passes one more
step on fresh handles is checked against a float64 evaluation of the reference's chain (bench_verify.py); a failed check
makes the exit code 3.  `value` counts input samples entering the first block (x channels for the multi-channel block),
summed over ranks, per second of max-over-ranks wall time.
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
import bench_dropin  # noqa: E402
import bench_verify  # noqa: E402
from bench_workloads import HBM_PEAK_GBS, FP32_PEAK_TFLOPS, METRIC, WORKLOADS, make as make_workload  # noqa: E402


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
            "sample_short": f"{fed} samples of the same stream, {win}-sample work() windows, {dt:.1f} s, 1 thread, gcc -O3 -march=native, strict f32",
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


def _r(x, n=4):
    return None if x is None else round(float(x), n)


def summarize(w, units_all, wall_s, steps, kms, launches, dom_units, step_ms):
    """one workload's figures: rate, the dominant kernel's mean launch duration against BOTH rooflines (algorithmic bytes /
    8 TB/s; executed flops / 157.3 TFLOP/s), `bound` = the larger fraction unless the workload names another bound"""
    avg_s = kms / max(launches, 1) * 1e-3
    per_launch = dom_units / max(launches, 1)
    have = kms > 0 and launches > 0
    gbs = w.dominant_bytes_per_unit * per_launch / avg_s / 1e9 if have else None
    tfx = w.exec_flops_per_unit() * per_launch / avg_s / 1e12 if have else None
    tfl = w.dominant_flops_per_unit * per_launch / avg_s / 1e12 if have else None
    hbm_frac = None if gbs is None else gbs / HBM_PEAK_GBS
    fp_frac = None if tfx is None else tfx / FP32_PEAK_TFLOPS
    bound = w.bound or ("hbm" if (hbm_frac or 0) >= (fp_frac or 0) else "vector_fp32")
    rate = units_all / wall_s / 1e6
    d = {"workload": w.desc or w.name, "msamples_per_s": _r(rate, 1), "ms_per_step": _r(wall_s / steps * 1e3),
         "ms_per_step_median": _r(statistics.median(step_ms)) if step_ms else None,
         "chain_alg_gbs": _r(w.alg_bytes_per_sample * rate * 1e6 / 1e9, 1),
         "bound": bound, "dominant_kernel": w.kernel, "dominant_kernel_ms": _r(avg_s * 1e3) if have else None,
         "launches": launches, "alg_bytes_per_launch": w.dominant_bytes_per_unit * per_launch,
         "executed_flops_per_launch": w.exec_flops_per_unit() * per_launch,
         "nominal_reference_flops_per_launch": w.dominant_flops_per_unit * per_launch,
         "dominant_kernel_alg_gbs": _r(gbs, 1), "dominant_kernel_hbm_frac": _r(hbm_frac),
         "dominant_kernel_executed_tflops": _r(tfx, 2), "dominant_kernel_executed_fp32_frac": _r(fp_frac),
         "dominant_kernel_nominal_reference_tflops": _r(tfl, 2),
         "dominant_kernel_nominal_reference_fp32_frac": None if tfl is None else _r(tfl / FP32_PEAK_TFLOPS)}
    if w.rotator:
        d["rotator"] = w.rotator
    if w.bound == "sequential_rotator":
        d["outputs_per_s"] = _r(units_all / 8 / wall_s, 1)
        d["rotator_ns_per_output"] = _r(wall_s / (units_all / 8) * 1e9, 2)
    if w.bound_note:
        d["bound_note"] = w.bound_note
    return d


def roofline_object(w, s, wname):
    """the line's `roofline` (contract: bound, achieved, peak, unit, frac, traffic) from a summarize() record"""
    traffic, tnote = measured_traffic(wname)
    r = {"bound": s["bound"], "kernel": s["dominant_kernel"],
         "avg_kernel_ms": s["dominant_kernel_ms"], "launches": s["launches"], "alg_bytes_per_launch": s["alg_bytes_per_launch"],
         "traffic": traffic}
    if s["bound"] == "vector_fp32":
        r.update({"achieved": s["dominant_kernel_executed_tflops"], "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": s["dominant_kernel_executed_fp32_frac"]})
    else:
        r.update({"achieved": s["dominant_kernel_alg_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": s["dominant_kernel_hbm_frac"]})
    r.update({"hbm_frac": s["dominant_kernel_hbm_frac"], "executed_fp32_frac": s["dominant_kernel_executed_fp32_frac"],
              "executed_flops_per_launch": s["executed_flops_per_launch"]})
    return r, tnote


LINE_LIMIT = 4000          # bytes: the driver's record of round 5 could not take a 22 KB line


def compact_line(line):
    """strict JSON under LINE_LIMIT bytes: the optional tables go first if it does not fit (they are all in the detail file)"""
    for drop in (None, "others_brief", "verified_worst", "parity", "north_star_target", "metric_chain"):
        if drop:
            line.pop(drop, None)
        s = json.dumps(line, allow_nan=False, separators=(",", ":"))
        if len(s) <= LINE_LIMIT:
            return s
    raise SystemExit(f"bench.py: the result line is {len(s)} bytes, over {LINE_LIMIT}")


def _clean(o):
    """NaN / Inf -> None so that the detail file is strict JSON too"""
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    if isinstance(o, (np.floating, np.integer)):
        return _clean(o.item())
    return o


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
    ap.add_argument("--no-verify", action="store_true", help="skip the f64 check of what was timed (bench_verify.py)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--detail-out", default=os.path.join(ROOT, "gpurun_out", "bench_detail.json"),
                    help="where the full record goes (every workload, the drop-in path, the CPU modes, the multi-GPU prediction, "
                         "prose); the stdout line stays under 4 KB")
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
        w = make_workload(wname, dev, rank, world, shared_src)
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

    def collective_report(fan_, steps_, wall_ms, kernel_ms, fmt):
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
                # written down BEFORE any run on more than one GPU (none has happened: DESIGN §7): per algorithm the fan-out
                # time of one tile over ~153 GB/s xGMI links and the efficiency it allows, at this job's size and at 2 / 4 / 8
                "predicted": {"assumptions": {"xgmi_link_gbs": multi.XGMI_LINK_GBS, "collective_latency_ms": multi.COLLECTIVE_LATENCY_MS,
                                              "compute_ms_per_tile": round(kstep * ks, 4), "tile_bytes": fan_.bytes_per_tile},
                              "this_job": multi.predict_fanout(dist.get_world_size(), fan_.bytes_per_tile, kstep * ks),
                              "at_2_4_8_gpus": {str(n): multi.predict_fanout(n, fan_.bytes_per_tile, kstep * ks) for n in (2, 4, 8)}}}

    def check(wl):
        """bench_verify on this rank's copy of the workload (needs the source resident: rank 0 of an N > 1 job, any rank at N = 1)"""
        if args.no_verify or rank != 0 or (wl.ref is None and not hasattr(wl, "ref_of")):
            return None
        with rr.build_options(**opts):
            try:
                return bench_verify.verify(wl, stream)
            except Exception as e:            # a check that cannot run is a failed check, not a skipped one
                return {"ok": False, "segments": 0, "why": f"{type(e).__name__}: {e}"}

    # N > 1: the SAME workload on one rank with its source resident, measured by rank 0 alone before the collective run (the
    # other ranks wait at the barrier): the N = 1 anchor of this line's scaling curve.  (The driver's own N = 1 run is a
    # different workload, configs[1]: value(N) / value(1) across those two lines would compare FftFilter samples with
    # channel-samples.)
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
    main_sum = summarize(w, units_all, dt, args.steps, kms, launches, dom_units, step_ms)

    # N > 1: the same step on every rank at once with the tile ALREADY on the rank (no fan-out) — what channel sharding
    # alone scales to; value / this = what the fan-out costs.
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
    stage("verify the headline workload")
    verified = {wname: check(w)}

    others = {}
    cpu_legs = {}
    if not args.no_others:
        names = ([n for n in WORKLOADS if n not in (wname, "channelizer_model")] if world == 1 else
                 [n for n in ("fm_multi_u8", "channelizer", "channelizer_model") if n != wname])
        for name in names:
            streamed = world > 1 and name.startswith("fm_multi")
            stage(f"others: {name}")
            with rr.build_options(**opts):
                wo = make_workload(name, dev, rank, world, shared_src if streamed else (lambda gen, numel, dtype: gen()))
            k = max(3, min(args.steps, 10)) if name != "fir" else 200
            replay = wo.rotator == "replay"
            if replay:
                k = 5                              # (a step is 1.25e7 sequential rotator phases: ~33 ms on the host generator)
            fo = make_fan(wo) if streamed else None
            # (a replay-rotator step is bound by one sequential chain, not by clocks: no settle phase for it; warm-up 6: the
            #  default starts on the device chain and hands a block whose calls outrun it — these back-to-back steps do — to the
            #  host generator after three such calls; the timed steps are the sustained state)
            u, t, km, ln, du, sm = run_timed(wo, k, 6 if replay else 2, dist, stream, fo, settle_ms=0.0 if replay else args.settle_ms / 2)
            ua, ta = multi.aggregate(dist, u, t, dev)
            others[name] = summarize(wo, ua, ta, k, km, ln, du, sm)
            if not streamed:
                verified[name] = check(wo)
                others[name]["verified"] = verified[name]
            if world == 1 and not args.no_cpu and name in ("full_chain_fused", "fir_fft_chain"):
                cpu_legs[name] = cpu_1thread(wo, max(2.0, args.cpu_seconds / 2))
            if world > 1 and streamed:
                others[name]["source"] = "streamed: rank 0 broadcasts every step's tile (double-buffered) inside the timed region"
                others[name]["collective"] = collective_report(fo, k, ta / k * 1e3, others[name]["dominant_kernel_ms"] or 0.0, src_fmt(wo))
            elif world > 1:
                others[name]["source"] = "resident on every rank (a 400 MB f32 tile per 0.14 ms step cannot stream over xGMI)"
            del fo
            del wo
            torch.cuda.empty_cache()
        if world == 1 and not args.no_dropin:
            stage("drop-in path (host windows, reference-sized rings)")
            others.update(bench_dropin.dropin_report())

    rc = 0
    if rank == 0:
        value = units_all / dt / 1e6
        roof, tnote = roofline_object(w, main_sum, wname)
        par = "1 GPU" if world == 1 else f"{world} ranks, channel-sharded; shared IQ source fanned out per tile inside the timed region"
        cold = getattr(w, "cold_ms_per_step", None)
        parity_full = parity_report()
        checks = {k: v for k, v in verified.items() if v is not None}
        bad = [k for k, v in checks.items() if not v["ok"]]
        worst = max(checks.items(), key=lambda kv: (kv[1].get("max_err", float("inf")) / kv[1].get("tol", 1.0)) if kv[1]["ok"] else float("inf"),
                    default=(None, None))
        # ---- the full record (detail file + stderr) ----
        detail = {
            "metric": METRIC, "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_step_median": round(statistics.median(step_ms), 4),
            "ms_per_step_from_idle": _r(cold), "value_from_idle": None if cold is None else round(units_all / (cold * 1e-3 * args.steps) / 1e6, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": w.dtype, "data": "synthetic",
            "config": {"workload": w.desc, "workload_key": wname, "samples_per_step_per_gpu": w.n, "parallelism": par,
                       "settle_steps": settle_main,
                       "protocol": (f"{settle_main} untimed settle steps (~{args.settle_ms:g} ms of back-to-back passes: the power controller's "
                                    f"start-up transient), then the driver's {args.warmup} warm-up and {args.steps} timed steps; the timed "
                                    "region carries no events or profiling; a pass with the library's HIP-event brackets around the dominant "
                                    "kernel on its launch stream follows (roofline.avg_kernel_ms), then one with an event pair per step (median)"),
                       "why_this_workload": ("BASELINE.json configs[1] is the single-GPU configuration the metric is quoted on; the "
                                             "metric's four-block chain is metric_chain (others.full_chain_fused / others.full_chain)"
                                             if wname == "fftfilter" else
                                             "BASELINE.json configs[3], the multi-GPU configuration (32 channels per GPU)"
                                             if wname == "fm_multi" else "--workload")},
            "headline": main_sum,
            "roofline": dict(roof, traffic_note=tnote,
                             frac_counts="hbm: ALGORITHMIC bytes (SURVEY §8d: compulsory input + output) per launch / mean kernel duration / "
                                         "8 TB/s; vector_fp32: EXECUTED flops per launch / mean kernel duration / 157.3 TFLOP/s; `bound` = "
                                         "the larger of the two fractions"),
            "parity": parity_full, "verified": checks, "others": others,
        }
        if world > 1:
            detail["scale_anchor"] = anchor
            detail["scaling_efficiency_vs_anchor"] = (round(value / (world * anchor["n1_value"]), 4) if anchor and anchor["n1_value"] > 0 else None)
            detail["collective"] = collective_report(fan, args.steps, dt / args.steps * 1e3, main_sum["dominant_kernel_ms"] or 0.0, src_fmt(w))
            detail["resident_source"] = resident
            detail["fanout_efficiency"] = round(value / resident["value"], 4) if resident and resident["value"] > 0 else None
        if world == 1 and not args.no_cpu:
            detail["cpu_baseline"] = cpu_baseline(w, args.cpu_seconds)
            detail["gpu_over_cpu_1thread_port"] = round(value / detail["cpu_baseline"]["value"], 1)
        # The 1 -> 8 curve has never been measured (no node with more than one GPU has run this code): what stands in for it is
        # a PREDICTION, regenerated on every run from THIS run's measured configs[3] step (tools/fanout_table.py prints it).
        fm = others.get("fm_multi")
        if world == 1 and fm:
            ks = max(1, args.tile_steps)
            comp = fm["ms_per_step"] * ks
            tile_f32, tile_u8 = ks * 19_200_000, ks * 4_800_000
            detail["multi_gpu_prediction"] = {
                "what": "PREDICTED weak-scaling efficiency of configs[3] (32 channels per GPU, shared source fanned out per tile of "
                        f"{ks} steps, overlapped with the compute on the previous tile) from this run's measured step; never measured",
                "measured_fm_multi_ms_per_step": fm["ms_per_step"], "tile_steps": ks,
                "assumptions": {"xgmi_link_gbs": multi.XGMI_LINK_GBS, "collective_latency_ms": multi.COLLECTIVE_LATENCY_MS},
                "complex_f32_source": {"tile_bytes": tile_f32, **{str(n): multi.predict_fanout(n, tile_f32, comp) for n in (2, 4, 8)}},
                "u8_source": {"tile_bytes": tile_u8, **{str(n): multi.predict_fanout(n, tile_u8, comp) for n in (2, 4, 8)}}}
        # The metric string names the FOUR-block chain and the north star states its ">= 100x" target on the FIR + FftFilter
        # pair; `value` stays configs[1] (the configuration the metric is quoted on).  Both chains with their own roofline
        # fractions and their own CPU leg:
        fc = others.get("full_chain_fused")
        if fc and "full_chain_fused" in cpu_legs:
            cb = cpu_legs["full_chain_fused"]
            detail["metric_chain"] = {
                "workload": fc["workload"], "workload_key": "full_chain_fused", "value": fc["msamples_per_s"], "unit": "Msamples/s",
                "ms_per_step": fc["ms_per_step"], "kernel": fc["dominant_kernel"], "avg_kernel_ms": fc["dominant_kernel_ms"],
                "bound": fc["bound"], "hbm_frac": fc["dominant_kernel_hbm_frac"], "executed_fp32_frac": fc["dominant_kernel_executed_fp32_frac"],
                "alg_bytes_per_sample": 9.0, "bound_note": fc.get("bound_note"),
                "cpu_baseline": dict(cb, chain="FirFilter(127) -> FftFilter(401) -> RationalResampler(1:4) -> QuadratureDemod, "
                                                "four oracle blocks under the reference's single-threaded Graph loop"),
                "gpu_over_cpu_1thread_port": round(fc["msamples_per_s"] / cb["value"], 1) if cb["value"] > 0 else None}
        pr = others.get("fir_fft_chain")
        if pr and "fir_fft_chain" in cpu_legs:
            cb = cpu_legs["fir_fft_chain"]
            detail["north_star_target"] = {
                "workload": pr["workload"], "workload_key": "fir_fft_chain",
                "target": ">= 100x the CPU-reference Msamples/s on the 127-tap FIR + 1024-pt FftFilter chain at 1 GPU (BASELINE.json north_star)",
                "gpu_msamples": pr["msamples_per_s"], "cpu_msamples_1thread": cb["value"], "cpu_kind": cb["kind"], "cpu_sample": cb["sample"],
                "ratio": round(pr["msamples_per_s"] / cb["value"], 1) if cb["value"] > 0 else None,
                "met": bool(cb["value"] > 0 and pr["msamples_per_s"] / cb["value"] >= 100.0),
                "hbm_frac": pr["dominant_kernel_hbm_frac"]}
        detail = _clean(detail)
        dtext = json.dumps(detail, allow_nan=False)
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.detail_out)), exist_ok=True)
            with open(args.detail_out, "w") as f:
                f.write(json.dumps(detail, allow_nan=False, indent=1) + "\n")
            detail_where = os.path.relpath(args.detail_out, ROOT)
        except OSError as e:
            detail_where = f"stderr only ({e})"
        print("bench.py detail: " + dtext, file=sys.stderr, flush=True)

        # ---- the line: contract keys + config + roofline + cpu_baseline + compact companions, < 4 KB, strict JSON ----
        line = {
            "metric": METRIC, "value": detail["value"], "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": detail["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": w.dtype, "data": "synthetic",
            "config": {"workload": w.name[:118], "workload_key": wname, "samples_per_step_per_gpu": w.n, "parallelism": par[:118],
                       "settle_steps": settle_main},
            "roofline": _clean(roof),
            "ms_per_step_median": detail["ms_per_step_median"], "value_from_idle": detail["value_from_idle"],
        }
        if "cpu_baseline" in detail:
            cb = detail["cpu_baseline"]
            allc = cb["modes"].get("one_chain_per_core_all_cores", {})
            line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": 1, "kind": cb["kind"], "sample": cb["sample_short"],
                                    "host_cores": cb["host_cores"], "all_cores_value": allc.get("msamples_per_s"),
                                    "all_cores_processes": allc.get("processes")}
        if world > 1:
            c = detail["collective"]
            line["collective"] = {k: c[k] for k in ("backend", "fanout", "algorithm", "ranks", "tile_bytes", "tile_steps", "broadcasts_timed",
                                                    "broadcast_ms_per_tile", "source_broadcast_gbs", "kernel_ms_per_step", "overlap")}
            line["scale_anchor_n1_value"] = anchor["n1_value"] if anchor else None
            line["scaling_efficiency_vs_anchor"] = detail["scaling_efficiency_vs_anchor"]
            line["resident_source_value"] = resident["value"] if resident else None
            line["fanout_efficiency"] = detail["fanout_efficiency"]
        if "metric_chain" in detail:
            m = detail["metric_chain"]
            line["metric_chain"] = {"workload_key": m["workload_key"], "value": m["value"], "ms_per_step": m["ms_per_step"], "kernel": m["kernel"],
                                    "avg_kernel_ms": m["avg_kernel_ms"], "bound": m["bound"], "hbm_frac": m["hbm_frac"],
                                    "executed_fp32_frac": m["executed_fp32_frac"], "cpu_1thread": m["cpu_baseline"]["value"],
                                    "x_cpu_1thread": m["gpu_over_cpu_1thread_port"]}
        if "north_star_target" in detail:
            t = detail["north_star_target"]
            line["north_star_target"] = {k: t[k] for k in ("workload_key", "gpu_msamples", "cpu_msamples_1thread", "ratio", "met", "hbm_frac")}
        aps = parity_full.get("above_plain_share")
        line["parity"] = {"tol": parity_full["tol"], "chain_bound": parity_full["chain_bound"], "used_max": parity_full.get("used_max"),
                          "above_plain_share_max": max(aps.values()) if isinstance(aps, dict) else None}
        line["verified"] = None if args.no_verify else {"ok": not bad, "workloads": len(checks), "segments_each": 16, "failed": bad}
        if worst[0] is not None and not bad:
            line["verified_worst"] = {"workload": worst[0], "max_err": _r(worst[1]["max_err"], 9), "tol": worst[1]["tol"]}
        if others:
            line["others_brief"] = {"cols": ["msamples_per_s", "ms_per_step", "hbm_frac", "executed_fp32_frac"],
                                    **{k: [v["msamples_per_s"], v["ms_per_step"], v["dominant_kernel_hbm_frac"], v["dominant_kernel_executed_fp32_frac"]]
                                       for k, v in others.items() if "msamples_per_s" in v}}
        line["detail"] = detail_where
        sys.stderr.flush()
        print(compact_line(_clean(line)), flush=True)
        if bad:
            print(f"bench.py: the f64 check of what was timed FAILED for {bad}: {[checks[k] for k in bad]}", file=sys.stderr, flush=True)
            rc = 3
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
