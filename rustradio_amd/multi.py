"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" = RCCL on the
GPU box, "gloo" in the CPU tests).

The hot path shards by channel: independent chains share nothing but the IQ source
(the reference fans a source out with a `Tee` tree, src/tee.rs:10-24).  So there is exactly
one collective — the fan-out broadcast of the shared source from the rank that owns it —
and no reduction on the data path; timing is reduced as max-over-ranks, work as a sum.
"""
from __future__ import annotations

import time

import numpy as np
import torch


def shard_channels(n_channels: int, world: int, rank: int) -> range:
    """Contiguous block of channels owned by `rank` (chain c -> rank c // ceil(n/world));
    every channel is owned exactly once, blocks differ by at most one channel."""
    base, extra = divmod(n_channels, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def channel_frequency(c: int, n_channels: int, spacing_hz: float) -> float:
    """Centre of channel c of a bank of n_channels spaced `spacing_hz` around DC."""
    return (c - n_channels // 2) * spacing_hz


def channel_taps(proto, fs: float, f_c: float):
    """Channel of a shared source: the low-pass prototype shifted to f_c (complex band-pass taps,
    t[k] * exp(2 pi i f_c k / fs), formed in f64 and rounded once)."""
    proto = np.asarray(proto)
    if f_c == 0.0:
        return proto.astype(np.complex64)
    k = np.arange(len(proto), dtype=np.float64)
    return (proto.astype(np.complex128) * np.exp(2j * np.pi * f_c * k / fs)).astype(np.complex64)


CFG4_CHANNELS, CFG4_SPACING_HZ, CFG4_FS = 256, 8e3, 2.4e6


def cfg4_taps(proto, chans, total: int = CFG4_CHANNELS, spacing_hz: float = CFG4_SPACING_HZ, fs: float = CFG4_FS):
    """BASELINE configs[3]: taps of channels `chans` of the 256-channel bank, chain c =
    the configs[2] low-pass shifted to f_c = (c - 128) * 8 kHz -> [len(chans)][ntaps] complex64."""
    return np.stack([channel_taps(proto, fs, channel_frequency(c, total, spacing_hz)) for c in chans])


def cfg5_translate_hz(rank: int, world: int, fs: float = 100e6) -> float:
    """BASELINE configs[4], 8-GPU variant: GPU g runs channel offset f_g through
    FirFilter::translate(fs, f_g) (src/fir.rs:476-486); the analytic band (0, fs/2) cut into `world` slices, 0 for a single GPU."""
    if world <= 1:
        return 0.0
    return (rank + 0.5) * fs / (2.0 * world)       # the analytic signal occupies (0, fs/2): one slice per GPU


XGMI_LINK_GBS = 153.0          # /opt/skills/guides/MI355X_MICROARCH.md: 7 point-to-point links x ~153 GB/s per GPU
COLLECTIVE_LATENCY_MS = 0.02   # launch + protocol latency of one RCCL collective on a communication stream (assumed)


def predict_fanout(world: int, tile_bytes: int, compute_ms_per_tile: float):
    """PREDICTED cost of fanning one tile of the shared source out to `world` GPUs of one xGMI mesh, per algorithm, and
    the weak-scaling efficiency that follows when the fan-out of tile t+1 overlaps the compute on tile t:
    efficiency = compute / max(compute, fan-out).  Written down BEFORE any multi-GPU run (no node with more than one
    GPU has ever run this code: VERDICT r2 #5) so that the first measured curve can be checked against it.

      bcast              every byte of the tile leaves the owner over ONE link (ring / tree edge): tile / link
      scatter_allgather  the owner's world-1 links carry tile / world each, then every rank gathers world-1 pieces of
                         tile / world over its own links to the other receivers: 2 tile / (world link)
    (two GPUs share one link, so at world = 2 both are tile / link.)"""
    link = XGMI_LINK_GBS * 1e9
    out = {}
    for algo, ncoll in (("bcast", 1), ("scatter_allgather", 2)):
        if world <= 1:
            t = 0.0
        elif algo == "bcast":
            t = tile_bytes / link * 1e3 + ncoll * COLLECTIVE_LATENCY_MS
        else:
            t = 2.0 * tile_bytes / (world * link) * 1e3 + ncoll * COLLECTIVE_LATENCY_MS
        out[algo] = {"fanout_ms_per_tile": round(t, 4),
                     "per_link_gbs_needed_to_hide": None if compute_ms_per_tile <= 0 else round(
                         (tile_bytes if algo == "bcast" else 2.0 * tile_bytes / max(world, 1)) / (compute_ms_per_tile * 1e-3) / 1e9, 1),
                     "efficiency": round(compute_ms_per_tile / max(compute_ms_per_tile, t), 3) if compute_ms_per_tile > 0 else None}
    return out


def broadcast_source(dist, rank: int, make, device, src_rank: int = 0):
    """Fan-out of the shared IQ source.  `make()` builds the float32 tensor on the owning
    rank only; every rank returns an identical tensor on `device`.  -> (tensor, GB/s)."""
    if dist is None:
        return make(), None
    if rank == src_rank:
        t = make()
        meta = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    else:
        t = None
        meta = torch.zeros(1, dtype=torch.int64, device=device)
    dist.broadcast(meta, src=src_rank)
    if t is None:
        t = torch.empty(int(meta.item()), dtype=torch.float32, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(t, src=src_rank)
    if device.type == "cuda":
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return t, t.numel() * 4 / max(dt, 1e-9) / 1e9


def aggregate(dist, units: float, seconds: float, device):
    """Whole-job figures: units summed over ranks, wall time = max over ranks."""
    if dist is None:
        return float(units), float(seconds)
    tt = torch.tensor([seconds], dtype=torch.float64, device=device)
    uu = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dist.all_reduce(uu, op=dist.ReduceOp.SUM)
    return float(uu.item()), float(tt.item())


class TileFanout:
    """Streaming fan-out of the shared IQ source, one tile per step (replaces the reference's
    Tee tree, src/tee.rs:10-24): rank `src_rank` produces tile t+1 into one half of a double buffer and
    fans it out (RCCL over xGMI on the GPU box, gloo in the CPU tests) on a COMMUNICATION stream while every
    rank computes on tile t from the other half on its compute stream.  Two events per half order the streams:
    `ready[h]` (fan-out of the tile in half h finished -> compute may read it) and `free[h]` (compute on
    half h finished -> the next fan-out may overwrite it).  On CPU tensors (tests) the calls are synchronous.

    Two fan-out algorithms, same result:
      "bcast"              one `broadcast` of the tile: every byte leaves the source GPU once per ring / tree edge —
                           bounded by ONE xGMI link (~153 GB/s) however many links the source has;
      "scatter_allgather"  the source scatters 1/world of the tile to each rank, then every rank all-gathers the
                           pieces: the source's 7 links each carry 1/8 of the tile, and the rest moves over the
                           links BETWEEN the receivers, which a broadcast leaves idle.  xGMI is a full mesh, so the
                           tile arrives in ~2/world of the single-link time (plus one more collective's latency);
      "auto"               times both on this job's actual fabric before the run (`calibrate`) and keeps the
                           faster — all ranks decide on the same max-over-ranks numbers.

        fan = TileFanout(dist, rank, tile_elems, dtype, device, produce)   # produce(t, out) fills `out` on src_rank
        fan.prefetch(0)
        for t in range(steps):
            fan.prefetch(t + 1)                 # fan-out of tile t+1 overlaps ...
            x = fan.acquire(t, compute_stream)  # ... the compute on tile t
            ... launch work on compute_stream reading x ...
            fan.release(t, compute_stream)
    """

    ALGOS = ("bcast", "scatter_allgather")
    TIME_EVERY = 4          # every 4th fan-out is timed with HIP events on the communication stream

    def __init__(self, dist, rank, tile_elems, dtype, device, produce, src_rank=0, algo="bcast"):
        self.dist, self.rank, self.src, self.produce = dist, rank, src_rank, produce
        self.device = device
        self.cuda = device.type == "cuda"
        self.world = dist.get_world_size() if dist is not None else 1
        self.n = tile_elems
        self.piece = -(-tile_elems // self.world)                      # elements per rank in the scattered form
        self.full = [torch.empty(self.piece * self.world, dtype=dtype, device=device) for _ in range(2)]
        self.buf = [f[:tile_elems] for f in self.full]                 # what producer and consumers see
        self.part = [torch.empty(self.piece, dtype=dtype, device=device) for _ in range(2)]
        self.bytes_per_tile = tile_elems * self.buf[0].element_size()
        self.issued = -1
        self.n_bcast = 0
        self.calibration = None
        backend = dist.get_backend() if dist is not None else None
        # gloo moves CUDA tensors for broadcast / all_reduce only: the one-GPU fallback of bench.py stays on "bcast"
        self.can_scatter = dist is not None and self.world > 1 and not (self.cuda and backend != "nccl")
        self.into_tensor = backend == "nccl"
        if self.cuda:
            self.comm = torch.cuda.Stream(device=device)
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            self.t_beg = []
            self.t_end = []
            self.used = [False, False]
        if algo == "auto":
            algo = self.calibrate() if self.can_scatter else "bcast"
        if algo not in self.ALGOS:
            raise ValueError(f"TileFanout: unknown algorithm {algo!r}")
        if algo == "scatter_allgather" and not self.can_scatter:
            algo = "bcast"
        self.algo = algo

    # ---- the collective(s) of one tile, on the current stream ------------------------------------------------------
    def _fan(self, h, algo):
        d = self.dist
        if algo == "bcast":
            w = d.broadcast(self.full[h], src=self.src, async_op=True)
            w.wait()                                             # the current (comm) stream waits, not the host
            return
        pieces = list(self.full[h].chunk(self.world)) if self.rank == self.src else None
        w = d.scatter(self.part[h], scatter_list=pieces, src=self.src, async_op=True)
        w.wait()
        if self.into_tensor:
            w = d.all_gather_into_tensor(self.full[h], self.part[h], async_op=True)
        else:
            w = d.all_gather(list(self.full[h].chunk(self.world)), self.part[h], async_op=True)
        w.wait()

    def calibrate(self, tiles=6):
        """time both algorithms on this job's fabric (no compute alongside) -> the faster; identical on every rank"""
        res = {}
        for algo in self.ALGOS:
            for k in range(tiles + 2):                           # two untimed: connection set-up
                if k == 2:
                    if self.cuda:
                        torch.cuda.synchronize()
                    self.dist.barrier()
                    t0 = time.perf_counter()
                if self.cuda:
                    with torch.cuda.stream(self.comm):
                        self._fan(k % 2, algo)
                else:
                    self._fan(k % 2, algo)
            if self.cuda:
                torch.cuda.synchronize()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.device)
            self.dist.all_reduce(dt, op=self.dist.ReduceOp.MAX)
            res[algo] = float(dt.item()) / tiles * 1e3
        self.calibration = {k: round(v, 4) for k, v in res.items()}
        return min(self.ALGOS, key=lambda a: res[a])

    def prefetch(self, t):
        """enqueue production + fan-out of tile t (idempotent)"""
        if t <= self.issued:
            return
        assert t == self.issued + 1
        self.issued = t
        h = t % 2
        if not self.cuda:
            if self.rank == self.src:
                self.produce(t, self.buf[h])
            if self.dist is not None:
                self._fan(h, self.algo)
            self.n_bcast += 1
            return
        with torch.cuda.stream(self.comm):
            if self.used[h]:
                self.comm.wait_event(self.free[h])          # compute on the previous occupant of this half is done
            if self.rank == self.src:
                self.produce(t, self.buf[h])
            if self.dist is not None:
                if t % self.TIME_EVERY == 0:                # a sample of the tiles is timed: two events and their
                    b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # creation cost host time
                    b.record(self.comm)
                    self._fan(h, self.algo)
                    e.record(self.comm)
                    self.t_beg.append(b)
                    self.t_end.append(e)
                else:
                    self._fan(h, self.algo)
            self.ready[h].record(self.comm)
        self.n_bcast += 1

    def acquire(self, t, compute_stream=None):
        self.prefetch(t)
        h = t % 2
        if self.cuda:
            (compute_stream or torch.cuda.current_stream()).wait_event(self.ready[h])
        return self.buf[h]

    def release(self, t, compute_stream=None):
        if self.cuda:
            h = t % 2
            self.free[h].record(compute_stream or torch.cuda.current_stream())
            self.used[h] = True

    def reset_timing(self):
        if self.cuda:
            self.t_beg, self.t_end = [], []

    def broadcast_ms(self):
        """-> (summed fan-out time in ms on the comm stream, fan-outs timed); call after a device sync"""
        if not self.cuda or not self.t_beg:
            return 0.0, 0
        return sum(b.elapsed_time(e) for b, e in zip(self.t_beg, self.t_end)), len(self.t_beg)


class _DevMem:
    """a raw device allocation presented through __cuda_array_interface__, so torch can alias it (no copy)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class AbiFanout:
    """The same fan-out through the C ABI (`rr_fanout_*`, include/rustradio_amd.h — what a Rust graph binds) behind
    TileFanout's interface, so bench.py can run either (`--fanout abi`).  The 128-byte group id travels over the
    torch.distributed group that launched the ranks; the double buffer, the communication stream, the events and the
    RCCL broadcasts are the library's.  `produce(t, out)` fills the torch alias of the half on the owning rank."""

    def __init__(self, rr, dist, rank, tile_elems, dtype, device, produce, src_rank=0, rccl_always=False, mesh=False, timing=True):
        self.rank, self.src, self.produce, self.dtype, self.device = rank, src_rank, produce, dtype, device
        self.bytes_per_tile = tile_elems * torch.empty(0, dtype=dtype).element_size()
        world = dist.get_world_size() if dist is not None else 1
        gid = [rr.fanout_unique_id() if (rank == src_rank and (world > 1 or rccl_always)) else None]
        if dist is not None and world > 1:
            dist.broadcast_object_list(gid, src=src_rank)
        flags = (rr.FANOUT_TIMING if timing else 0) | (rr.FANOUT_RCCL_ALWAYS if rccl_always else 0) | (rr.FANOUT_MESH if mesh else 0)
        self.algo = "scatter_allgather" if mesh else "bcast"
        self.calibration = None
        self.fan = rr.Fanout(gid[0], rank, world, self.bytes_per_tile, src_rank, flags)
        self.pstream = torch.cuda.Stream(device=device)      # the source block's stream on the owning rank
        self.views = {}
        self.issued = -1
        self.n_bcast = 0

    def _view(self, ptr):
        v = self.views.get(ptr)
        if v is None:
            v = torch.as_tensor(_DevMem(ptr, self.bytes_per_tile), device=self.device).view(self.dtype)
            self.views[ptr] = v
        return v

    def prefetch(self, t):
        if t <= self.issued:
            return
        assert t == self.issued + 1
        self.issued = t
        sh = self.pstream.cuda_stream
        if self.rank == self.src:
            out = self._view(self.fan.produce_buf(t, sh))
            with torch.cuda.stream(self.pstream):
                self.produce(t, out)
        self.fan.submit(t, sh)
        self.n_bcast += 1

    def acquire(self, t, compute_stream=None):
        self.prefetch(t)
        return self._view(self.fan.acquire(t, (compute_stream or torch.cuda.current_stream()).cuda_stream))

    def release(self, t, compute_stream=None):
        self.fan.release(t, (compute_stream or torch.cuda.current_stream()).cuda_stream)

    def reset_timing(self):
        self.fan.stats()

    def broadcast_ms(self):
        return self.fan.stats()


def calibrate_abi_fanout(rr, dist, rank, tile_elems, dtype, device, tiles=6):
    """Time both algorithms of `rr_fanout_*` on THIS group's fabric at the job's tile size, no compute alongside (what
    TileFanout.calibrate does for the torch.distributed form) -> (the faster algorithm, {algorithm: ms per tile}); identical
    on every rank (MAX over ranks).  Any failure of one algorithm leaves the other; of both: ("bcast", None)."""
    res = {}
    stream = torch.cuda.current_stream()
    for name, mesh in (("bcast", False), ("scatter_allgather", True)):
        ok = 1
        dt = 0.0
        try:
            fan = AbiFanout(rr, dist, rank, tile_elems, dtype, device, lambda t, out: None, mesh=mesh, timing=False)
            fan.prefetch(0)
            for t in range(tiles + 2):                           # two untimed: connection set-up
                if t == 2:
                    torch.cuda.synchronize()
                    dist.barrier()
                    t0 = time.perf_counter()
                if t + 1 < tiles + 2:
                    fan.prefetch(t + 1)
                fan.acquire(t, stream)
                fan.release(t, stream)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            del fan
        except Exception:                                        # noqa: BLE001 — the other algorithm still counts
            ok = 0
        v = torch.tensor([dt if ok else 1e9, float(ok)], dtype=torch.float64, device=device)
        dist.all_reduce(v[:1], op=dist.ReduceOp.MAX)
        dist.all_reduce(v[1:], op=dist.ReduceOp.MIN)
        if v[1].item() > 0:
            res[name] = float(v[0].item()) / tiles * 1e3
    if not res:
        return "bcast", None
    return min(res, key=lambda a: res[a]), {k: round(x, 4) for k, x in res.items()}


def verify_abi_fanout(rr, dist, rank, device, nbytes=(1 << 20) + 13, ntiles=3):
    """First contact of `rr_fanout_*` with a group of more than one GPU (it has only ever run on one rank: DESIGN §6):
    before a measurement relies on it, fan a few small tiles of known content out with both algorithms and compare a
    checksum of what every rank acquired with the owner's.  -> (ok on EVERY rank, reason).  A mismatch or an error makes
    bench.py fall back to the torch.distributed fan-out and say so in its line; only a hang inside RCCL cannot be caught."""
    ok, why = True, ""
    try:
        idx = torch.arange(nbytes, device=device, dtype=torch.int64)
        stream = torch.cuda.current_stream()
        for mesh in (False, True):
            def produce(t, out, _idx=idx):
                out.copy_(((_idx * (t + 3)) % 251).to(torch.uint8), non_blocking=True)
            fan = AbiFanout(rr, dist, rank, nbytes, torch.uint8, device, produce, mesh=mesh, timing=False)
            fan.prefetch(0)
            for t in range(ntiles):
                if t + 1 < ntiles:
                    fan.prefetch(t + 1)
                x = fan.acquire(t, stream)
                got = int(x.to(torch.int64).sum().item())
                fan.release(t, stream)
                want = int(((idx * (t + 3)) % 251).sum().item())
                if got != want:
                    ok, why = False, f"tile {t} ({'mesh' if mesh else 'bcast'}): checksum {got} != {want} on rank {rank}"
            torch.cuda.synchronize()
            del fan
    except Exception as e:                      # noqa: BLE001 — anything the C ABI reports
        ok, why = False, f"rank {rank}: {e}"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    all_ok = bool(flag.item())
    if not all_ok and not why:
        why = "another rank reported a mismatch"
    return all_ok, why
