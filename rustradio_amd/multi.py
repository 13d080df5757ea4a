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
    broadcasts it (RCCL over xGMI on the GPU box, gloo in the CPU tests) on a COMMUNICATION stream while every
    rank computes on tile t from the other half on its compute stream.  Two events per half order the streams:
    `ready[h]` (broadcast of the tile in half h finished -> compute may read it) and `free[h]` (compute on
    half h finished -> the next broadcast may overwrite it).  On CPU tensors (tests) the calls are synchronous.

        fan = TileFanout(dist, rank, tile_elems, dtype, device, produce)   # produce(t, out) fills `out` on src_rank
        fan.prefetch(0)
        for t in range(steps):
            fan.prefetch(t + 1)                 # broadcast of tile t+1 overlaps ...
            x = fan.acquire(t, compute_stream)  # ... the compute on tile t
            ... launch work on compute_stream reading x ...
            fan.release(t, compute_stream)
    """

    def __init__(self, dist, rank, tile_elems, dtype, device, produce, src_rank=0):
        self.dist, self.rank, self.src, self.produce = dist, rank, src_rank, produce
        self.device = device
        self.cuda = device.type == "cuda"
        self.buf = [torch.empty(tile_elems, dtype=dtype, device=device) for _ in range(2)]
        self.bytes_per_tile = tile_elems * self.buf[0].element_size()
        self.issued = -1
        self.n_bcast = 0
        if self.cuda:
            self.comm = torch.cuda.Stream(device=device)
            self.ready = [torch.cuda.Event() for _ in range(2)]
            self.free = [torch.cuda.Event() for _ in range(2)]
            self.t_beg = []
            self.t_end = []
            self.used = [False, False]

    def prefetch(self, t):
        """enqueue production + broadcast of tile t (idempotent)"""
        if t <= self.issued:
            return
        assert t == self.issued + 1
        self.issued = t
        h = t % 2
        if not self.cuda:
            if self.rank == self.src:
                self.produce(t, self.buf[h])
            if self.dist is not None:
                self.dist.broadcast(self.buf[h], src=self.src)
            self.n_bcast += 1
            return
        with torch.cuda.stream(self.comm):
            if self.used[h]:
                self.comm.wait_event(self.free[h])          # compute on the previous occupant of this half is done
            if self.rank == self.src:
                self.produce(t, self.buf[h])
            if self.dist is not None:
                b, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                b.record(self.comm)
                w = self.dist.broadcast(self.buf[h], src=self.src, async_op=True)
                w.wait()                                     # the comm stream (not the host) waits for the collective
                e.record(self.comm)
                self.t_beg.append(b)
                self.t_end.append(e)
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
        """-> (summed broadcast time in ms on the comm stream, broadcasts timed); call after a device sync"""
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

    def __init__(self, rr, dist, rank, tile_elems, dtype, device, produce, src_rank=0, rccl_always=False):
        self.rank, self.src, self.produce, self.dtype, self.device = rank, src_rank, produce, dtype, device
        self.bytes_per_tile = tile_elems * torch.empty(0, dtype=dtype).element_size()
        world = dist.get_world_size() if dist is not None else 1
        gid = [rr.fanout_unique_id() if (rank == src_rank and (world > 1 or rccl_always)) else None]
        if dist is not None and world > 1:
            dist.broadcast_object_list(gid, src=src_rank)
        flags = rr.FANOUT_TIMING | (rr.FANOUT_RCCL_ALWAYS if rccl_always else 0)
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
