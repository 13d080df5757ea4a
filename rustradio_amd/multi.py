"""Multi-GPU host logic (one process per GPU, torch.distributed; backend "nccl" = RCCL on the
GPU box, "gloo" in the CPU tests).

The hot path shards by channel: independent chains share nothing but the IQ source
(the reference fans a source out with a `Tee` tree, src/tee.rs:10-24).  So there is exactly
one collective — the fan-out broadcast of the shared source from the rank that owns it —
and no reduction on the data path; timing is reduced as max-over-ranks, work as a sum.
"""
from __future__ import annotations

import time

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
