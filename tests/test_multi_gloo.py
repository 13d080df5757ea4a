"""CPU, world_size 2, gloo: the N>1 host logic bench.py uses (channel sharding, fan-out
broadcast of the shared IQ source, max-time / sum-units reduction).  The per-rank compute is
the single-GPU path already covered by the -m gpu tests; here it is replaced by a checksum."""
import os
import socket

import torch
import torch.multiprocessing as mp

from rustradio_amd import multi


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    made = []

    def make():
        made.append(rank)
        g = torch.Generator().manual_seed(1234)
        return torch.rand(100_000, generator=g, dtype=torch.float32)

    src, gbs = multi.broadcast_source(dist, rank, make, dev)
    chans = list(multi.shard_channels(5, world, rank))
    # stand-in for the per-channel GPU chain: a channel-dependent checksum of the shared source
    work = sum(float((src * (c + 1)).sum()) for c in chans)
    units, secs = multi.aggregate(dist, len(chans) * src.numel(), 0.5 + rank, dev)
    q.put((rank, made, chans, float(src.sum()), work, units, secs, gbs is not None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_channel_sharding_and_broadcast():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, made0, ch0, s0, w0, u0, t0, b0), (r1, made1, ch1, s1, w1, u1, t1, b1) = res
    assert made0 == [0] and made1 == []            # only the owning rank synthesises the source
    assert s0 == s1                                # identical source on every rank after the broadcast
    assert sorted(ch0 + ch1) == [0, 1, 2, 3, 4] and not set(ch0) & set(ch1)
    assert u0 == u1 == 5 * 100_000                 # units: sum over ranks
    assert t0 == t1 == 1.5                         # wall time: max over ranks
    assert b0 and b1


def test_shard_channels_properties():
    for n in (1, 7, 32, 256):
        for world in (1, 2, 3, 4, 8):
            owned = [c for r in range(world) for c in multi.shard_channels(n, world, r)]
            assert owned == list(range(n))
            sizes = [len(multi.shard_channels(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert len(multi.shard_channels(256, 8, 3)) == 32   # BASELINE configs[3]: 32 chains per GPU
    assert multi.channel_frequency(128, 256, 8e3) == 0.0


def test_single_process_passthrough():
    t, gbs = multi.broadcast_source(None, 0, lambda: torch.ones(4), torch.device("cpu"))
    assert gbs is None and t.sum() == 4
    assert multi.aggregate(None, 10, 2.0, torch.device("cpu")) == (10.0, 2.0)
