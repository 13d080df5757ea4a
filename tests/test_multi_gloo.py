"""world_size 2, gloo: the N > 1 host logic bench.py uses — channel sharding, the STREAMING fan-out of the shared IQ
source (multi.TileFanout: rank 0 produces tile t+1 and broadcasts it while every rank works on tile t), max-time /
sum-units reduction.  The CPU test drives it with CPU tensors and a checksum per channel; the GPU test (-m gpu) runs both
ranks on the visible GPU, each driving its real rr.FmMulti block on the streamed tiles, and checks every channel of the
two shards against its own oracle chain."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from rustradio_amd import multi


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _tile(t, n):
    g = torch.Generator().manual_seed(1234 + t)
    return torch.rand(n, generator=g, dtype=torch.float32)


def _worker(rank, world, port, q, algo="bcast"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    made = []

    def make():
        made.append(rank)
        return _tile(0, 100_000)

    src, gbs = multi.broadcast_source(dist, rank, make, dev)
    chans = list(multi.shard_channels(5, world, rank))
    # streaming fan-out: 6 tiles, produced on rank 0 only, consumed in order on every rank
    produced = []

    def produce(t, out):
        produced.append(t)
        out.copy_(_tile(t, out.numel()))

    fan = multi.TileFanout(dist, rank, 50_000, torch.float32, dev, produce, algo=algo)
    sums = []
    fan.prefetch(0)
    for t in range(6):
        fan.prefetch(t + 1)
        x = fan.acquire(t)
        # stand-in for the per-channel chain: a channel-dependent checksum of the tile
        sums.append([float((x * (c + 1)).sum()) for c in chans])
        fan.release(t)
    units, secs = multi.aggregate(dist, len(chans) * src.numel(), 0.5 + rank, dev)
    q.put((rank, made, chans, float(src.sum()), sums, produced, units, secs, gbs is not None, fan.n_bcast, fan.algo,
           fan.calibration))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, target, *extra):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + extra) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_channel_sharding_and_streaming_fanout():
    ((r0, made0, ch0, s0, sums0, prod0, u0, t0, b0, n0, a0, c0),
     (r1, made1, ch1, s1, sums1, prod1, u1, t1, b1, n1, a1, c1)) = _run(2, _worker)
    assert a0 == a1 == "bcast" and c0 is None
    assert made0 == [0] and made1 == []            # only the owning rank synthesises the source
    assert s0 == s1                                # identical source on every rank after the broadcast
    assert sorted(ch0 + ch1) == [0, 1, 2, 3, 4] and not set(ch0) & set(ch1)
    assert prod0 == list(range(7)) and prod1 == []  # tiles 0..6 produced in order on rank 0 only (one prefetched ahead)
    assert n0 == n1 == 7
    for t in range(6):                             # every rank saw tile t at step t: checksums match the tile itself
        ref = _tile(t, 50_000)
        for c, v in zip(ch0, sums0[t]):
            assert v == float((ref * (c + 1)).sum())
        for c, v in zip(ch1, sums1[t]):
            assert v == float((ref * (c + 1)).sum())
    assert u0 == u1 == 5 * 100_000                 # units: sum over ranks
    assert t0 == t1 == 1.5                         # wall time: max over ranks
    assert b0 and b1


@pytest.mark.parametrize("world,algo", [(2, "scatter_allgather"), (3, "scatter_allgather"), (3, "auto")])
def test_fanout_by_scatter_and_allgather(world, algo):
    """the full-mesh fan-out: 1/world of the tile to each rank, then an all-gather between the receivers — same tiles on
    every rank as the broadcast, also when the tile does not divide by the number of ranks (50,000 over 3); "auto" times
    both and every rank keeps the same one"""
    res = _run(world, _worker, algo)
    algos = {r[10] for r in res}
    assert len(algos) == 1 and algos <= {"bcast", "scatter_allgather"}
    if algo != "auto":
        assert algos == {algo}
    else:
        cal = [r[11] for r in res]
        assert all(c == cal[0] and set(c) == {"bcast", "scatter_allgather"} for c in cal)
    assert [r[5] for r in res] == [list(range(7))] + [[]] * (world - 1)      # produced on rank 0 only
    owned = []
    for r in res:
        for t in range(6):
            ref = _tile(t, 50_000)
            for c, v in zip(r[2], r[4][t]):
                assert v == float((ref * (c + 1)).sum())
        owned += r[2]
    assert sorted(owned) == [0, 1, 2, 3, 4]


def test_shard_channels_properties():
    for n in (1, 7, 32, 256):
        for world in (1, 2, 3, 4, 8):
            owned = [c for r in range(world) for c in multi.shard_channels(n, world, r)]
            assert owned == list(range(n))
            sizes = [len(multi.shard_channels(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert len(multi.shard_channels(256, 8, 3)) == 32   # BASELINE configs[3]: 32 chains per GPU
    assert multi.channel_frequency(128, 256, 8e3) == 0.0
    # configs[4] on 8 GPUs: one slice of the analytic band (0, fs/2) per GPU, none for a single GPU
    assert multi.cfg5_translate_hz(0, 1) == 0.0
    fs = [multi.cfg5_translate_hz(r, 8) for r in range(8)]
    assert fs == sorted(fs) and 0 < fs[0] and fs[-1] < 50e6 and len(set(fs)) == 8


def test_single_process_passthrough():
    t, gbs = multi.broadcast_source(None, 0, lambda: torch.ones(4), torch.device("cpu"))
    assert gbs is None and t.sum() == 4
    assert multi.aggregate(None, 10, 2.0, torch.device("cpu")) == (10.0, 2.0)
    fan = multi.TileFanout(None, 0, 8, torch.float32, torch.device("cpu"), lambda t, out: out.fill_(float(t)))
    assert [float(fan.acquire(t)[0]) for t in range(4)] == [0.0, 1.0, 2.0, 3.0]


# ---- both ranks on the visible GPU, real blocks ---------------------------------------------------------------------
def _gpu_worker(rank, world, port, q, ntiles, tile):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import rustradio_amd as rr
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    rr.set_device(0)
    x = _stations(ntiles * tile)
    store = torch.from_numpy(x.view(np.float32)).to(dev) if rank == 0 else None

    def produce(t, out):
        out.copy_(store[2 * t * tile:2 * (t + 1) * tile], non_blocking=True)

    fan = multi.TileFanout(dist, rank, 2 * tile, torch.float32, dev, produce)
    proto = rr.low_pass_complex(multi.CFG4_FS, 100e3, 12.5e3)
    chans = list(multi.shard_channels(8, world, rank))           # 4 channels per rank around the band centre
    taps = multi.cfg4_taps(proto, [c + 124 for c in chans])
    blk = rr.FmMulti(taps, 1, 6, 1.0)
    cap = tile // 6 + 1024
    dout = torch.zeros(len(chans) * cap, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream()
    outs = [[] for _ in chans]
    fan.prefetch(0)
    for t in range(ntiles):
        fan.prefetch(t + 1) if t + 1 < ntiles else None
        xt = fan.acquire(t, stream)
        st, c, p, need = blk.work_dev(xt.data_ptr(), tile, dout.data_ptr(), cap, stream.cuda_stream)
        fan.release(t, stream)
        assert c == tile                                          # the fused block takes the whole tile (pending kept inside)
        blk.sync()
        h = dout.cpu().numpy().reshape(len(chans), cap)
        for i in range(len(chans)):
            outs[i].append(h[i, :p].copy())
    q.put((rank, [c + 124 for c in chans], [np.concatenate(o) for o in outs]))
    dist.barrier()
    dist.destroy_process_group()


def _stations(n):
    t = np.arange(n, dtype=np.float64)
    r = np.random.default_rng(5)
    x = 0.01 * (r.standard_normal(n) + 1j * r.standard_normal(n))
    for i, f in enumerate([-24e3, 0.0, 17e3]):
        x += np.exp(2j * np.pi * np.cumsum(f + 75e3 * np.sin(2 * np.pi * (1e3 + 50 * i) * t / multi.CFG4_FS)) / multi.CFG4_FS) / 3
    return x.astype(np.complex64)


@pytest.mark.gpu
def test_two_ranks_drive_real_blocks_on_streamed_tiles():
    from harness import run_chain
    from oracle import pyoracle as orc
    ntiles, tile = 5, 100_000
    res = _run(2, _gpu_worker, ntiles, tile)
    x = _stations(ntiles * tile)
    proto = orc.low_pass_complex(multi.CFG4_FS, 100e3, 12.5e3)
    seen = []
    for rank, chans, outs in res:
        taps = multi.cfg4_taps(proto, chans)
        for i, c in enumerate(chans):
            yo = run_chain([orc.FftFilter(taps[i]), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)], x)
            ro = run_chain([orc.FftFilter(taps[i]), orc.RationalResampler(1, 6)], x)
            yg = outs[i]
            assert len(yg) == len(yo) > 0
            eps = 1e-5 * float(np.max(np.abs(ro)))
            mag = np.abs(ro.astype(np.complex128))
            bound = 1e-5 * np.pi + eps / np.maximum(mag[:-1], 1e-30) + eps / np.maximum(mag[1:], 1e-30)
            d = np.abs(yg.astype(np.float64) - yo.astype(np.float64))
            d = np.minimum(d, 2 * np.pi - d)
            assert np.all(d <= bound[:len(d)]), (c, float(np.max(d - bound[:len(d)])))
            seen.append(c)
    assert sorted(seen) == list(range(124, 132))                # the union of the two shards: every channel once


# ---- the RCCL code path itself, as far as one GPU allows: a one-rank "nccl" group -------------------------------------
def _rccl_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # exactly bench.py's call
    n = 1 << 20
    store = torch.arange(8 * n, dtype=torch.float32, device=dev)

    def produce(t, out):
        out.copy_(store[t * n:(t + 1) * n], non_blocking=True)

    fan = multi.TileFanout(dist, rank, n, torch.float32, dev, produce)
    stream = torch.cuda.current_stream()
    sums = []
    fan.prefetch(0)
    for t in range(8):
        if t + 1 < 8:
            fan.prefetch(t + 1)
        x = fan.acquire(t, stream)
        sums.append(x.double().sum())                       # compute-stream work on the tile
        fan.release(t, stream)
    torch.cuda.synchronize()
    ms, nb = fan.broadcast_ms()
    units, secs = multi.aggregate(dist, 3.0, 0.25, dev)
    q.put((rank, [float(v) for v in sums], ms, nb, units, secs, dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def _abi_check_worker(rank, world, port, q):
    import torch.distributed as dist
    import rustradio_amd as rr
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ok, why = multi.verify_abi_fanout(rr, dist, rank, dev)
    q.put((rank, ok, why))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_abi_fanout_self_check_on_a_one_rank_rccl_group():
    """the self-check bench.py runs before it relies on rr_fanout_* (multi.verify_abi_fanout: known tiles through both
    algorithms, checksums, all-reduced verdict), on the group this box can form"""
    (rank, ok, why), = _run(1, _abi_check_worker)
    assert ok and why == ""


@pytest.mark.gpu
def test_tile_fanout_on_a_one_rank_rccl_group():
    """bench.py's N > 1 path runs over RCCL, which wants one GPU per rank; on the one-GPU box this drives the same calls
    (init_process_group("nccl", device_id=...), async broadcast on the communication stream, event hand-over, all_reduce)
    through a one-rank RCCL group: tiles arrive intact and in order, every broadcast is timed."""
    (rank, sums, ms, nb, units, secs, backend), = _run(1, _rccl_worker)
    n = 1 << 20
    for t, v in enumerate(sums):
        lo = t * n
        assert v == float(n) * lo + n * (n - 1) / 2
    assert backend == "nccl" and nb == 2 and ms > 0          # tiles 0 and 4 of the 8 (TileFanout.TIME_EVERY)
    assert (units, secs) == (3.0, 0.25)


def test_fanout_prediction_table():
    """multi.predict_fanout: the numbers DESIGN §6 writes down before any multi-GPU run (76.8 MB Complex<f32> tile of four
    0.0806 ms steps over ~153 GB/s point-to-point links): one broadcast is single-link-bound at every N, the mesh form hides
    behind the compute from 4 GPUs on, two GPUs share one link whatever the algorithm."""
    tile, comp = 4 * 19_200_000, 4 * 0.0806
    p2, p4, p8 = (multi.predict_fanout(n, tile, comp) for n in (2, 4, 8))
    assert p2["bcast"]["fanout_ms_per_tile"] == p4["bcast"]["fanout_ms_per_tile"] == p8["bcast"]["fanout_ms_per_tile"]
    assert abs(p8["bcast"]["fanout_ms_per_tile"] - (tile / 153e9 * 1e3 + 0.02)) < 1e-3
    assert 0.6 < p8["bcast"]["efficiency"] < 0.65
    assert p2["scatter_allgather"]["efficiency"] < 0.65 and p4["scatter_allgather"]["efficiency"] == 1.0 == p8["scatter_allgather"]["efficiency"]
    assert abs(p8["scatter_allgather"]["fanout_ms_per_tile"] - (2 * tile / (8 * 153e9) * 1e3 + 0.04)) < 1e-3
    u8 = multi.predict_fanout(8, tile // 4, comp)            # the RTL-SDR wire format: 2 B per sample
    assert u8["bcast"]["efficiency"] == 1.0
    assert multi.predict_fanout(1, tile, comp)["bcast"]["fanout_ms_per_tile"] == 0.0


def test_bench_watchdog_names_the_stage_and_ends_the_rank():
    """bench.py --gpus N: a rank that stands in one stage longer than --stage-timeout (a collective another rank never
    entered) says where, dumps its stacks and exits 4 — instead of the whole job waiting for the launcher's limit in silence."""
    code = ("import sys, time; sys.argv = ['bench.py']; import bench; bench.start_watchdog(3, 0.3); "
            "bench.stage('fan-out self-check'); time.sleep(20); print('survived')")
    out = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), capture_output=True, text=True, timeout=120)
    assert out.returncode == 4, (out.returncode, out.stderr[-500:])
    assert "rank 3" in out.stderr and "'fan-out self-check'" in out.stderr and "survived" not in out.stdout
