"""GPU: rr_fanout_* (include/rustradio_amd.h) — the C-ABI streaming fan-out of a shared source (replaces the
reference's in-process Tee tree, src/tee.rs:10-24).  The box has one GPU, so the group has one rank: once without a
communicator (the plain double buffer) and once through RCCL (RR_FANOUT_RCCL_ALWAYS: ncclCommInitRank + ncclBroadcast on
the communication stream).  The "source block" and the consumer are real blocks driven on device pointers."""
import json
import os
import subprocess
import sys
import tempfile
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rr():
    import rustradio_amd as m
    return m


def _drive(rr, fan, ntiles, n, stream, two_streams):
    dev = torch.device("cuda", 0)
    src = torch.arange(ntiles * n, dtype=torch.float32, device=dev) * 0.5
    out = torch.zeros(ntiles * n, dtype=torch.float32, device=dev)
    producer, consumer = rr.MultiplyConst(1.0), rr.MultiplyConst(3.0)
    s_src = torch.cuda.Stream() if two_streams else stream
    s_src.wait_stream(stream)                                   # src / out were made on the compute stream

    def produce(t):
        p = fan.produce_buf(t, s_src.cuda_stream)
        st, c, pr, need = producer.work_dev(src.data_ptr() + 4 * t * n, n, p, n, s_src.cuda_stream)
        assert (c, pr) == (n, n)
        fan.submit(t, s_src.cuda_stream)

    produce(0)
    for t in range(ntiles):
        if t + 1 < ntiles:
            produce(t + 1)                                       # tile t + 1 in flight while tile t is consumed
        x = fan.acquire(t, stream.cuda_stream)
        st, c, pr, need = consumer.work_dev(x, n, out.data_ptr() + 4 * t * n, n, stream.cuda_stream)
        assert (c, pr) == (n, n)
        fan.release(t, stream.cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(out, src * 3.0)


@pytest.mark.parametrize("two_streams", [False, True])
def test_one_rank_fanout_is_a_double_buffer(rr, two_streams):
    n = 1 << 20
    fan = rr.Fanout(None, 0, 1, 4 * n)
    _drive(rr, fan, 7, n, torch.cuda.current_stream(), two_streams)
    assert fan.stats() == (0.0, 0)                               # no communicator, nothing timed


@pytest.mark.parametrize("mesh", [False, True])
def test_one_rank_fanout_through_rccl(rr, mesh):
    """ncclBroadcast, and (mesh) ncclScatter + in-place ncclAllGather, on a one-rank group"""
    n = (1 << 20) + 3
    gid = rr.fanout_unique_id()
    assert len(gid) == rr.FANOUT_ID_BYTES and any(gid)
    fan = rr.Fanout(gid, 0, 1, 4 * n, flags=rr.FANOUT_RCCL_ALWAYS | rr.FANOUT_TIMING | (rr.FANOUT_MESH if mesh else 0))
    _drive(rr, fan, 7, n, torch.cuda.current_stream(), True)
    ms, nb = fan.stats()
    assert nb == 2 and ms > 0                                    # tiles 0 and 4 of the 7: every 4th broadcast is timed
    assert fan.stats() == (0.0, 0)                               # drained


def test_fanout_protocol_errors(rr):
    with pytest.raises(RuntimeError, match="out of range"):
        rr.Fanout(None, 1, 1, 1024)
    with pytest.raises(RuntimeError, match="nonzero"):
        rr.Fanout(None, 0, 1, 0)
    with pytest.raises(RuntimeError, match="needs the id"):
        rr.Fanout(None, 0, 2, 1024)
    fan = rr.Fanout(None, 0, 1, 1024)
    with pytest.raises(RuntimeError, match="in order"):
        fan.submit(1)
    with pytest.raises(RuntimeError, match="not in the double buffer"):
        fan.acquire(0)
    fan.produce_buf(0); fan.submit(0)
    fan.produce_buf(1); fan.submit(1)
    with pytest.raises(RuntimeError, match="has not been released"):
        fan.produce_buf(2)                                       # tile 0 still owns that half
    with pytest.raises(RuntimeError, match="never submitted"):
        fan.release(5)
    fan.acquire(0); fan.release(0)
    assert fan.produce_buf(2)
    fan.submit(2)
    with pytest.raises(RuntimeError, match="not in the double buffer"):
        fan.acquire(0)                                           # overwritten by tile 2


def test_abi_fanout_adapter_matches_tilefanout_interface(rr):
    """multi.AbiFanout (bench.py --fanout abi) on a one-rank group through RCCL: torch aliases of the library's double
    buffer, tiles produced one ahead on the source stream, consumed in order by a real block."""
    from rustradio_amd import multi
    dev = torch.device("cuda", 0)
    n, ntiles = 1 << 18, 6
    store = torch.arange(ntiles * n, dtype=torch.float32, device=dev)

    def produce(t, out):
        out.copy_(store[t * n:(t + 1) * n], non_blocking=True)

    fan = multi.AbiFanout(rr, None, 0, n, torch.float32, dev, produce, rccl_always=True)
    stream = torch.cuda.current_stream()
    blk = rr.MultiplyConst(2.0)
    out = torch.zeros(ntiles * n, dtype=torch.float32, device=dev)
    fan.prefetch(0)
    fan.reset_timing()
    for t in range(ntiles):
        if t + 1 < ntiles:
            fan.prefetch(t + 1)
        x = fan.acquire(t, stream)
        assert x.dtype == torch.float32 and x.numel() == n
        blk.work_dev(x.data_ptr(), n, out.data_ptr() + 4 * t * n, n, stream.cuda_stream)
        fan.release(t, stream)
    torch.cuda.synchronize()
    assert torch.equal(out, store * 2.0)
    ms, nb = fan.broadcast_ms()
    assert nb == 1 and ms > 0 and fan.n_bcast == ntiles          # tile 4 (tile 0's timing was drained by reset_timing)


def _run_ranks(rr, world, flags, nbytes, ntiles=6):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fanout_rank import tile_bytes_of
    want = [zlib.crc32(tile_bytes_of(t, nbytes).tobytes()) for t in range(ntiles)]
    with tempfile.TemporaryDirectory() as xdir:
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fanout_rank.py"),
                                   str(r), str(world), xdir, str(flags), str(nbytes), str(ntiles)],
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
        outs = [p.communicate(timeout=600) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, (r, e[-2000:])
        d = json.loads([l for l in o.strip().split("\n") if l.startswith("{")][-1])
        assert d["rank"] == r and d["crcs"] == want, (r, d["crcs"], want)
        assert d["timed"] >= 1 and d["fanout_ms"] > 0


def test_fanout_rank_processes_one_rank(rr):
    """the rank script of the multi-GPU test below on a one-rank RCCL group (what this box can run): child process, group id
    through a file, tiles produced one ahead, CRC32 of every tile as acquired"""
    _run_ranks(rr, 1, rr.FANOUT_RCCL_ALWAYS | rr.FANOUT_MESH, (1 << 20) + 13)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: the multi-rank RCCL paths of rr_fanout_* "
                    "(ncclBroadcast with world > 1, in-place ncclScatter + ncclAllGather, cross-rank event ordering) have never "
                    "run on hardware — the development box has one GPU (ADVICE r2)")
@pytest.mark.parametrize("mesh", [False, True])
@pytest.mark.parametrize("nbytes", [4 << 20, (4 << 20) + 13])
def test_fanout_across_gpus(rr, mesh, nbytes):
    """rr_fanout_* on one rank PER GPU (2 ranks, or 4 when the node has them): rank processes are started before any of
    them touches a GPU, the group id travels through a file, every rank reports the CRC32 of every tile it acquired — all
    must equal the owner's content, for the broadcast and for the mesh algorithm, also when the tile does not divide by the
    number of ranks."""
    world = 4 if torch.cuda.device_count() >= 4 else 2
    _run_ranks(rr, world, rr.FANOUT_MESH if mesh else 0, nbytes)
