"""One rank of the multi-GPU rr_fanout_* test (tests/test_gpu_fanout.py::test_fanout_across_gpus): started as a child process
BEFORE anything in it touches a GPU, one process per device.

    python tests/fanout_rank.py <rank> <world> <exchange_dir> <flags> <tile_bytes> <ntiles>

Rank 0 writes the 128-byte RCCL group id to <exchange_dir>/id.bin (what an application ships out of band); every rank then
drives the fan-out protocol — the owner produces tile t + 1 while tile t is consumed — and prints one JSON line with the
CRC32 of every tile as it arrived on this rank."""
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tile_bytes_of(t, n):
    """the deterministic content of tile t (every rank can compute the expected checksum)"""
    r = np.random.default_rng(1000 + t)
    return r.integers(0, 256, n, dtype=np.uint8)


def main():
    rank, world, xdir, flags, nbytes, ntiles = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    import torch
    import rustradio_amd as rr
    torch.cuda.set_device(rank)
    rr.set_device(rank)
    idp = os.path.join(xdir, "id.bin")
    if rank == 0:
        gid = rr.fanout_unique_id()
        with open(idp + ".tmp", "wb") as f:
            f.write(gid)
        os.replace(idp + ".tmp", idp)
    else:
        t0 = time.time()
        while not os.path.exists(idp):
            if time.time() - t0 > 120:
                raise SystemExit("no group id")
            time.sleep(0.05)
        gid = open(idp, "rb").read()
    fan = rr.Fanout(gid, rank, world, nbytes, 0, flags | rr.FANOUT_TIMING)
    dev = torch.device("cuda", rank)
    s_src, s_cmp = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    host = [torch.from_numpy(tile_bytes_of(t, nbytes)).pin_memory() for t in range(ntiles)] if rank == 0 else None
    got = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    crcs = []

    def produce(t):
        p = fan.produce_buf(t, s_src.cuda_stream)
        if rank == 0:
            view = torch.as_tensor(_Dev(p, nbytes), device=dev)
            with torch.cuda.stream(s_src):
                view.copy_(host[t], non_blocking=True)
        fan.submit(t, s_src.cuda_stream)

    produce(0)
    for t in range(ntiles):
        if t + 1 < ntiles:
            produce(t + 1)
        x = fan.acquire(t, s_cmp.cuda_stream)
        with torch.cuda.stream(s_cmp):
            got.copy_(torch.as_tensor(_Dev(x, nbytes), device=dev), non_blocking=True)
        fan.release(t, s_cmp.cuda_stream)
        s_cmp.synchronize()
        crcs.append(zlib.crc32(got.cpu().numpy().tobytes()))
    torch.cuda.synchronize()
    ms, nb = fan.stats()
    print(json.dumps({"rank": rank, "crcs": crcs, "fanout_ms": ms, "timed": nb}), flush=True)


class _Dev:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


if __name__ == "__main__":
    main()
