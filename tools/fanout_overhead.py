#!/usr/bin/env python3
"""GPU box: host-side cost of the streamed fan-out loop of bench.py at N > 1, measured on a ONE-rank RCCL group (the
collectives move nothing, the host does everything it does at 8 ranks): wall time per step of the fm_multi loop with the
fan-out (both algorithms) against the bare loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import numpy as np, torch, torch.distributed as dist
import rustradio_amd as rr
from rustradio_amd import multi

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000
taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
blk = rr.FmMulti(multi.cfg4_taps(taps, range(32)), 1, 6, 1.0)
store = torch.rand(2 * n, device=dev) * 2 - 1
cap = n // 6 + 1024
out = torch.empty(32 * cap, device=dev)
stream = torch.cuda.current_stream()


def loop(fan, steps=60, k=1):
    t = 0
    if fan: fan.prefetch(0)
    def one(t):
        if fan:
            T, sub = divmod(t, k)
            if sub == 0: fan.prefetch(T + 1)
            x = fan.acquire(T, stream); p = x.data_ptr() + sub * 8 * n
        else:
            p = store.data_ptr()
        blk.work_dev(p, n, out.data_ptr(), cap, stream.cuda_stream)
        if fan and t % k == k - 1: fan.release(t // k, stream)
    for _ in range(8): one(t); t += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): one(t); t += 1
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return host / steps * 1e3, (time.perf_counter() - t0) / steps * 1e3


def produce(t, o): o.copy_(store, non_blocking=True)
print("samples/step", n)
print("bare loop                host %.3f ms/step, wall %.3f" % loop(None))
for algo in ("bcast", "scatter_allgather"):
    fan = multi.TileFanout(dist, 0, 2 * n, torch.float32, dev, produce, algo="bcast")
    if algo != "bcast":
        fan.can_scatter, fan.algo = True, algo          # one rank: force the two-collective form for its host cost
    print("%-24s host %.3f ms/step, wall %.3f" % (algo, *loop(fan)))
for k in (2, 4):
    def producek(t, o): o.view(k, 2 * n).copy_(store.unsqueeze(0).expand(k, 2 * n), non_blocking=True)
    for algo in ("bcast", "scatter_allgather"):
        fan = multi.TileFanout(dist, 0, 2 * n * k, torch.float32, dev, producek, algo="bcast")
        if algo != "bcast":
            fan.can_scatter, fan.algo = True, algo
        print("%-24s host %.3f ms/step, wall %.3f" % (f"{algo} tile={k} steps", *loop(fan, 60, k)))
for mesh, timing, rccl in ((False, True, True), (True, True, True), (False, False, True), (False, False, False)):
    fan = multi.AbiFanout(rr, None, 0, 2 * n, torch.float32, dev, produce, rccl_always=rccl, mesh=mesh, timing=timing)
    print("%-24s host %.3f ms/step, wall %.3f" % ("abi " + ("mesh" if mesh else "bcast") + ("" if timing else " untimed") + ("" if rccl else " no-rccl"), *loop(fan)))
if os.environ.get("RR_PROFILE"):
    import cProfile, pstats
    fan = multi.TileFanout(dist, 0, 2 * n, torch.float32, dev, produce, algo="bcast")
    fan.can_scatter, fan.algo = True, "scatter_allgather"
    pr = cProfile.Profile(); pr.enable(); loop(fan, 200); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
dist.destroy_process_group()
