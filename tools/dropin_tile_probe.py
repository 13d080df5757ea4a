#!/usr/bin/env python3
"""GPU box: the drop-in path (rr_block_work on 4,096,000-byte registered HOST windows, kernels working in place over PCIe)
for FftFilter 401 taps on each tile size, against the path's own ceiling (a block that only copies) — us per call.
VERDICT r4 item 7."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustradio_amd as rr


def per_call(blk, x, out, out_cap, seconds=0.6):
    for _ in range(5): blk.work_into(x, out, out_cap)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        blk.work_into(x, out, out_cap); n += 1
    return (time.perf_counter() - t0) / n * 1e6


rng = np.random.default_rng(7)
rin, rout = rr.host_ring(4_096_000), rr.host_ring(4_096_000)
rr.host_register(rin); rr.host_register(rout)
xf = rin[:4_096_000].view(np.float32); xf[:] = rng.uniform(-1, 1, len(xf)).astype(np.float32)
us = per_call(rr.MultiplyConst(1.0), xf, rout[:4_096_000].view(np.float32), 1_024_000)
print(f"copy (MultiplyConst f32)      {us:7.1f} us per call = {4.096 / us * 1e3:5.1f} GB/s each way")
xc = rin[:4_096_000].view(np.complex64)
oc = rout[:4_096_000].view(np.complex64)
taps = rr.low_pass_complex(10e6, 1e6, 60e3)
for lg in (0, 10, 11, 12, 13):
    with rr.build_options(**({"fft_log2f": lg} if lg else {})):
        b = rr.FftFilter(taps)
    us = per_call(b, xc, oc, 512_000)
    print(f"FftFilter 401 taps F={'auto' if not lg else 1 << lg:>5}  {us:7.1f} us per call = {512_000 / us:7.1f} Msamples/s")
tb = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
xb = rin[:4_096_000]
of = rout[:4_096_000].view(np.float32)
us = per_call(rr.FmChainU8(tb, 1, 6, 1.0), xb, of, 1_024_000)
print(f"fused RTL-SDR chain           {us:7.1f} us per call = {2_048_000 / us:7.1f} Msamples/s")
