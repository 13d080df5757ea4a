#!/usr/bin/env python3
"""GPU box: end-to-end rate of an examples/rtl_fm.rs-style graph, host memory in and out (PCIe-inclusive):
  (a) every ring in host memory: each block's work() is a PCIe round trip (rr_block_work);
  (b) rings in HBM (rr_dstream): only the source push and the sink pop cross the bus;
  (c) as (b) with the three blocks fused (rr_fm_chain_create)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import rustradio_amd as rr
from harness import run_chain

AGAIN = 0


def run_chain_device(blocks, x, stream_bytes=4_096_000):
    rings = [rr.DeviceStream(blocks[0].in_dtype, stream_bytes)] + [rr.DeviceStream(b.out_dtype, stream_bytes) for b in blocks]
    pos, outs = 0, []
    while True:
        moved = rings[0].push(x[pos:]); pos += moved
        for i, b in enumerate(blocks):
            while True:
                st, c, p, need = b.work_streams(rings[i], rings[i + 1])
                moved += c + p
                if st != AGAIN or (c == 0 and p == 0):
                    break
        y = rings[-1].pop(); moved += len(y)
        if len(y):
            outs.append(y)
        if moved == 0:
            break
    return np.concatenate(outs)


fs, n = 2.4e6, 12_000_000
rng = np.random.default_rng(1)
x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
taps = rr.low_pass_complex(fs, 100e3, 12.5e3)
mk = lambda: [rr.FftFilter(taps), rr.RationalResampler(1, 6, np.complex64), rr.QuadratureDemod(1.0)]
for name, fn in (("host rings, 3 blocks", lambda: run_chain(mk(), x)),
                 ("HBM rings, 3 blocks", lambda: run_chain_device(mk(), x)),
                 ("HBM rings, fused FmChain", lambda: run_chain_device([rr.FmChain(taps, 1, 6)], x))):
    fn()
    t0 = time.perf_counter(); y = fn(); dt = time.perf_counter() - t0
    print(f"{name:28s} {n / dt / 1e6:8.1f} Msamples/s  ({len(y)} outputs, {dt*1e3:.0f} ms)")
rr.host_register(x)          # what the shim does once for the source ring (rr_host_register)
for name, fn in (("pinned src, host rings", lambda: run_chain(mk(), x)),
                 ("pinned src, HBM rings", lambda: run_chain_device(mk(), x)),
                 ("pinned src, HBM, fused", lambda: run_chain_device([rr.FmChain(taps, 1, 6)], x))):
    fn()
    t0 = time.perf_counter(); y = fn(); dt = time.perf_counter() - t0
    print(f"{name:28s} {n / dt / 1e6:8.1f} Msamples/s  ({len(y)} outputs, {dt*1e3:.0f} ms)")
rr.host_unregister(x)
