#!/usr/bin/env python3
"""GPU box: do the kernels of a zero-copy call (rr_block_work on rr_host_register'd windows) always see what the CPU wrote
into the window a moment ago, and does the CPU always see what they wrote?  The same addresses carry new data on every
call (a ring does that); any stale line — CPU cache, GPU L2 — shows up as a mismatch with the value computed on the host.

    python tools/zerocopy_coherence.py [iterations] [RR_LIB_PATH via env]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustradio_amd as rr

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rng = np.random.default_rng(3)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400_000
ring_in = np.zeros(N + 64, np.float32); ring_out = np.zeros(N + 64, np.float32)
rr.host_register(ring_in); rr.host_register(ring_out)
blk = rr.MultiplyConst(0.5)
bad = 0
t0 = time.time()
for k in range(iters):
    n = int(rng.integers(1, N))
    oi, oo = int(rng.integers(0, 64)), int(rng.integers(0, 64))
    x = rng.standard_normal(n).astype(np.float32)
    ring_in[oi:oi + n] = x                                   # the writer fills the window ...
    ring_out[oo:oo + n] = -7.0                               # (poison: a store that never arrives shows)
    st, c, p, need = blk.work_into(ring_in[oi:oi + n], ring_out[oo:], n)   # ... and the block runs on it in place
    assert c == n and p == n
    y = ring_out[oo:oo + n]
    if not np.array_equal(y, x * np.float32(0.5)):
        d = np.flatnonzero(y != x * np.float32(0.5))
        stale_in = int(np.sum(y[d] != -7.0))
        bad += 1
        print(f"iteration {k}: {len(d)} of {n} outputs differ (first {d[:4]}, last {d[-2:]}); {stale_in} look like stale INPUT, "
              f"{len(d) - stale_in} like a store that did not arrive", flush=True)
print(f"{iters} calls, {bad} with mismatches, {time.time() - t0:.1f} s")
rr.host_unregister(ring_in); rr.host_unregister(ring_out)
sys.exit(1 if bad else 0)
