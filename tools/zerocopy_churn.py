#!/usr/bin/env python3
"""GPU box: register / run in place / unregister / free, over and over with FRESH arrays (the allocator hands the same virtual
addresses out again with new pages behind them) — does a zero-copy call right after a re-registration always reach the new
pages?  (tests/harness.py::drive_registered used to do exactly this once per block.)

    python tools/zerocopy_churn.py [rounds]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustradio_amd as rr

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 600
variant = sys.argv[2] if len(sys.argv) > 2 else "plain"      # plain | touch (write every page first) | nohuge (touch + MADV_NOHUGEPAGE)
                                                             # dma: page-locked behind the library's back (hipHostRegister): staged DMA copies
                                                             # pageable: no registration at all
                                                             # mmap: like plain, the arrays are views of fresh anonymous mmap()s (page-
                                                             #       aligned, unmapped at the end of the round) instead of numpy's own
                                                             # byhand: hipHostRegister + hipHostGetDevicePointer here, the block's
                                                             #         work_dev on the device views (no work_host, no registry)
import ctypes
_libc = ctypes.CDLL("libc.so.6", use_errno=True)
def prepare(a):
    if variant == "plain":
        return
    a[:] = 1.0                                                # every page written: real, private pages before they are locked
    if variant == "nohuge":
        lo = a.ctypes.data & ~4095
        hi = (a.ctypes.data + a.nbytes + 4095) & ~4095
        _libc.madvise(ctypes.c_void_p(lo), ctypes.c_size_t(hi - lo), 15)       # MADV_NOHUGEPAGE
_hip = ctypes.CDLL("libamdhip64.so") if variant in ("dma", "byhand", "byhandmmap") else None
def devptr(a):
    p = ctypes.c_void_p()
    assert _hip.hipHostGetDevicePointer(ctypes.byref(p), ctypes.c_void_p(a.ctypes.data), 0) == 0
    return p.value
def reg(a):
    if variant in ("dma", "byhand", "byhandmmap"):
        assert _hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(a.nbytes), 0) == 0
    elif variant != "pageable":
        rr.host_register(a)
def unreg(a):
    if variant in ("dma", "byhand", "byhandmmap"):
        assert _hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data)) == 0
    elif variant != "pageable":
        rr.host_unregister(a)
rng = np.random.default_rng(5)
blk = rr.MultiplyConst(0.5)
bad, addrs = 0, set()
t0 = time.time()
for k in range(rounds):
    n = int(rng.integers(50_000, 600_000))
    if variant in ("mmap", "byhandmmap"):
        import mmap as _mm
        m1, m2 = _mm.mmap(-1, (n + 16) * 4), _mm.mmap(-1, (n + 16) * 4)
        ring_in, ring_out = np.frombuffer(m1, np.float32), np.frombuffer(m2, np.float32)
    else:
        ring_in = np.zeros(n + 16, np.float32); ring_out = np.zeros(n + 16, np.float32)  # fresh mappings
    addrs.add(ring_in.ctypes.data)
    prepare(ring_in); prepare(ring_out)
    reg(ring_in); reg(ring_out)
    try:
        for j in range(int(rng.integers(1, 6))):
            x = rng.standard_normal(n).astype(np.float32)
            ring_in[3:3 + n] = x
            ring_out[:] = -7.0
            if variant in ("byhand", "byhandmmap"):
                st, c, p, need = blk.work_dev(devptr(ring_in) + 12, n, devptr(ring_out) + 20, n, 0)
                assert _hip.hipDeviceSynchronize() == 0
            else:
                st, c, p, need = blk.work_into(ring_in[3:3 + n], ring_out[5:], n)
            y = ring_out[5:5 + n]
            if not np.array_equal(y, x * np.float32(0.5)):
                d = np.flatnonzero(y != x * np.float32(0.5))
                bad += 1
                print(f"round {k} call {j}: {len(d)} of {n} outputs differ (first {d[:3]}, last {d[-2:]}), "
                      f"{int(np.sum(y[d] == -7.0))} never written", flush=True)
    finally:
        unreg(ring_in); unreg(ring_out)
    del ring_in, ring_out
    if variant in ("mmap", "byhandmmap"):
        try:
            del y
        except NameError:
            pass
        m1.close(); m2.close()
print(f"[{variant}] {rounds} registrations ({len(addrs)} distinct addresses), {bad} calls with mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
