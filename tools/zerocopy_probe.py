#!/usr/bin/env python3
"""GPU box: the drop-in path's 4,096,000-byte HOST windows — staged copies (rr_block_work: upload, kernels, download, one after
the other: DMA up and down do not overlap on this pool) against ZERO-COPY: the kernels read the page-locked input window and
write the page-locked output window over PCIe themselves (full duplex: reads and writes at the same time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
_rings = {}
def ring(which, like):
    a = _rings.get(which)
    if a is None:
        a = np.zeros((8 << 20) + 64, np.uint8); rr.host_register(a); _rings[which] = a
    off = (-a.ctypes.data) % 64
    return a[off:off + like.nbytes].view(like.dtype)
def per_call(fn, reps):
    """us per call, the better of two passes (the second block instance of a process meets a one-time ~40 ms stall of the
    runtime somewhere in its first passes — seen as 340-470 us averages that no later pass repeats)"""
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        dt = (time.perf_counter() - t0) / reps * 1e6
        best = dt if best is None else min(best, dt)
    return best
def bench(name, mk, in_dtype, n_in, out_dtype, cap, reps=200):
    xin = torch.empty(n_in, dtype=in_dtype).pin_memory()
    if in_dtype == torch.uint8: xin.random_(0, 255)
    else: xin.uniform_(-1, 1)
    yout = torch.empty(cap, dtype=out_dtype).pin_memory()
    xd = torch.empty(n_in, dtype=in_dtype, device="cuda"); yd = torch.empty(cap, dtype=out_dtype, device="cuda")
    es_in = {torch.uint8: 1, torch.float32: 4}[in_dtype]
    # elements as the block counts them
    def count(n, dt): return n
    res = {}
    blk = mk()
    xin_np, yout_np = xin.numpy(), yout.numpy()
    nin_elems = n_in if in_dtype == torch.uint8 else (n_in // 2 if blk.in_dtype == np.complex64 else n_in)
    cap_elems = cap // 2 if blk.out_dtype == np.complex64 else cap
    xv = xin_np.view(blk.in_dtype); yv = yout_np.view(blk.out_dtype)
    # (1) the product call, rr_block_work, on pageable windows (staged) and on windows of rr_host_register'd arrays (zero copy)
    xp, yp = xv.copy(), yv.copy()
    for _ in range(5): blk.work_into(xp, yp, cap_elems)
    res["rr_block_work pageable"] = per_call(lambda: blk.work_into(xp, yp, cap_elems), reps)
    xr, yr = ring("in", xv), ring("out", yv)              # (registered once per process: re-registered addresses are retired from zero-copy)
    xr[:] = xv
    blk = mk()
    for _ in range(5): blk.work_into(xr, yr, cap_elems)
    res["rr_block_work registered"] = per_call(lambda: blk.work_into(xr, yr, cap_elems), reps)
    # (2) the forms by hand (torch page-locked tensors as device windows): zero-copy both ways
    blk = mk()
    for _ in range(5): blk.work_dev(xin.data_ptr(), nin_elems, yout.data_ptr(), cap_elems, s); torch.cuda.synchronize()
    def f2(): blk.work_dev(xin.data_ptr(), nin_elems, yout.data_ptr(), cap_elems, s); torch.cuda.synchronize()
    res["zero-copy in+out"] = per_call(f2, reps)
    # (3) DMA in, kernel writes the host window
    blk = mk()
    for _ in range(5): xd.copy_(xin, non_blocking=True); blk.work_dev(xd.data_ptr(), nin_elems, yout.data_ptr(), cap_elems, s); torch.cuda.synchronize()
    def f3(): xd.copy_(xin, non_blocking=True); blk.work_dev(xd.data_ptr(), nin_elems, yout.data_ptr(), cap_elems, s); torch.cuda.synchronize()
    res["DMA in, zero-copy out"] = per_call(f3, reps)
    # (4) kernel reads the host window, DMA out
    blk = mk()
    for _ in range(5): blk.work_dev(xin.data_ptr(), nin_elems, yd.data_ptr(), cap_elems, s); yout.copy_(yd, non_blocking=True); torch.cuda.synchronize()
    def f4(): blk.work_dev(xin.data_ptr(), nin_elems, yd.data_ptr(), cap_elems, s); yout.copy_(yd, non_blocking=True); torch.cuda.synchronize()
    res["zero-copy in, DMA out"] = per_call(f4, reps)
    print(f"{name}: " + "  ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
taps = rr.low_pass_complex(10e6, 1e6, 60e3)
bench("FftFilter 401 taps, 512,000 Complex in / out", lambda: rr.FftFilter(taps), torch.float32, 2 * 512_000, torch.float32, 2 * 512_000)
t2 = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
bench("FmChainU8 463 taps 1:6, 4,096,000 bytes in", lambda: rr.FmChainU8(t2, 1, 6, 1.0), torch.uint8, 4_096_000, torch.float32, 1_024_000)
bench("FirFilter 127 taps, 512,000 Complex", lambda: rr.FirFilter(rr.low_pass_complex(10e6, 1e6, 190e3)), torch.float32, 2 * 512_000, torch.float32, 2 * 512_000)
bench("HilbertFir 65*255 /8, 1,024,000 f32 in", lambda: rr.HilbertFir(65, rr.low_pass_complex(100e6, 5e6, 943e3), 8), torch.float32, 1_024_000, torch.float32, 2 * 128_000)
