#!/usr/bin/env python3
"""GPU box, timing library (make -C rustradio_amd/csrc TIMING=1 OUT=../lib_timing, loaded through RR_LIB_PATH — the product
library is never overwritten): per-phase cycle counts of one FftFilter tile (wave 0 of workgroup 0, its 3rd tile).
Usage: RR_FFT_STAMPS=1 RR_LIB_PATH=$PWD/rustradio_amd/lib_timing/librustradio_amd.so python tools/stamps.py [log2f] [samples]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
try:
    os.environ["RR_FFT_STAMPS"] = "1"
    import torch, numpy as np
    import rustradio_amd as rr
    OPTS = {"fft_log2f": int(sys.argv[1])} if len(sys.argv) > 1 else {}
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    taps = rr.low_pass_complex(10e6, 1e6, 60e3)
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    y = torch.empty(2 * (n + 4096), device="cuda")
    with rr.build_options(**OPTS):
        b = rr.FftFilter(taps)
    for _ in range(3):
        b.work_dev(x.data_ptr(), n, y.data_ptr(), n + 4096, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    k = rr.lib().rr_debug_fft_stamps(st)
    s = list(st)
    names = ["global load (vmcnt0)", "fwd pass0 math", "store0+sync", "load1 (lgkm0)", "fwd pass1 math", "store1+sync+load2",
             "fwd2+H+inv2 math", "store2+sync+load1", "inv pass1 math", "store1+sync+load0", "inv pass0 math", "global stores issue"]
    print("stamps:", k, "total cycles for the tile:", s[12] - s[0])
    for i, nm in enumerate(names):
        print(f"  {nm:28s} {s[i+1]-s[i]:7d}")
finally:
    pass
