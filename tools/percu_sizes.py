#!/usr/bin/env python3
"""GPU box, measurement library (RR_LIB_PATH=rustradio_amd/lib_m/..., RR_FFT_PERCU read per first launch of a kernel, so one
process per setting): FftFilter / decimating FIR kernel time against input size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
res = []
for n in (2_000_000, 5_000_000, 10_000_000, 20_000_000):
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    y = torch.empty(2 * n + 4096, device="cuda")
    for name, blk, cap in (("fftfilter401", rr.FftFilter(rr.low_pass_complex(10e6, 1e6, 60e3)), n + 1024),
                           ("fir255/8", rr.FirFilter(rr.low_pass_complex(100e6, 5e6, 943e3), deci=8), n // 8 + 8),
                           ("fir127", rr.FirFilter(rr.low_pass_complex(10e6, 1e6, 190e3)), n)):
        for _ in range(3): blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        b.record(); torch.cuda.synchronize()
        res.append(f"{name}@{n//1000000}M {a.elapsed_time(b)/20*1e3:.1f}us")
print(os.environ.get("RR_FFT_PERCU", "default"), " ".join(res))
