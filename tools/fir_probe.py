#!/usr/bin/env python3
"""GPU box: non-decimating FirFilter<Complex> kernel time at 1e8 samples (127 real taps = BASELINE configs[0] taps,
and 127 complex taps) on the direct-form kernel; `python tools/fir_probe.py <cfg>` forces tile shape 0..7 (rr_build_opts.fir_cfg)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * n, device="cuda")
taps = rr.low_pass_complex(10e6, 1e6, 190e3)
for name, t in (("127 real taps", taps), ("127 complex taps", (taps * np.exp(0.3j * np.arange(len(taps)))).astype(np.complex64))):
    with rr.build_options(fir_path="direct", **({"fir_cfg": int(sys.argv[1])} if len(sys.argv) > 1 else {})):
        f = rr.FirFilter(t)
    for _ in range(2):
        f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
    torch.cuda.synchronize()
    f.set_profiling(True)
    for _ in range(5):
        f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
    torch.cuda.synchronize()
    ms, k = f.profile()
    flop = (2 if "real" in name else 4) * 2 * len(t) * n
    print(f"{name}: {ms/k:.4f} ms  {16*n/(ms/k*1e-3)/1e12:.2f} TB/s alg  {flop/(ms/k*1e-3)/1e12:.1f} TFLOP/s")
