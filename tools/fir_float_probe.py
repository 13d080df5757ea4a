#!/usr/bin/env python3
"""GPU box: FirFilter<Float> and FftFilterFloat at 1e8 real samples: direct-form kernel vs real-stream tiles
(k_fftfilt_real), and FftFilterFloat's real inner filter vs the f32 -> Complex -> FftFilter -> .re path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.empty(n, device="cuda")
rng = np.random.default_rng(1)
def run(f, k=4):
    for _ in range(2):
        f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
    torch.cuda.synchronize()
    f.set_profiling(True)
    for _ in range(k):
        f.work_dev(x.data_ptr(), n, y.data_ptr(), n)
    torch.cuda.synchronize()
    ms, c = f.profile()
    return ms / c
for L, d in ((65, 1), (127, 1), (463, 1), (463, 6), (64, 4), (255, 4), (127, 8), (255, 8), (1000, 8), (255, 16), (1000, 16)):
    t = (rng.uniform(-1, 1, L) / L).astype(np.float32)
    row = []
    for opts in ({"fir_path": "direct"}, {"fir_path": "fft", "fir_prune": -1}, {"fir_prune": 1}, {}):
        with rr.build_options(**opts):
            row.append(run(rr.FirFilter(t, deci=d)))
    print(f"FirFilter<Float> L={L:5d} d={d:3d}: direct {row[0]:.4f}  tiles {row[1]:.4f}  pruned {row[2]:.4f}  auto {row[3]:.4f} ms", flush=True)
# FftFilterFloat: inner streams are 512,000 samples, so time whole-stream throughput over ring-sized windows
m = 512_000
for L in (127, 401, 2467):
    t = (rng.uniform(-1, 1, L) / L).astype(np.float32)
    for env in (None, "fftfloat_complex"):
        with rr.build_options(**({env: 1} if env else {})):
            f = rr.FftFilterFloat(t)
        tot = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            st, c, p, need = f.work_dev(x.data_ptr() + 4 * i * m, m, y.data_ptr(), m)
            tot += c
        f.sync(); dt = time.perf_counter() - t0
        print(f"FftFilterFloat L={L:5d} {'complex inner' if env else 'real inner   '}: {tot / dt / 1e6:.0f} Msamples/s over 512k-sample windows", flush=True)
