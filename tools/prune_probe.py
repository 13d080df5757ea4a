#!/usr/bin/env python3
"""GPU box: decimating filters with deci 4 / 8 / 16 at 1e8 input samples: direct form, overlap-save tiles with a
decimating store, and tiles with the pruned inverse transform (k_fftfilt_prune); FirFilter<Complex> and the fused
Hilbert -> FirFilter (real input)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * n, device="cuda")
rng = np.random.default_rng(1)
OPTS = {}
def run(make, nin):
    with rr.build_options(**OPTS):
        f = make()
    for _ in range(2):
        f.work_dev(x.data_ptr(), nin, y.data_ptr(), n)
    torch.cuda.synchronize()
    f.set_profiling(True)
    for _ in range(4):
        f.work_dev(x.data_ptr(), nin, y.data_ptr(), n)
    torch.cuda.synchronize()
    ms, k = f.profile()
    return ms / k
def setenv(**kw):
    OPTS.clear(); OPTS.update(kw)
for L, d in ((32, 4), (64, 4), (127, 4), (401, 4), (64, 8), (127, 8), (255, 8), (401, 8), (1000, 8), (255, 16), (401, 16), (1000, 16), (2000, 16)):
    for cplx in (False, True):
        t = rng.uniform(-1, 1, L) + (1j * rng.uniform(-1, 1, L) if cplx else 0)
        t = (t / L).astype(np.complex64)
        row = []
        for env in ({"fir_path": "direct"}, {"fir_path": "fft", "fir_prune": -1, "fir_poly": -1}, {"fir_prune": 1}, {}):
            setenv(**env)
            row.append(run(lambda: rr.FirFilter(t, deci=d), n))
        print(f"FirFilter L={L:5d} d={d:3d} {'complex' if cplx else 'real   '} taps: direct {row[0]:.4f}  deci-store {row[1]:.4f}  pruned {row[2]:.4f}  auto {row[3]:.4f} ms", flush=True)
taps = rr.low_pass_complex(100e6, 5e6, 943e3)
for hn, tp, d in ((65, taps, 8), (65, taps, 16), (65, taps, 4), (129, rr.low_pass_complex(100e6, 2e6, 400e3), 16)):
    row = []
    for env in ({"fir_prune": -1}, {"fir_prune": 1}):
        setenv(**env)
        row.append(run(lambda: rr.HilbertFir(hn, tp, d), n))
    print(f"HilbertFir hn={hn} L={len(tp)} d={d}: direct {row[0]:.4f}  pruned {row[1]:.4f} ms per 1e8 real samples", flush=True)
