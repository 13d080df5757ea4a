#!/usr/bin/env python3
"""GPU box: fixed cost of one work_dev() call per block type — 20 back-to-back calls on windows of 64k / 512k samples, events
around the batch (GPU time per call) and host time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
s = torch.cuda.current_stream().cuda_stream
lp = rr.low_pass_complex
blocks = [("FftFilter401", lambda: rr.FftFilter(lp(10e6, 1e6, 60e3)), 8, 8, 1),
          ("Fir127", lambda: rr.FirFilter(lp(10e6, 1e6, 190e3)), 8, 8, 1),
          ("Fir255/8", lambda: rr.FirFilter(lp(100e6, 5e6, 943e3), deci=8), 8, 8, 8),
          ("Fir401/6 poly", lambda: rr.FirFilter(lp(10e6, 1e6, 60e3), deci=6), 8, 8, 6),
          ("FmChain463 1:6", lambda: rr.FmChain(lp(2.4e6, 100e3, 12.5e3), 1, 6, 1.0), 8, 4, 6),
          ("HilbertFir65*255/8", lambda: rr.HilbertFir(65, lp(100e6, 5e6, 943e3), 8), 4, 8, 8),
          ("Fir255/8 no-prune", lambda: _opt(dict(fir_prune=-1), lambda: rr.FirFilter(lp(100e6, 5e6, 943e3), deci=8)), 8, 8, 8),
          ("Fir255/8 direct", lambda: _opt(dict(fir_path="direct"), lambda: rr.FirFilter(lp(100e6, 5e6, 943e3), deci=8)), 8, 8, 8),
          ("Fir401/6 no-poly", lambda: _opt(dict(fir_poly=-1), lambda: rr.FirFilter(lp(10e6, 1e6, 60e3), deci=6)), 8, 8, 6),
          ("Fir401/6 direct", lambda: _opt(dict(fir_path="direct"), lambda: rr.FirFilter(lp(10e6, 1e6, 60e3), deci=6)), 8, 8, 6),
          ("FmChain no-poly", lambda: _opt(dict(fm_poly=-1), lambda: rr.FmChain(lp(2.4e6, 100e3, 12.5e3), 1, 6, 1.0)), 8, 4, 6),
          ("FmChain full", lambda: _opt(dict(fm_poly=-1, fm_full=1), lambda: rr.FmChain(lp(2.4e6, 100e3, 12.5e3), 1, 6, 1.0)), 8, 4, 6),
          ("HilbertFir direct", lambda: _opt(dict(fir_path="direct"), lambda: rr.HilbertFir(65, lp(100e6, 5e6, 943e3), 8)), 4, 8, 8),
          ("HilbertFir no-prune", lambda: _opt(dict(fir_prune=-1), lambda: rr.HilbertFir(65, lp(100e6, 5e6, 943e3), 8)), 4, 8, 8),
          ("Fir1000/16", lambda: rr.FirFilter(lp(100e6, 2e6, 240e3)[:1000], deci=16), 8, 8, 16),
          ("Fir2000/5", lambda: rr.FirFilter(np.concatenate([lp(100e6, 2e6, 240e3), lp(100e6, 2e6, 240e3)])[:2000], deci=5), 8, 8, 5),
          ("Fir1000/4", lambda: rr.FirFilter(lp(100e6, 2e6, 240e3)[:1000], deci=4), 8, 8, 4),
          ("FirFloat255/8", lambda: rr.FirFilter(lp(100e6, 5e6, 943e3).real.astype(np.float32), deci=8), 4, 4, 8),
          ("FirFloat255/8 direct", lambda: _opt(dict(fir_path="direct"), lambda: rr.FirFilter(lp(100e6, 5e6, 943e3).real.astype(np.float32), deci=8)), 4, 4, 8),
          ("FirFloat255/8 no-prune", lambda: _opt(dict(fir_prune=-1), lambda: rr.FirFilter(lp(100e6, 5e6, 943e3).real.astype(np.float32), deci=8)), 4, 4, 8),
          ("FirFloat1000/4", lambda: rr.FirFilter(lp(100e6, 2e6, 240e3)[:1000].real.astype(np.float32), deci=4), 4, 4, 4),
          ("FirFloat1000/4 no-prune", lambda: _opt(dict(fir_prune=-1), lambda: rr.FirFilter(lp(100e6, 2e6, 240e3)[:1000].real.astype(np.float32), deci=4)), 4, 4, 4),
          ("FftFilterFloat963", lambda: rr.FftFilterFloat(rr.low_pass(200e3, 15e3, 5e3)), 4, 4, 1),
          ("AudioChain963 6:25", lambda: rr.AudioChain(rr.low_pass(200e3, 15e3, 5e3), 6, 25, 0.5), 4, 4, 4),
          ("FftStream1024", lambda: rr.FftStream(1024), 8, 8, 1),
          ("FftStream3000", lambda: rr.FftStream(3000), 8, 8, 1),
          ("FftStream65536", lambda: rr.FftStream(65536), 8, 8, 1),
          ("FftFilter2467", lambda: rr.FftFilter(lp(1.024e6, 100e3, 1e3)), 8, 8, 1),
          ("FftFilter2467 F=4096", lambda: _opt(dict(fft_log2f=12), lambda: rr.FftFilter(lp(1.024e6, 100e3, 1e3))), 8, 8, 1),
          ("FftFilter2467 F=16384", lambda: _opt(dict(fft_log2f=14), lambda: rr.FftFilter(lp(1.024e6, 100e3, 1e3))), 8, 8, 1),
          ("FmChain2467 25:128", lambda: rr.FmChain(lp(1.024e6, 100e3, 1e3), 200000, 1024000, 1.0), 8, 4, 5),
          ("Resampler1:6", lambda: rr.RationalResampler(1, 6), 8, 8, 6),
          ("QuadDemod", lambda: rr.QuadratureDemod(1.0), 8, 4, 1),
          ("Hilbert65", lambda: rr.Hilbert(65), 4, 8, 1)]
def _opt(o, f):
    with rr.build_options(**o):
        return f()


for n in (512_000, 2_000_000, 8_000_000):
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    y = torch.empty(2 * n + 4096, device="cuda")
    print()
    for name, mk, ies, oes, d in blocks:
        b = mk(); cap = n // d + 16
        for _ in range(5): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); a.record()
        for _ in range(20): b.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
        e.record(); host = (time.perf_counter() - t0) / 20 * 1e6
        torch.cuda.synchronize()
        print(f"n={n:7d} {name:20s} gpu {a.elapsed_time(e) / 20 * 1e3:6.1f} us/call  host {host:5.1f} us/call")
