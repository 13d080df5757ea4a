import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(2 * n, device="cuda") * 2 - 1
y = torch.empty(2 * (n + 1024), device="cuda")
s = torch.cuda.current_stream().cuda_stream
def t(blk, d):
    cap = n // d + 8
    for _ in range(2): blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): blk.work_dev(x.data_ptr(), n, y.data_ptr(), cap, s)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / 3
rng = np.random.default_rng(1)
for d in (3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
    row = []
    for Ls in (300, 448, 520, 600, 680, 760):
        L = d * Ls - 1
        taps = (rng.standard_normal(L) / L).astype(np.complex64)
        with rr.build_options(fir_poly=-1): o = t(rr.FirFilter(taps, deci=d), d)
        try:
            with rr.build_options(fir_poly=1): p = t(rr.FirFilter(taps, deci=d), d)
        except Exception: p = float("nan")
        row.append(f"Ls={Ls} L={L}: other {o:.3f} poly {p:.3f}")
    print(f"/{d}: " + " | ".join(row), flush=True)
