#!/usr/bin/env python3
"""GPU box: sample rocm-smi clocks/power while the FftFilter kernel (or packed-FMA FIR) runs back to back."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr

which = sys.argv[1] if len(sys.argv) > 1 else "fft"           # fft | fir | fm_chain | fm_multi | channelizer
n = 100_000_000
cap = n + 2048
if which in ("fm_chain", "fm_multi"):
    from rustradio_amd import multi
    n = 24_000_000 if which == "fm_chain" else 2_400_000
    taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    cap = n // 6 + 1024
    if which == "fm_chain":
        blk, y = rr.FmChain(taps, 1, 6, 1.0), torch.empty(cap, device="cuda")
    else:
        blk, y = rr.FmMulti(multi.cfg4_taps(taps, range(32)), 1, 6, 1.0), torch.empty(32 * cap, device="cuda")
elif which == "channelizer":
    x = torch.rand(n, device="cuda") * 2 - 1
    cap = n // 8 + 8
    blk, y = rr.HilbertFir(65, rr.low_pass_complex(100e6, 5e6, 943e3), 8), torch.empty(2 * cap, device="cuda")
else:
    x = torch.rand(2 * n, device="cuda") * 2 - 1
    y = torch.empty(2 * n + 4096, device="cuda")
    blk = rr.FftFilter(rr.low_pass_complex(10e6, 1e6, 60e3)) if which == "fft" else rr.FirFilter(rr.low_pass_complex(10e6, 1e6, 190e3))
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True, timeout=10).stdout
            keep = [l.strip() for l in out.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power", "GPU use"))]
            samples.append(" | ".join(keep))
        except Exception as e:
            samples.append(f"rocm-smi failed: {e}")
            return
th = threading.Thread(target=poll); th.start()
t0 = time.time()
calls = 0
while time.time() - t0 < 6:
    for _ in range(200):
        blk2 = blk
        blk2.work_dev(x.data_ptr(), n, y.data_ptr(), cap)
    torch.cuda.synchronize()
    calls += 200
wall = time.time() - t0
stop = True; th.join()
print(f"{which}: {wall / calls * 1e3:.4f} ms per call over {calls} back-to-back calls (lib: {os.environ.get('RR_LIB_PATH', 'product')})")
print(f"{which}: idle-first then loaded samples")
for s in samples[:2] + samples[-4:]:
    print("  ", s[:300])
