#!/usr/bin/env python3
"""GPU box: the fused FM chain with a LONG filter (examples/rtl_fm.rs: 2467 taps, 25:128) on each of its tiles — plain
4096-point (k_fm_chain<12>), 8192-point split (k_fm_chain_split<2>), 16384-point split (<4>) — per window size and source
type: us per work_dev() call, events around 20 back-to-back calls.  Feeds FftFilter::alt_wins (csrc/blocks.hpp) and
profiles/r05_rtl_fm_tiles.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr

s = torch.cuda.current_stream().cuda_stream
fs = 1.024e6
base = rr.low_pass_complex(fs, 100e3, 1e3)                    # 2467 taps
rng = np.random.default_rng(1)


def taps_of(L):
    if L <= len(base):
        return base[:L].copy()
    return (rng.uniform(-1, 1, L) / L).astype(np.complex64)


print(f"{'taps':>5s} {'src':>4s} {'samples':>9s} " + " ".join(f"{'F=' + str(1 << lg):>10s}" for lg in (12, 13, 14)) + f" {'default':>10s}   us per call")
for L in ((1500, 2467, 3330) if os.environ.get("CHAIN", "1") != "0" else ()):
    for u8 in (False, True):
        for n in (512_000, 2_000_000, 8_000_000, 24_000_000, 64_000_000):
            x = (torch.randint(0, 256, (2 * n,), device="cuda", dtype=torch.uint8) if u8 else torch.rand(2 * n, device="cuda") * 2 - 1)
            y = torch.empty(n // 4 + 4096, device="cuda")
            row = []
            for lg in (12, 13, 14, 0):
                try:
                    with rr.build_options(**({"fft_log2f": lg} if lg else {})):
                        b = (rr.FmChainU8 if u8 else rr.FmChain)(taps_of(L), 200000, 1024000, 1.0)
                except Exception:
                    row.append(float("nan")); continue
                nin = 2 * n if u8 else n
                for _ in range(3): b.work_dev(x.data_ptr(), nin, y.data_ptr(), n // 4, s)
                torch.cuda.synchronize()
                a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(20): b.work_dev(x.data_ptr(), nin, y.data_ptr(), n // 4, s)
                e.record(); torch.cuda.synchronize()
                row.append(a.elapsed_time(e) / 20 * 1e3)
            print(f"{L:5d} {'u8' if u8 else 'c32':>4s} {n:9d} " + " ".join(f"{v:10.1f}" for v in row))
            del x, y

# the FftFilter block alone on the same tiles (k_fftfilt_os<12> / k_fftfilt_split<2|4>): alt_wins serves both
print()
print(f"{'taps':>5s} {'blk':>4s} {'samples':>9s} " + " ".join(f"{'F=' + str(1 << lg):>10s}" for lg in (12, 13, 14)) + f" {'default':>10s}   us per call (FftFilter)")
for L in (1500, 2467, 3330):
    for n in (512_000, 2_000_000, 8_000_000, 24_000_000, 64_000_000):
        x = torch.rand(2 * n, device="cuda") * 2 - 1
        y = torch.empty(2 * n + 4096, device="cuda")
        row = []
        for lg in (12, 13, 14, 0):
            with rr.build_options(**({"fft_log2f": lg} if lg else {})):
                b = rr.FftFilter(taps_of(L))
            for _ in range(3): b.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
            torch.cuda.synchronize()
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): b.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
            e.record(); torch.cuda.synchronize()
            row.append(a.elapsed_time(e) / 20 * 1e3)
        print(f"{L:5d} {'fft':>4s} {n:9d} " + " ".join(f"{v:10.1f}" for v in row))
        del x, y
