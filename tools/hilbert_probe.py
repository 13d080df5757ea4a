#!/usr/bin/env python3
"""GPU box: Hilbert GPU time per 1e8 real samples for a few transformer lengths (the pair-sample kernel up to ~199 taps,
the real-stream tiles beyond), device-resident windows.  ms per call (kernel time from HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rustradio_amd as rr
n = 100_000_000
x = torch.rand(n, device="cuda") * 2 - 1
y = torch.empty(2 * n, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for L in ([int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else (31, 65, 129, 301, 1001)):
    blk = rr.Hilbert(L)
    for _ in range(3): blk.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
    torch.cuda.synchronize(); blk.set_profiling(True)
    for _ in range(10): blk.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
    torch.cuda.synchronize()
    ms, k = blk.profile()
    print(f"Hilbert {L:5d} taps: {ms / k:.4f} ms per 1e8 samples", flush=True)
