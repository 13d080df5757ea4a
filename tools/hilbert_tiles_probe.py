#!/usr/bin/env python3
"""GPU: Hilbert(ntaps) on the real-stream transform tiles (200 ... 3584 taps, k_fftfilt_real<.., HILB>) with the refold pass
behind it (k_hilbert_refold_nonfinite) — run under rocprofv3 --kernel-trace --stats for the two kernels' durations:
    rocprofv3 --kernel-trace --stats -d gpurun_out/hil -- python3 tools/hilbert_tiles_probe.py [ntaps] [samples]"""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import rustradio_amd as rr  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 301
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
blk = rr.Hilbert(L)
x = torch.rand(n, dtype=torch.float32, device="cuda") - 0.5
y = torch.empty(2 * n, dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(30):
    blk.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20):
    blk.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
b.record(); b.synchronize()
print(f"Hilbert({L}), {n} samples per call: {a.elapsed_time(b) / 20:.4f} ms per call")
