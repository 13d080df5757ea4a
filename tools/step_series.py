#!/usr/bin/env python3
"""GPU box: per-step kernel time of the headline workload over a long back-to-back run, from a cold start — clock ramp-up
and power-cap behaviour (which steps are the slow ones?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
w = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "fftfilter"](dev, 0, 1, lambda gen, numel, dtype: gen())
stream = torch.cuda.current_stream(); cs = stream.cuda_stream; w.dom_units = 0
torch.cuda.synchronize(); time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 2.0)     # let the GPU go idle
K = int(sys.argv[3]) if len(sys.argv) > 3 else 120
evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
evs[0].record(stream)
for i in range(K):
    w.step(cs); evs[i + 1].record(stream)
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(K)]
print("first 12:", " ".join(f"{x:.3f}" for x in ms[:12]))
for a in range(0, K, max(20, K // 12)):
    seg = ms[a:a + max(20, K // 12)]; print(f"steps {a:4d}+: mean {sum(seg) / len(seg):.4f}  min {min(seg):.4f}  max {max(seg):.4f}")
