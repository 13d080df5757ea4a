#!/usr/bin/env python3
"""GPU box: what the timing instrumentation of bench.py's loop costs per step (fftfilter and fm_chain workloads):
bare loop / block profiling events / + per-step events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rustradio_amd as rr
dev = torch.device("cuda", 0)
for name in ("fftfilter", "fm_chain"):
    w = bench.WORKLOADS[name](dev, 0, 1, lambda gen, numel, dtype: gen())
    stream = torch.cuda.current_stream(); cs = stream.cuda_stream
    def run(prof, per_step, K=40):
        w.blocks[w.dominant].set_profiling(prof); w.dom_units = 0
        for _ in range(3): w.step(cs)
        torch.cuda.synchronize()
        if prof: w.blocks[w.dominant].profile(reset=True)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        t0 = time.perf_counter()
        for i in range(K):
            if per_step: evs[i][0].record(stream)
            w.step(cs)
            if per_step: evs[i][1].record(stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        km = w.blocks[w.dominant].profile(reset=True) if prof else (0, 1)
        w.blocks[w.dominant].set_profiling(False)
        return dt, km[0] / max(km[1], 1)
    for rep in range(2):
        print(name, "bare %.4f" % run(False, False)[0], "| prof %.4f (kernel %.4f)" % run(True, False), "| prof+step %.4f (kernel %.4f)" % run(True, True),
              "| step only %.4f" % run(False, True)[0])
