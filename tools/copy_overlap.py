#!/usr/bin/env python3
"""GPU box: do a 4 MB upload and a 4 MB download on two streams overlap on this machine?  (pinned torch tensors and
hipHostRegister'ed numpy arrays; also in four 1 MB pieces each)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
n = 1 << 20                                     # floats: 4 MB
d1, d2 = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def bench(f, k=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
for kind in ("pinned", "registered"):
    if kind == "pinned":
        h1, h2 = torch.empty(n).pin_memory(), torch.empty(n).pin_memory()
    else:
        a1, a2 = np.zeros(n, np.float32), np.zeros(n, np.float32)
        rr.host_register(a1); rr.host_register(a2)
        h1, h2 = torch.from_numpy(a1), torch.from_numpy(a2)
    def up():
        with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
        s1.synchronize()
    def down():
        with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
        s2.synchronize()
    def both():
        with torch.cuda.stream(s1): d1.copy_(h1, non_blocking=True)
        with torch.cuda.stream(s2): h2.copy_(d2, non_blocking=True)
        s1.synchronize(); s2.synchronize()
    def both4():
        q = n // 4
        for i in range(4):
            with torch.cuda.stream(s1): d1[i*q:(i+1)*q].copy_(h1[i*q:(i+1)*q], non_blocking=True)
            with torch.cuda.stream(s2): h2[i*q:(i+1)*q].copy_(d2[i*q:(i+1)*q], non_blocking=True)
        s1.synchronize(); s2.synchronize()
    print(f"{kind}: up {bench(up):.0f} us, down {bench(down):.0f} us, both at once {bench(both):.0f} us, both in 4 pieces {bench(both4):.0f} us")
