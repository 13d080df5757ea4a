#!/usr/bin/env python3
"""GPU box: create / use / destroy every block type a few hundred times and compare free device memory before and after."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import rustradio_amd as rr
from rustradio_amd import multi
lp = rr.low_pass_complex
x = (np.random.default_rng(0).uniform(-1, 1, 40000) + 0j).astype(np.complex64)
xf = x.real.copy()
xb = np.random.default_rng(1).integers(0, 256, 80000).astype(np.uint8)
makers = {
    "FirFilter": lambda: (rr.FirFilter(lp(10e6, 1e6, 190e3), deci=3), x),
    "FirFilter/8": lambda: (rr.FirFilter(lp(100e6, 5e6, 943e3), deci=8), x),
    "FirFloat": lambda: (rr.FirFilter(lp(10e6, 1e6, 190e3).real.astype(np.float32), deci=4), xf),
    "FftFilter": lambda: (rr.FftFilter(lp(10e6, 1e6, 60e3)), x),
    "FftFilter20000": lambda: (rr.FftFilter((np.ones(20000) / 20000).astype(np.complex64)), x),
    "FftFilterFloat": lambda: (rr.FftFilterFloat(rr.low_pass(200e3, 15e3, 5e3)), xf),
    "FftStream3000": lambda: (rr.FftStream(3000), x),
    "FftStream32768": lambda: (rr.FftStream(32768), x),
    "FmChain": lambda: (rr.FmChain(lp(2.4e6, 100e3, 12.5e3), 1, 6), x),
    "FmChainU8": lambda: (rr.FmChainU8(lp(2.4e6, 100e3, 12.5e3), 1, 6), xb),
    "FmMulti": lambda: (rr.FmMulti(multi.cfg4_taps(lp(2.4e6, 100e3, 12.5e3), range(4)), 1, 6), x),
    "HilbertFir": lambda: (rr.HilbertFir(65, lp(100e6, 5e6, 943e3), 8), xf),
    "AudioChain": lambda: (rr.AudioChain(rr.low_pass(200e3, 15e3, 5e3), 6, 25, 0.5), xf),
    "Hilbert": lambda: (rr.Hilbert(65), xf),
    "Resampler": lambda: (rr.RationalResampler(3, 7), x),
}
torch.cuda.init(); torch.cuda.synchronize()
for name, mk in makers.items():
    for _ in range(3):
        b, inp = mk(); b.work(inp, 200000); del b
    gc.collect(); torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(150):
        b, inp = mk(); b.work(inp, 200000); del b
    gc.collect(); torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    print(f"{name:16s} free memory change after 150 create/work/destroy: {(free1 - free0) / 1e6:+.2f} MB")
f0 = torch.cuda.mem_get_info()[0]
for _ in range(100):
    fan = rr.Fanout(None, 0, 1, 1 << 20); fan.produce_buf(0); fan.submit(0); fan.acquire(0); fan.release(0); del fan
    ds = rr.DeviceStream(np.complex64); ds.push(x); ds.pop(); del ds
gc.collect(); torch.cuda.synchronize()
print(f"{'Fanout+DeviceStream':16s} free memory change after 100: {(torch.cuda.mem_get_info()[0] - f0) / 1e6:+.2f} MB")
