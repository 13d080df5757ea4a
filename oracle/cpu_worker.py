"""CPU-baseline worker (TEST / MEASUREMENT INFRASTRUCTURE, like everything under oracle/): times the oracle's blocks the
way the reference's single-threaded Graph runs them (src/graph.rs:113), on 4,096,000-byte rings (src/stream.rs:105).
Used by bench.py's cpu_baseline leg only — in-process for the 1-thread figure, and as `python oracle/cpu_worker.py
<npz> <seconds>` child processes (one chain per core) for the all-cores figure."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(_HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(_HERE))


def chain_for(kind, taps):
    from oracle import pyoracle as orc
    if kind == "channelizer":
        return [orc.Hilbert(65), orc.FirFilter(taps, deci=8)]
    if kind == "FirFilterFloat":
        return [orc.FirFilter(taps)]
    if kind == "fir_fft_chain":
        return [orc.FirFilter(taps[0]), orc.FftFilter(taps[1])]
    if kind == "full_chain":
        return [orc.FirFilter(taps[0]), orc.FftFilter(taps[1]), orc.RationalResampler(1, 4), orc.QuadratureDemod(1.0)]
    if kind == "rtl_fm_example":
        return [orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(200000, 1024000), orc.QuadratureDemod(1.0)]
    if kind == "rtl_fm_chain":
        return [orc.RtlSdrDecode(), orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)]
    return {"FftFilter": lambda: [orc.FftFilter(taps)],
            "FirFilter": lambda: [orc.FirFilter(taps)],
            "fm_chain": lambda: [orc.FftFilter(taps), orc.RationalResampler(1, 6), orc.QuadratureDemod(1.0)]}[kind]()


def graph_1thread(chain, host, win, in_mult, seconds):
    """Graph (src/graph.rs:113): every block's work() on ONE thread -> (samples fed, seconds)"""
    nwin = max(1, len(host) // win)
    rings = [np.zeros(0, b.in_dtype) for b in chain]
    t0 = time.perf_counter()
    fed = i = 0
    while time.perf_counter() - t0 < seconds:
        chunk = host[(i % nwin) * win:(i % nwin + 1) * win]
        i += 1
        rings[0] = np.concatenate([rings[0], chunk])
        fed += len(chunk) // in_mult
        for j, b in enumerate(chain):
            while True:
                st, c, p, need, out = b.work(rings[j], 4_096_000 // b.out_dtype.itemsize)
                rings[j] = rings[j][c:]
                if j + 1 < len(chain):
                    rings[j + 1] = np.concatenate([rings[j + 1], out])
                if st == 1 or (c == 0 and p == 0):      # WAIT_SRC, or no progress (the output is drained every call)
                    break
    return fed, time.perf_counter() - t0


if __name__ == "__main__":
    d = np.load(sys.argv[1], allow_pickle=False)
    kind = str(d["kind"])
    taps = (d["taps0"], d["taps1"]) if "taps1" in d else d["taps0"]
    fed, dt = graph_1thread(chain_for(kind, taps), d["host"], int(d["win"]), int(d["in_mult"]), float(sys.argv[2]))
    print(fed, dt)
