#!/usr/bin/env python3
"""GPU box: host->HBM ring push rate (rr_dstream_copy_in), pageable vs page-locked (rr_host_register) source."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustradio_amd as rr
x = np.ones(64 * 512_000, np.complex64)
s = rr.DeviceStream(np.complex64)
def run(tag):
    for rep in range(2):
        t0 = time.perf_counter(); pos = 0
        while pos < len(x):
            n = s.push(x[pos:pos + 512_000]); pos += n
            s.pop(1); lib_consume = rr.lib().rr_dstream_consume(s._h, s.readable())
        dt = time.perf_counter() - t0
    print(f"{tag}: {x.nbytes / dt / 1e9:.1f} GB/s  ({dt / 64 * 1e6:.0f} us per 4 MB window)")
run("pageable")
rr.host_register(x)
run("registered")
rr.host_unregister(x)
