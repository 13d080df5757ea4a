import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rustradio_amd as rr
taps = rr.low_pass_complex(2.4e6, 100e3, 12.5e3)
rng = np.random.default_rng(7)
x = rng.integers(0, 256, 4_096_000, dtype=np.uint8); out = np.zeros(1_024_000, np.float32)
rr.host_register(x); rr.host_register(out)
T0 = time.perf_counter()
for inst in range(4):
    blk = rr.FmChainU8(taps, 1, 6, 1.0)
    ts = []
    for i in range(300):
        t0 = time.perf_counter(); blk.work_into(x, out, len(out)); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    big = np.argsort(ts)[-4:][::-1]
    print(f"instance {inst} at {time.perf_counter() - T0:6.3f}s: median {np.median(ts):.1f} us, sum {ts.sum() / 1e3:.1f} ms, largest calls:", [(int(i), round(float(ts[i]))) for i in big], flush=True)
    # is it a slow stretch rather than one call?
    slow = np.where(ts > 2 * np.median(ts))[0]
    if len(slow): print("   calls above 2x median:", len(slow), "first", int(slow[0]), "last", int(slow[-1]))
    del blk
