#!/usr/bin/env python3
"""GPU: cost of FirFilter::translate's rotator per output, REPLAY (the default: the reference's f32 recurrence on one lane,
generated ahead on a side stream) against the opt-in f64 MODEL, and how much of REPLAY the look-ahead hides when the
caller paces its calls.

    python tools/replay_rate.py [outputs_per_call]
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import rustradio_amd as rr  # noqa: E402


def run(n, rotator, calls=6, pause_s=0.0):
    one = np.ones(1, np.complex64)
    blk = rr.FirFilter(one, translate=(1e6, 123456.7), rotator=rotator)
    x = torch.ones(2 * n, dtype=torch.float32, device="cuda")
    y = torch.empty(2 * n, dtype=torch.float32, device="cuda")
    ts = torch.cuda.current_stream()
    s = ts.cuda_stream
    times = []
    torch.cuda.synchronize()
    for _ in range(calls):
        if pause_s:
            time.sleep(pause_s)            # the source's pace: the side stream keeps generating meanwhile
        t0 = time.perf_counter()
        blk.work_dev(x.data_ptr(), n, y.data_ptr(), n, s)
        ts.synchronize()                   # the COMPUTE stream only: the look-ahead keeps running on the block's side stream
        times.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    return times


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    for name, mode, calls in (("replay (default: adaptive)", rr.ROT_REPLAY, 30), ("replay_host (generator thread)", rr.ROT_REPLAY_HOST, 12),
                              ("replay_device (one lane)", rr.ROT_REPLAY_DEVICE, 12), ("model", rr.ROT_MODEL, 12)):
        t = run(n, mode, calls=calls)
        later = np.median(t[-8:])          # (the default starts on the device and moves to the host generator after three starved calls)
        print(f"{name:30s} back to back : first call {t[0] * 1e3:8.3f} ms, last 8 calls {later * 1e3:8.3f} ms per {n} outputs "
              f"= {later / n * 1e9:6.2f} ns/output")
    per = np.median(run(n, rr.ROT_REPLAY_DEVICE)[1:])
    t = run(n, rr.ROT_REPLAY, pause_s=1.3 * per)       # (never starves: stays on the device chain)
    print(f"replay  paced (caller idles 1.3x the chain time between calls): {np.median(t[1:]) * 1e3:8.3f} ms per call "
          f"(the chain ran ahead on the side stream)")


if __name__ == "__main__":
    main()
