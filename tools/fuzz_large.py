#!/usr/bin/env python3
"""GPU box: tests/test_gpu_large_windows.py's random large-window trial on further seeds:
python tools/fuzz_large.py [trials=16] [first_seed=100]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import rustradio_amd as rr
from test_gpu_large_windows import large_window_trial

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 16
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fails = 0
for k in range(first, first + trials):
    try:
        what, worst = large_window_trial(rr, k)
        print(f"seed {k}: {what}, worst allowance used {worst:.3f}", flush=True)
    except AssertionError as e:
        fails += 1
        print(f"seed {k}: FAIL {e}", flush=True)
sys.exit(1 if fails else 0)
