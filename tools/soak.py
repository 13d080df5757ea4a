#!/usr/bin/env python3
"""GPU box: soak run of the seeded fuzz trials of tests/test_gpu_fuzz.py with fresh seeds (offset by argv[1], count argv[2])
— the same oracle-vs-HIP comparisons, thousands of random shapes; prints every failing (test, seed)."""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import rustradio_amd as rr
import test_gpu_fuzz as F


class MP:
    def __init__(self): self.saved = {}
    def setenv(self, k, v):
        self.saved.setdefault(k, os.environ.get(k)); os.environ[k] = v
    def setattr(self, obj, name, value):
        self.saved.setdefault(("attr", id(obj), name), (obj, name, getattr(obj, name))); setattr(obj, name, value)
    def undo(self):
        for k, v in list(self.saved.items()):
            if isinstance(k, tuple) and k[0] == "attr":
                setattr(v[0], v[1], v[2]); del self.saved[k]
        for k, v in self.saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v
        self.saved = {}


off, cnt = int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 200
fails = 0
names = [n for n in dir(F) if n.startswith("test_fuzz_")] if len(sys.argv) <= 3 else sys.argv[3:]
for name in names:
    fn = getattr(F, name)
    takes_mp = "monkeypatch" in fn.__code__.co_varnames[:fn.__code__.co_argcount]
    for seed in range(off, off + cnt):
        mp = MP()
        try:
            if takes_mp: fn(rr, mp, seed)
            else: fn(rr, seed)
        except Exception:
            fails += 1
            print("FAIL", name, seed); traceback.print_exc(limit=2)
        finally:
            mp.undo()
    print(name, "done", flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)
