#!/usr/bin/env python3
"""GPU box: the seeded fuzz tests of tests/test_gpu_fuzz.py on seeds BEYOND the committed ranges (one-off deep runs after a
kernel change): python tools/fuzz_more.py [first_seed=100] [count=150].  Prints failures, exits non-zero on any."""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import pytest
import rustradio_amd as rr
import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 150
fails = 0
mp = pytest.MonkeyPatch()
for name in ("test_fuzz_fir", "test_fuzz_fftfilter_and_chain", "test_fuzz_streaming_blocks", "test_fuzz_hilbert_fir",
             "test_fuzz_translate_and_u8_chain", "test_fuzz_fir_every_path", "test_fuzz_hilbert_fir_and_even_ratio_chains",
             "test_fuzz_fftstream_float_filters_and_multi", "test_fuzz_compositions_and_fastfm"):
    fn = getattr(F, name)
    fn = getattr(fn, "__wrapped__", fn)
    takes_mp = "monkeypatch" in fn.__code__.co_varnames[:fn.__code__.co_argcount]
    ok = 0
    for seed in range(first, first + count):
        try:
            if takes_mp:
                fn(rr, mp, seed)
                mp.undo()
            else:
                fn(rr, seed)
            ok += 1
        except Exception:
            fails += 1
            print(f"FAIL {name} seed {seed}"); traceback.print_exc(limit=3)
            mp.undo()
    print(f"{name}: {ok} / {count} seeds passed", flush=True)
sys.exit(1 if fails else 0)
