"""GPU: bench.py's drop-in runs (rr_block_work on reference-sized registered host windows) with and without one rr_build_opts
override, alternately on one box.  python tools/ab_dropin.py KEY=VALUE [rounds] [fftfilter|rtl_fm]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

rr = bench.rr
k, v = sys.argv[1].split("=")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
kind = sys.argv[3] if len(sys.argv) > 3 else "fftfilter"
n_in = 4_096_000 // 8 if kind == "fftfilter" else 4_096_000 // 2
for _ in range(rounds):
    a = bench.dropin_host_windows(kind, True)
    with rr.build_options(**{k: int(v)}):
        b = bench.dropin_host_windows(kind, True)
    print(f"{kind}: default {a:8.1f} Msamples/s = {n_in / a:6.1f} us per call     {k}={v} {b:8.1f} Msamples/s = {n_in / b:6.1f} us per call")
