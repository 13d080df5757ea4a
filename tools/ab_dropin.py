"""GPU: FftFilter on reference-sized registered host windows (bench.py's dropin_fftfilter) with and without one
rr_build_opts override, alternately on one box.  python tools/ab_dropin.py KEY=VALUE [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

rr = bench.rr
k, v = sys.argv[1].split("=")
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for _ in range(rounds):
    a = bench.dropin_host_windows("fftfilter", True)
    with rr.build_options(**{k: int(v)}):
        b = bench.dropin_host_windows("fftfilter", True)
    print(f"default {a:8.1f} Msamples/s = {512000 / a:6.1f} us per call     {k}={v} {b:8.1f} Msamples/s = {512000 / b:6.1f} us per call")
