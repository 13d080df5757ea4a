#!/usr/bin/env python3
"""After tools/round_all.sh <tag> ran on the GPU box: copy its summaries from gpurun_out/prof_<tag>/ into profiles/
(tracked) as <tag>_*.  Usage: python tools/round_collect.py <tag>"""
import os, shutil, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
n = 0
for f in sorted(os.listdir(src)):
    if f.endswith(("_kernel_stats.txt", "_traffic_pmc.txt", "_stall_counters.txt")) or f in ("bench_default.json", "bench_detail.json", "pcie_inplace.txt", "rtl_fm_tiles.txt", "clocks.txt", "rotor_rate.txt", "ab_nonfinite_pass.txt"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}")); n += 1
    if f in ("traffic.json", "parity_allowance.json"):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f)); n += 1
print(n, "files copied into profiles/")
