#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/pmc_run.sh / tools/round_profiles.sh.
HBM bytes per launch = FETCH_SIZE[KB] * 1024 * f_read + WRITE_SIZE[KB] * 1024 * f_write with the factors CALIBRATED on this
part for the tile kernels' own access shapes (profiles/fetch_calibration.json, tools/fetchcal.sh: known byte counts read
2 / 4 / 8 / 16 B per lane, lane-consecutive and tile-strided -> f_read = 2.000 for every width, f_write = 1.000; the
guide's x2, MI355X_MICROARCH.md §HBM, was stated for 16 B/lane only)."""
import hashlib
import json
import os
import re
import sys


def sources_hash():
    """the same hash bench.py computes: traffic is only valid for the kernel sources it was collected on"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    d = os.path.join(root, "rustradio_amd", "csrc")
    for f in sorted(os.listdir(d)):
        # kernels, their headers and the block logic that picks between them (not the ABI / fan-out / ring plumbing)
        if (f.endswith((".hip", ".hpp")) and f != "dstream.hpp") or f == "blocks.cpp":
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def factors():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cal = json.load(open(os.path.join(root, "profiles", "fetch_calibration.json")))
    return cal["rd_tiles<float2,64>"]["factor"], cal["wr_tiles<float2,128>"]["factor"]


summary, workload, kernel = sys.argv[1], sys.argv[2], sys.argv[3]
f_read, f_write = factors()
vals, cur = {}, None
for line in open(summary):
    if line.startswith("=="):
        cur = line[2:].strip()
    m = re.match(r"\s+(\w+)\s+launches=\s*(\d+)\s+avg/launch=([\d.e+]+)", line)
    if m and cur and kernel in cur:
        vals[m.group(1)] = float(m.group(3))
out_path = "profiles/traffic.json"
try:
    d = json.load(open(out_path))
except Exception:
    d = {}
d[workload] = {"hbm_bytes_per_launch": vals["FETCH_SIZE"] * 1024 * f_read + vals["WRITE_SIZE"] * 1024 * f_write,
               "fetch_factor": f_read, "write_factor": f_write,
               "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb_raw": vals["WRITE_SIZE"],
               "kernel": kernel, "sources_sha16": sources_hash(),
               "note": "factors calibrated on known byte counts in the kernels' access shapes (profiles/fetch_calibration.json); separate --pmc passes"}
json.dump(d, open(out_path, "w"), indent=1)
print(d[workload])
