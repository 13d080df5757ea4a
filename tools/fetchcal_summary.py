#!/usr/bin/env python3
"""profiles/fetch_calibration.json from the two rocprofv3 --pmc passes of tools/micro/fetchcal.bin (tools/fetchcal.sh):
for every access shape, factor = known bytes / (counter [KB] x 1024).  tools/pmc_traffic.py multiplies FETCH_SIZE /
WRITE_SIZE of a workload by the factor of ITS access shape instead of the guide's x2 (calibrated for 16 B/lane only)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
known = {}
for line in open(os.path.join(out, "run.log")):
    m = re.match(r"KNOWN (\S.*\S) (\d+)$", line.strip())
    if m:
        known[m.group(1)] = int(m.group(2))
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            name = name.replace("HIP_vector_type<float, 4u>", "float4").replace("HIP_vector_type<float, 2u>", "float2")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for name, b in known.items():
    key = next((k for k in acc if k.replace(" ", "") == name.replace(" ", "")), None)
    if key is None:
        continue
    ctr = "WRITE_SIZE" if name.startswith("wr_") else "FETCH_SIZE"
    v = acc[key].get(ctr)
    if not v:
        continue
    kb = sum(v) / len(v)
    res[name] = {"counter": ctr, "known_bytes": b, "counter_kb": kb, "factor": round(b / (kb * 1024.0), 4), "launches": len(v)}
    print(f"{name:36s} {ctr:10s} known {b:12d} B   counter {kb:12.1f} KB   factor {res[name]['factor']:.4f}")
json.dump(res, open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
