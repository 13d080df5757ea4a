#!/usr/bin/env python3
"""Condense rocprofv3 --pmc passes (tools/pmc_run.sh) into per-kernel per-launch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "rr::"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pass*", "*counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if want not in k:
                continue
            name = k.split("(")[0].replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    print(f"== {name}")
    for c, v in sorted(ctrs.items()):
        print(f"   {c:28s} launches={len(v):3d}  avg/launch={sum(v)/len(v):.6g}")
