#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv into a short table."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    n = r["Name"].split("(")[0].replace("void ", "")[:70]
    print(f"{n:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {r['Percentage']:>6s}")
