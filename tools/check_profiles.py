#!/usr/bin/env python3
"""Here (no GPU): are the tracked measurement records of profiles/ stamped with THIS tree's kernel-source hash
(bench._sources_hash: csrc/*.hip, *.hpp, blocks.cpp)?  traffic.json's figures reach the bench line only then.
    python tools/check_profiles.py [tag]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
h = bench._sources_hash()
ok = True
for name, get in (("profiles/parity_allowance.json", lambda d: d.get("kernel_sources")),
                  (f"profiles/{tag}_bench_detail.json", lambda d: d["roofline"]["traffic_note"].rsplit(" ", 1)[-1])):
    got = get(json.loads(open(os.path.join(ROOT, name)).read().strip().splitlines()[-1] if name.endswith("default.json") else open(os.path.join(ROOT, name)).read()))
    print(f"{name}: {got} {'==' if got == h else '!='} tree {h}")
    ok &= got == h
t, note = bench.measured_traffic("fftfilter")
print("traffic.json as bench.py reads it for the headline:", t, "|", note)
ok &= t is not None
sys.exit(0 if ok else 1)
