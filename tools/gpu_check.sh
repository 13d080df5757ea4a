#!/bin/bash
# GPU box: parity tests + a short bench line (condensed).  Usage: bash tools/gpu_check.sh [bench args]
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --steps 20 --warmup 3 --no-cpu "$@" 2>&1 | tail -1 | python3 -c '
import sys, json
d = json.loads(sys.stdin.read())
r = d["roofline"]
print("value %.0f Msps  kernel %.4f ms  achieved %.0f GB/s  frac %.3f" % (d["value"], r["avg_kernel_ms"], r["achieved"], r["frac"]))
for k, v in d.get("others", {}).items():
    print("  %-12s %9.0f Msps  step %.4f ms  dom-kernel %s GB/s" % (k, v["msamples_per_s"], v["ms_per_step"], v["dominant_kernel_alg_gbs"]))
'
