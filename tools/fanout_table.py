#!/usr/bin/env python3
"""DESIGN.md §7's fan-out prediction table, regenerated from a bench.py line (its `multi_gpu_prediction` object, which
bench.py computes from that run's MEASURED configs[3] step through multi.predict_fanout).

    python bench.py > line.json ; python tools/fanout_table.py line.json        (or a BENCH_rNN.json of the driver)"""
import json
import sys

d = json.load(open(sys.argv[1]))
d = d.get("parsed", d)
p = d["multi_gpu_prediction"]
print(f"measured configs[3] step: {p['measured_fm_multi_ms_per_step']} ms; tile = {p['tile_steps']} steps; "
      f"links {p['assumptions']['xgmi_link_gbs']} GB/s, {p['assumptions']['collective_latency_ms']} ms per collective\n")
print("| source | N | one broadcast: ms per tile / efficiency | scatter + all-gather: ms per tile / efficiency |")
print("|---|---|---|---|")
for key, name in (("complex_f32_source", "Complex<f32>"), ("u8_source", "u8 I/Q bytes")):
    for n in ("2", "4", "8"):
        e = p[key][n]
        print(f"| {name} ({p[key]['tile_bytes'] / 1e6:.1f} MB per tile) | {n} | {e['bcast']['fanout_ms_per_tile']} / {e['bcast']['efficiency']} | "
              f"{e['scatter_allgather']['fanout_ms_per_tile']} / {e['scatter_allgather']['efficiency']} |")
