#!/usr/bin/env python3
"""Per-launch durations of a workload's dominant kernel from a rocprofv3 --kernel-trace CSV, cut into bench.py's passes
(`roofline.frac` reproducible from profiles/ to 1 %).

bench.py --workload W --steps K launches the dominant kernel once per step: ... settle, warm-up, K timed steps, K steps
with the library's event brackets (the pass `roofline.avg_kernel_ms` comes from), K steps with one event pair per step.
So the LAST 3K launches of the trace are those three passes.  Usage: prof_launches.py <kernel_trace.csv> <kernel substr> <K>
[alg_bytes_per_launch]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[3])
alg = float(sys.argv[4]) if len(sys.argv) > 4 else None
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
name = rows[-1]["Kernel_Name"].split("(")[0].replace("void ", "")
print(f"# {name}: {len(dur)} launches in the trace; the last {3 * k} = bench.py's timed pass, event-bracket pass, per-step pass ({k} each)")
for label, seg in (("timed region", dur[-3 * k:-2 * k]), ("event-bracket pass (roofline.avg_kernel_ms)", dur[-2 * k:-k]), ("per-step event pass", dur[-k:])):
    m = sum(seg) / len(seg)
    line = f"{label}: mean {m:.2f} us, min {min(seg):.2f}, max {max(seg):.2f}"
    if alg:
        line += f"; algorithmic {alg:.6g} B / mean = {alg / m / 1e3:.1f} GB/s = {alg / m / 1e3 / 8000:.4f} of 8 TB/s"
    print(line)
    print("   us: " + " ".join(f"{d:.1f}" for d in seg))
all_m = sum(dur) / len(dur)
print(f"all {len(dur)} launches (what --stats averages, settle and start-up transient included): mean {all_m:.2f} us")
