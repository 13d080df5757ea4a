#!/bin/bash
# GPU box: what the driver runs at round end, on this tree: smoke(), then the N = 1 bench line (exit code, wall seconds, line size, key fields).
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
SECONDS=0; python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; echo "rc=$? wall ${SECONDS}s"
tail -1 gpurun_out/bench_final.json | wc -c
tail -1 gpurun_out/bench_final.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'], d['verified'], d['parity'])"
