#!/bin/bash
# GPU box: the whole -m gpu suite, smoke, the default bench line and a 2-rank run of the multi-GPU path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2_full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 2> gpurun_out/r2_bench_g2.err | tail -1 > gpurun_out/r2_bench_g2.json
cat gpurun_out/r2_full_tests.log; tail -2 gpurun_out/r2_smoke.log; tail -c 600 gpurun_out/r2_bench.err; tail -c 400 gpurun_out/r2_bench_g2.err
