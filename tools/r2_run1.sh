#!/bin/bash
# round 2, GPU call 1: new parity tests, the drop-in probe, a first bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests/test_gpu_cfg4.py tests/test_gpu_parity.py -x -q -m gpu -k "cfg4 or fir_fftfilter or fir_fm_chain or fm_multi or fm_chain_fused_block or every_tile_shape or decimating_both" 2>&1 | tail -15 > gpurun_out/r2_t1.log
python tools/dropin_probe.py > gpurun_out/r2_dropin.log 2>&1
timeout 900 python bench.py > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/r2_bench_g2.json 2> gpurun_out/r2_bench_g2.err
tail -5 gpurun_out/r2_t1.log; cat gpurun_out/r2_dropin.log; tail -c 1500 gpurun_out/r2_bench1.err; tail -c 800 gpurun_out/r2_bench_g2.err; head -c 600 gpurun_out/r2_bench_g2.json
