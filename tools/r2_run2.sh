#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg4.py tests/test_gpu_fuzz.py tests/test_multi_gloo.py -x -q -m gpu -k "fm_ or cfg4 or fuzz or two_ranks" 2>&1 | tail -25 > gpurun_out/r2_t2.log
tail -12 gpurun_out/r2_t2.log
for w in fm_chain fm_multi rtl_fm_chain full_chain_fused; do
  python bench.py --workload $w --no-others --no-cpu --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w', d['ms_per_step'], d['ms_per_step_median'], r['avg_kernel_ms'], r.get('achieved'), r.get('frac'), r.get('hbm_frac'))"
  python bench.py --workload $w --no-others --no-cpu --steps 20 --opt fm_poly=-1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w nopoly', d['ms_per_step'], d['ms_per_step_median'], r['avg_kernel_ms'], r.get('achieved'), r.get('frac'), r.get('hbm_frac'))"
done 2>&1 | tee gpurun_out/r2_poly_bench.log
