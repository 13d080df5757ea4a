python3 -m pytest tests/test_gpu_edges_fullsize.py -q -p no:cacheprovider -k "nan_sets or fused_fm or audio" 2>&1 | tail -3
for w in fm_chain fm_multi; do
for o in 0 1; do
  python3 bench.py --workload $w --steps 30 --warmup 3 --no-others --no-cpu --no-dropin --no-verify --opt fft_nonfinite_tiles=$o --detail-out gpurun_out/ab_$o.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB $w fft_nonfinite_tiles=$o', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_kernel_ms'])"
done; done
