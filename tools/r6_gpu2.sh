set -x
python3 -m pytest tests/test_gpu_edges_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -q -x -p no:cacheprovider > gpurun_out/r6_t2.log 2>&1; echo "rc=$?" >> gpurun_out/r6_t2.log
tail -5 gpurun_out/r6_t2.log
for rep in 1 2; do
for o in 0 2 1; do
  python3 bench.py --workload fftfilter --steps 30 --warmup 3 --no-others --no-cpu --no-dropin --no-verify --opt fft_nonfinite_tiles=$o --detail-out gpurun_out/ab_$o.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB fft_nonfinite_tiles=$o', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_kernel_ms'])"
done; done
python3 - <<'P'
import json, bench_dropin
for k in ("fftfilter", "fm_chain_3", "fm_chain_fused"):
    print(k, json.dumps(bench_dropin.devgraph_resident_source(k)))
import rustradio_amd as rr
with rr.build_options(fft_nonfinite_tiles=2):
    print("fftfilter, separate pass", json.dumps(bench_dropin.devgraph_resident_source("fftfilter")))
P
