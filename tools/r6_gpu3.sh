python3 -m pytest tests/test_gpu_edges_fullsize.py -q -x -p no:cacheprovider -k "nonfinite or one_kernel" 2>&1 | tail -2
for rep in 1 2; do
for o in 0 2 1; do
  python3 bench.py --workload fftfilter --steps 30 --warmup 3 --no-others --no-cpu --no-dropin --no-verify --opt fft_nonfinite_tiles=$o --detail-out gpurun_out/ab_$o.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AB fft_nonfinite_tiles=$o', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_kernel_ms'])"
done; done
python3 - <<'P'
import json, bench_dropin
import rustradio_amd as rr
for o in (0, 2, 1):
    with rr.build_options(fft_nonfinite_tiles=o):
        print("fftfilter 512k, fft_nonfinite_tiles =", o, json.dumps(bench_dropin.devgraph_resident_source("fftfilter")))
P
