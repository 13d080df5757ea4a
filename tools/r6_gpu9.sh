for o in 0 10 11 12; do
  python3 bench.py --workload fir --steps 200 --warmup 5 --no-others --no-cpu --no-dropin --no-verify $( [ $o != 0 ] && echo --opt fft_log2f=$o ) --detail-out gpurun_out/x.json 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fir 1e6 fft_log2f=$o', d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_kernel_ms'])"
done
python3 - <<'P'
import json, bench_dropin
import rustradio_amd as rr
for lg in (0, 10, 11, 12):
    with rr.build_options(**({"fft_log2f": lg} if lg else {})):
        r = bench_dropin.devgraph_resident_source("fftfilter")
    print("fftfilter 401 taps 512k window, fft_log2f =", lg, r["us_per_call_wall"], r["kernel_us_per_call"], r["kernel_launches_per_round"])
P
