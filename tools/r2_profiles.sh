#!/bin/bash
# GPU box: the round's profile set -> gpurun_out/prof_<tag>/ (copied into profiles/ by tools/r2_collect.py afterwards).
#   rocprofv3 --kernel-trace --stats of the default bench command and of the main workloads,
#   FETCH_SIZE / WRITE_SIZE in separate --pmc passes for the dominant kernels (-> profiles/traffic.json).
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
# the default command, as the driver runs it
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/default" -o p -- python3 bench.py --no-cpu --no-dropin > "$OUT/bench_default_profiled.json" 2> "$OUT/default.log"
f=$(ls $OUT/default/*kernel_stats.csv 2>/dev/null | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-dropin   (default workload + the short runs of the others)"; python3 tools/prof_summary.py "$f" 24; } > "$OUT/default_kernel_stats.txt"
rm -rf "$OUT/default"
for w in fftfilter fm_chain fm_multi channelizer full_chain_fused fir_fft_chain rtl_fm_chain; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -o p -- python3 bench.py --workload $w --steps 20 --warmup 3 --no-others --no-cpu > "$OUT/bench_$w.json" 2> "$OUT/$w.log"
  f=$(ls $OUT/$w/*kernel_stats.csv 2>/dev/null | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 20 --warmup 3 --no-others --no-cpu"; python3 tools/prof_summary.py "$f" 8; echo; echo "# bench line of the same run:"; tail -1 "$OUT/bench_$w.json"; } > "$OUT/${w}_kernel_stats.txt"
  rm -rf "$OUT/$w"
done
for pair in "fftfilter:k_fftfilt_os" "fm_chain:k_fm_chain_poly" "fm_multi:k_fm_multi_poly" "channelizer:k_fftfilt_prune" "full_chain_fused:k_fm_chain_poly" "fir_fft_chain:k_fftfilt_os"; do
  w=${pair%%:*}; k=${pair##*:}
  T=$OUT/traffic_$w; mkdir -p $T; i=0
  for c in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$T/pass$i" -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu > "$T/pass$i.log" 2>&1
  done
  python3 tools/pmc_summary.py "$T" rr:: > "$OUT/${w}_traffic_pmc.txt"
  python3 tools/pmc_traffic.py "$OUT/${w}_traffic_pmc.txt" $w "$k"
  rm -rf $T
done
cp profiles/traffic.json $OUT/traffic.json
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.log"
tail -c 400 "$OUT/bench_default.log"
