#!/bin/bash
# GPU box: the round's profile set -> gpurun_out/prof_<tag>/ (copied into profiles/ by tools/round_collect.py <tag> afterwards).
#   per workload: rocprofv3 --kernel-trace --stats summary + the bench line of the same run + the per-launch durations of the
#                 dominant kernel in bench.py's last three passes (tools/prof_launches.py: roofline.frac from profiles/ alone),
#                 FETCH_SIZE / WRITE_SIZE in separate --pmc passes (-> profiles/traffic.json, stamped with the source hash);
#   the default bench command profiled and unprofiled.     Usage: bash tools/round_profiles.sh <tag> [workload:kernel ...]
TAG=${1:-r06}; shift
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
PAIRS=${@:-"fftfilter:k_fftfilt_os fm_chain:k_fm_chain_poly fm_multi:k_fm_multi_poly channelizer:k_fftfilt_prune full_chain_fused:k_fm_chain_poly fir_fft_chain:k_fftfilt_os rtl_fm_chain:k_fm_chain_poly rtl_fm_example:k_fm_chain_split fir_1e8:k_fftfilt_os fir_float:k_fftfilt_real"}
K=20
for pair in $PAIRS; do
  w=${pair%%:*}; k=${pair##*:}
  # (counter passes first: the traced run below then reports roofline.traffic from the fresh profiles/traffic.json)
  T=$OUT/traffic_$w; mkdir -p $T; i=0
  for c in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$T/pass$i" -o p -- python3 bench.py --workload $w --steps 3 --warmup 1 --no-others --no-cpu --no-dropin --no-verify > "$T/pass$i.log" 2>&1
  done
  python3 tools/pmc_summary.py "$T" rr:: > "$OUT/${w}_traffic_pmc.txt"
  python3 tools/pmc_traffic.py "$OUT/${w}_traffic_pmc.txt" $w "$k"
  rm -rf $T
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -o p -- python3 bench.py --workload $w --steps $K --warmup 3 --no-others --no-cpu --no-dropin --no-verify > "$OUT/bench_$w.json" 2> "$OUT/$w.log"
  f=$(ls $OUT/$w/*kernel_stats.csv 2>/dev/null | head -1); t=$(ls $OUT/$w/*kernel_trace.csv 2>/dev/null | head -1)
  alg=$(python3 -c "import json,sys; print(json.loads(open('$OUT/bench_$w.json').read().strip().splitlines()[-1])['roofline']['alg_bytes_per_launch'])" 2>/dev/null)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps $K --warmup 3 --no-others --no-cpu --no-dropin --no-verify"; python3 tools/prof_summary.py "$f" 8; echo; python3 tools/prof_launches.py "$t" "$k" $K $alg; echo; echo "# bench line of the same run:"; tail -1 "$OUT/bench_$w.json"; } > "$OUT/${w}_kernel_stats.txt"
  rm -rf "$OUT/$w"
  head -3 "$OUT/${w}_kernel_stats.txt" | tail -1
done
cp profiles/traffic.json $OUT/traffic.json
