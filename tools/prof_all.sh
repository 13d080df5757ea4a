#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats summary of every bench workload + the headline bench line.
# Usage: bash tools/prof_all.sh <tag>   -> gpurun_out/prof_<tag>/<workload>_kernel_stats.txt, bench_<workload>.json
TAG=${1:-x}
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for w in fftfilter fir fir_fft_chain fm_chain rtl_fm_chain rtl_fm_example fm_multi channelizer channelizer_unfused; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$w" -o p -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-others --no-cpu > "$OUT/bench_$w.json" 2> "$OUT/$w.log"
  f=$(ls $OUT/$w/*kernel_stats.csv 2>/dev/null | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-others --no-cpu"; python3 tools/prof_summary.py "$f" 8; echo; echo "# bench line of the same run:"; tail -1 "$OUT/bench_$w.json"; } > "$OUT/${w}_kernel_stats.txt"
  rm -rf "$OUT/$w"
done
python3 bench.py --steps 20 --warmup 3 > "$OUT/bench_default.json" 2> "$OUT/bench_default.log"
tail -1 "$OUT/bench_default.json"
