#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats summary of the named bench workloads.
# Usage: bash tools/prof_some.sh <tag> <workload>...   -> gpurun_out/prof_<tag>/<workload>_kernel_stats.txt
TAG=${1:-x}; shift
OUT=gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
for w in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw_$w" -o p -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-others --no-cpu --no-dropin $BENCH_EXTRA > "$OUT/bench_$w.json" 2> "$OUT/$w.log"
  f=$(ls $OUT/raw_$w/*kernel_stats.csv 2>/dev/null | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload $w --steps 10 --warmup 2 --no-others --no-cpu"; python3 tools/prof_summary.py "$f" 8; echo; echo "# bench line of the same run:"; tail -1 "$OUT/bench_$w.json"; } > "$OUT/${w}_kernel_stats.txt"
  head -4 "$OUT/${w}_kernel_stats.txt"
done
