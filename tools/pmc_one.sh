#!/bin/bash
# GPU box: one rocprofv3 --pmc pass over bench.py.  Usage: bash tools/pmc_one.sh <outdir> "<counters>" <kernel-substr> [bench args]
OUT=$1; CTRS=$2; KSUB=${3:-rr::}; shift 3
ARGS=${@:---steps 3 --warmup 1 --no-others --no-cpu}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pass1" -o p -- python3 bench.py $ARGS > "$OUT/pass1.log" 2>&1
python3 tools/pmc_summary.py "$OUT" "$KSUB"
