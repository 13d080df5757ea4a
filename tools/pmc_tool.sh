#!/bin/bash
# GPU box: rocprofv3 --pmc passes over any python tool.  Usage: bash tools/pmc_tool.sh <outdir> <kernel-substr> <script.py> "<ctrs pass1>" ["<ctrs pass2>" ...]
OUT=$1; KSUB=$2; SCRIPT=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
i=1
for CTRS in "$@"; do
  rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pass$i" -o p -- python3 $SCRIPT > "$OUT/pass$i.log" 2>&1
  i=$((i+1))
done
python3 tools/pmc_summary.py "$OUT" "$KSUB"
