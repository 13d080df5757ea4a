#!/bin/bash
# GPU box: tools/r5_all.sh, but only on a box whose headline kernel runs at the pool's usual speed (boxes differ by up to 5 %:
# profiles/r05_* of one tree were 0.3245 and 0.3311 ms on two of them).  bash tools/r5_all_if_fast.sh <tag> [max kernel ms]
TAG=${1:-r05}; LIM=${2:-0.3275}
ms=$(python3 bench.py --steps 20 --warmup 3 --no-others --no-cpu --no-dropin 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['roofline']['avg_kernel_ms'])")
echo "headline kernel on this box: $ms ms (limit $LIM)"
python3 -c "import sys; sys.exit(0 if float('$ms') <= float('$LIM') else 3)" || { echo "slow box: not collecting"; exit 3; }
bash tools/r5_all.sh $TAG
