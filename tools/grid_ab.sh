#!/bin/bash
# Runs on the GPU box: step time of one workload for several grid sizes (GYMX_GRID_BLOCKS), two rounds.
# usage: tools/grid_ab.sh "<bench args>" grid...
set -e
spec=$1; shift
for r in 1 2; do for g in "$@"; do
  GYMX_GRID_BLOCKS=$g timeout -k 10 200 python bench.py $spec --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', 'grid', $g, round(j['roofline']['kernel_ms'],4))" || exit 1
done; done
