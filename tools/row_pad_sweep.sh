#!/bin/bash
# Runs on the GPU box: step time against the padding between the SoA rows of a batch (GYMX_ROW_PAD, elements).
# With 16 Mi envs the rows of a batch are exactly 64 MiB apart; a pad moves them off that power of two.
# usage: tools/row_pad_sweep.sh "<bench args>" pad...
set -e
spec=$1; shift
for r in 1 2; do for pad in "$@"; do
  GYMX_ROW_PAD=$pad timeout -k 10 200 python bench.py $spec --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', 'pad', $pad, round(j['roofline']['kernel_ms'],4))" || exit 1
done; done
