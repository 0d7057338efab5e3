#!/bin/bash
# Runs on the GPU box: step time against the byte offset between the starts of a batch's separately allocated row
# arrays (GYMX_ROW_SKEW, bytes per row).  The second number of a pair was an offset of bench.py's action ring into
# its allocation, which made no difference and is no longer an option.
# usage: tools/row_skew_sweep.sh "<bench args>" "skew,0" ...
set -e
spec=$1; shift
for r in 1 2; do for pair in "$@"; do
  GYMX_ROW_SKEW=${pair%,*} timeout -k 10 200 python bench.py $spec --no-extra --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', 'row skew, ring skew', '$pair', round(j['roofline']['kernel_ms'],4))" || exit 1
done; done
