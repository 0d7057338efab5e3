#!/bin/bash
# kernel time of a table-driven workload against the workgroup size (GYMX_TABULAR_BLOCK) and the grid (GYMX_GRID_BLOCKS)
# usage: tools/tab_block_scan.sh "<blocks>" "<grids>" <bench args>
blocks=$1; grids=$2; shift 2
for b in default $blocks; do for g in default $grids; do
  if [ $b = default ]; then unset GYMX_TABULAR_BLOCK; else export GYMX_TABULAR_BLOCK=$b; fi
  if [ $g = default ]; then unset GYMX_GRID_BLOCKS; else export GYMX_GRID_BLOCKS=$g; fi
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 "$@" 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('block', '$b', 'grid', '$g', round(j['roofline']['kernel_ms'],4))" || exit 1
done; done
