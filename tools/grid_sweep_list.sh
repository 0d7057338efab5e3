#!/bin/bash
# kernel time per step against the number of workgroups, for a list of grid sizes
# usage: tools/grid_sweep_list.sh "<grid sizes>" <bench.py workload flags...>
grids=$1; shift
for g in default $grids; do
  if [ $g = default ]; then unset GYMX_GRID_BLOCKS; else export GYMX_GRID_BLOCKS=$g; fi
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 "$@" | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid', '$g', round(j['roofline']['kernel_ms'],4))" || exit 1
done
