#!/bin/bash
# step + raster time of a frame env against the raster launch shape (GYMX_FRAME_RASTER = threads,workgroups per frame)
# usage: tools/raster_shape_list.sh "<shapes>" <bench args>      GYMX_LIB selects another build
shapes=$1; shift
for s in default $shapes; do
  if [ $s = default ]; then unset GYMX_FRAME_RASTER; else export GYMX_FRAME_RASTER=$s; fi
  timeout -k 10 200 python bench.py --no-extra --no-cpu-baseline --steps 100 "$@" | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shape', '$s', round(j['roofline']['kernel_ms'],4))" || exit 1
done
