#!/bin/bash
# Runs on the GPU box: interleaved repeats of a few raster shapes.  usage: ENV=... SHAPES="256,12 128,25" [REPS=3] [ENVS=65536] tools/raster_shape_ab.sh
ENV=${ENV:-pong}
VAR=GYMX_FRAME_RASTER; [ "$ENV" = pong ] && VAR=GYMX_PONG_RASTER
for rep in $(seq 1 ${REPS:-3}); do
  for sh in $SHAPES; do
    ms=$(env $VAR=$sh timeout -k 10 60 python bench.py --env $ENV --num-envs ${ENVS:-65536} --steps 50 --warmup 10 --settle 0 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % j['roofline']['kernel_ms'])") || exit 1
    echo "$ENV n=${ENVS:-65536} $sh rep$rep $ms"
  done
done
