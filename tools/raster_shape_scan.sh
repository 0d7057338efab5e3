#!/bin/bash
# Runs on the GPU box: one short bench run per workgroup shape of a frame env's raster (threads, workgroups per frame),
# to find the shapes worth an interleaved comparison.
# usage: ENV=pong|breakout|spaceinvaders [THREADS="64 128 256"] [SPLITS="1 2 3 ..."] tools/raster_shape_scan.sh
ENV=${ENV:-pong}
VAR=GYMX_FRAME_RASTER; [ "$ENV" = pong ] && VAR=GYMX_PONG_RASTER
for t in ${THREADS:-64 128 256}; do
  for s in ${SPLITS:-$(seq 1 ${MAXSPLIT:-40})}; do
    ms=$(env $VAR=$t,$s timeout -k 10 60 python bench.py --env $ENV --steps 30 --warmup 5 --settle 0 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print('%.4f' % j['roofline']['kernel_ms'])") || exit 1
    echo "$ENV $t,$s $ms"
  done
done
