#!/bin/bash
# Runs on the GPU box: frames of ENV (breakout | spaceinvaders) bit-exact under every workgroup shape of its
# raster (GYMX_FRAME_RASTER=<threads>,<workgroups per frame>), then interleaved timing of the shapes.
ENV=${ENV:-breakout}
VARIANTS=${VARIANTS:-"256,1 64,22 64,27 64,33 64,44 64,66"}
for v in $VARIANTS; do
  GYMX_FRAME_RASTER=$v timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 180 --timeout-method thread -k "$ENV or invaders" 2>&1 | tail -1 || exit 1
done
B="python bench.py --env $ENV --steps 100 --warmup 10 --no-cpu-baseline --no-extra"
for r in 1 2 3; do
  for v in $VARIANTS; do
    echo -n "raster $v: "
    GYMX_FRAME_RASTER=$v timeout -k 10 120 $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print('%.4f ms  %.1f%%' % (j['roofline']['kernel_ms'], 100*j['roofline']['frac']))" || exit 1
  done
done
