#!/bin/bash
# Runs on the GPU box: Pong frames bit-exact with the 16-byte-store raster, then interleaved timing of its
# workgroup shapes (GYMX_PONG_RASTER=<threads>,<workgroups per frame>) against the 12-byte-store form (=12).
VARIANTS=${VARIANTS:-"256,1 12 256,5 256,25 64,1 64,5 64,25 64,99"}
for v in $VARIANTS; do
  GYMX_PONG_RASTER=$v timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 180 --timeout-method thread -k "pong or PONG" 2>&1 | tail -1 || exit 1
done
B="python bench.py --env pong --steps 100 --warmup 10 --no-cpu-baseline --no-extra"
for r in 1 2 3; do
  for v in $VARIANTS; do
    echo -n "raster $v: "
    GYMX_PONG_RASTER=$v timeout -k 10 120 $B 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.readlines()[-1]); print('%.4f ms  %.1f%%' % (j['roofline']['kernel_ms'], 100*j['roofline']['frac']))" || exit 1
  done
done
