#!/bin/bash
# Runs on the GPU box: parity of the classic-control kernels under each variant library, then interleaved A/B timing.
# Variants (tools/tools_variants.sh): remat = polynomial constants moved into a fixed scalar pair right before each
# v_fma_f64 instead of living in scalar registers; remat4 / minw4 = with / without that, compiled for four waves per SIMD.
for v in remat remat4 minw4; do
  GYMX_LIB=$PWD/gym_amd/lib/variants/libgymx_$v.so timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 180 --timeout-method thread -k "classic or cartpole or acrobot or pendulum or mountaincar" 2>&1 | tail -1 || exit 1
done
for spec in "--env acrobot --policy random" "--env acrobot" "--env cartpole" "--env pendulum" "--env cartpole --rollout 64 --steps 10" "--env acrobot --rk4 --rollout 64 --steps 5"; do
  echo "== $spec"
  timeout -k 10 600 python tools/tools_ab.py default remat remat4 minw4 -- $spec || exit 1
done
