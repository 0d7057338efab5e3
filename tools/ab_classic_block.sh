#!/bin/bash
# Runs on the GPU box: the classic-control steps with 128 and 256 threads per workgroup against 64 (one wave).
GYMX_LIB=$PWD/gym_amd/lib/variants/libgymx_block256.so timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 180 --timeout-method thread -k "classic or fused or rollout" 2>&1 | tail -1 || exit 1
for spec in "--env cartpole" "--env mountaincar" "--env mountaincarcontinuous" "--env pendulum" "--env acrobot" "--env cartpole --policy random" "--env cartpole --mode fp64"; do
  echo "== $spec"
  timeout -k 10 600 python tools/tools_ab.py default block128 block256 -- $spec || exit 1
done
