#!/bin/bash
# Runs on the GPU box: the binary32 classic-control steps with the stores recomputing their addresses (recomp) and
# with the lane's env states parked in LDS as well (ldsrecomp) against the default; parity of the variants first.
for v in recomp ldsrecomp; do
  GYMX_LIB=$PWD/gym_amd/lib/variants/libgymx_$v.so timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 180 --timeout-method thread -k "classic or cartpole or pendulum or mountaincar or fused or rollout" 2>&1 | tail -1 || exit 1
done
for spec in "--env cartpole" "--env cartpole --policy random" "--env mountaincar" "--env mountaincarcontinuous" "--env pendulum" "--env cartpole --mode fast32" "--env cartpole --rollout 64 --steps 10"; do
  echo "== $spec"
  timeout -k 10 600 python tools/tools_ab.py default recomp ldsrecomp -- $spec || exit 1
done
