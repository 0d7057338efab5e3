#!/bin/bash
# Runs on the GPU box: parity of the Acrobot step, then interleaved A/B runs of kernel variants
# (built with tools/tools_variants.sh into gym_amd/lib/variants/).
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py -m gpu -x -q --timeout 180 --timeout-method thread -k "acrobot or ACROBOT or classic or 4" 2>&1 | tail -3
for spec in "--env acrobot --policy random" "--env acrobot" "--env cartpole" "--env pendulum" "--env mountaincar" "--env cartpole --rollout 64 --steps 10" "--env acrobot --policy random --mode fast32"; do
  echo "== $spec"
  timeout -k 10 600 python tools/tools_ab.py default olddiv licmoff -- $spec || exit 1
done
