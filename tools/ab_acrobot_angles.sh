#!/bin/bash
# Runs on the GPU box: parity of the Acrobot kernels, then interleaved timing of the Acrobot EXACT step with the angles
# kept in two more rows (default; nointer = without the interleaved sincos evaluation) against the six-row form
# (GYMX_ACROBOT_ANGLES=0).
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q --timeout 180 --timeout-method thread -k "acrobot" 2>&1 | tail -2 || exit 1
for spec in "--env acrobot" "--env acrobot --policy random" "--env acrobot --rollout 64 --steps 10"; do
  echo "== $spec"
  timeout -k 10 600 python tools/tools_ab.py default nointer -- $spec || exit 1
  GYMX_ACROBOT_ANGLES=0 timeout -k 10 600 python tools/tools_ab.py default nointer -- $spec | sed 's/^/six-row /' || exit 1
done
