#!/bin/bash
# Runs on the GPU box: parity of the Acrobot kernels with the table-driven sin/cos, then interleaved A/B runs against
# builds with the polynomial routines (tools/tools_variants.sh notable / notable64 ... -DGX_ACROBOT_NO_TABLE).
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_fuzz.py tests/test_gpu_acrobot_angles.py -m gpu -x -q --timeout 180 --timeout-method thread 2>&1 | tail -5 || exit 1
for spec in "--env acrobot --settle 300" "--env acrobot --policy random --settle 300" "--env acrobot --rollout 64 --steps 10"; do
  echo "== $spec"
  timeout -k 10 900 python tools/tools_ab.py default notable -- $spec || exit 1
done
for spec in "--env acrobot --rk4 --rollout 64 --steps 5" "--env acrobot --rk4 --policy random --settle 300"; do
  echo "== $spec"
  timeout -k 10 900 python tools/tools_ab.py default notable64 -- $spec || exit 1
done
