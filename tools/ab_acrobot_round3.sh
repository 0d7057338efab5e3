#!/bin/bash
# Runs on the GPU box: timing-only experiment builds of the Acrobot kernel (results are wrong in these builds on
# purpose: they show what a piece of the step costs) against the default build, interleaved.
./tools/micro/rcp64_accuracy || exit 1
for spec in "--env acrobot --settle 300" "--env acrobot --policy random --settle 300"; do
  echo "== $spec"
  timeout -k 10 900 python tools/tools_ab.py default "$@" -- $spec || exit 1
done
