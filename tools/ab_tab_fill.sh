#!/bin/bash
# Runs on the GPU box: interleaved A/B of two builds of the tabular kernels (gym_amd/lib/variants/libgymx_base.so
# against the regular library) over the table-driven workloads at 64 Mi envs.
set -e
for spec in "--env taxi --num-envs 67108864" "--env taxi --num-envs 67108864 --policy random" "--env taxi --taxi-full --num-envs 67108864" "--env taxi --taxi-full --num-envs 67108864 --policy random" "--env frozenlake --map 8x8 --slippery --num-envs 67108864" "--env frozenlake --map 8x8 --slippery --num-envs 67108864 --policy random" "--env frozenlake --num-envs 67108864" "--env frozenlake --num-envs 67108864 --policy random"; do
  echo "== $spec"
  timeout -k 10 400 python tools/tools_ab.py base default -- $spec
done
