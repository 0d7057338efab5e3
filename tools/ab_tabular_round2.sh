#!/bin/bash
# Runs on the GPU box: single-step kernel times of the table-driven workloads at 64 Mi envs.
for spec in "--env frozenlake --map 8x8 --slippery --num-envs 67108864 --policy random" "--env frozenlake --map 8x8 --slippery --num-envs 67108864" "--env frozenlake --num-envs 67108864" "--env frozenlake --num-envs 67108864 --policy random" "--env taxi --num-envs 67108864" "--env taxi --num-envs 67108864 --policy random" "--env taxi --taxi-full --num-envs 67108864 --policy random"; do
  for rep in 1 2; do
  echo "$spec: $(timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra $spec 2>/dev/null | python -c 'import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4f ms  %.1f%% of 8 TB/s at %d B" % (j["roofline"]["kernel_ms"], 100*j["roofline"]["frac"], j["roofline"]["bytes_per_env_step"]))')"
  done
done
