#!/bin/bash
# Taxi (14 KB table staged in LDS per workgroup): step time against the number of workgroups
for spec in "--env taxi --num-envs 67108864" "--env taxi --taxi-full --num-envs 67108864 --policy random"; do
  for g in 0 16384 8192 4096 2048 1024; do
    if [ $g = 0 ]; then unset GYMX_GRID_BLOCKS; else export GYMX_GRID_BLOCKS=$g; fi
    timeout -k 10 200 python bench.py $spec --no-extra --no-cpu-baseline --steps 100 | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$spec', 'grid', $g, round(j['roofline']['kernel_ms'],4))" || exit 1
  done
done
