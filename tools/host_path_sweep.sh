#!/bin/bash
# host path latency with and without the pinned staging, over batch sizes
for n in 1024 8192 32768; do
  echo "== n=$n staged"; timeout -k 10 120 python tools/host_path_rate.py $n || exit 1
  echo "== n=$n copy engine"; GYMX_STAGE_LIMIT=0 timeout -k 10 120 python tools/host_path_rate.py $n || exit 1
done
for n in 65536 262144 1048576; do
  echo "== n=$n staged (limit raised)"; GYMX_STAGE_LIMIT=268435456 timeout -k 10 120 python tools/host_path_rate.py $n || exit 1
  echo "== n=$n copy engine"; GYMX_STAGE_LIMIT=0 timeout -k 10 120 python tools/host_path_rate.py $n || exit 1
done
