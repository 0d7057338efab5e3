#!/bin/bash
# Copies what tools/collect_profiles.sh left under gpurun_out/prof_<env>/ into profiles/<round>/ and refreshes
# profiles/traffic.json.  usage: tools/publish_profiles.sh r2
set -e
cd "$(dirname "$0")/.."
R=$1
for e in cartpole pong breakout spaceinvaders; do
  d=gpurun_out/prof_$e
  [ -d $d ] || continue
  if [ $e = cartpole ]; then pre=""; stats=kernel_stats_bench_default; sum=summary_cartpole_16Mi; n=16777216; else pre="${e}_"; stats=${e}_kernel_stats; sum=${e}_summary; n=65536; fi
  # gpurun merges a run's files into what earlier runs left here: take the newest of each kind
  newest() { ls -t $1 | head -1; }
  cp $(newest "$d/trace/vm/*_kernel_stats.csv") profiles/$R/$stats.csv
  cp $(newest "$d/pmc_fetch/vm/*_counter_collection.csv") profiles/$R/${pre}pmc_fetch_size.csv
  cp $(newest "$d/pmc_write/vm/*_counter_collection.csv") profiles/$R/${pre}pmc_write_size.csv
  cp $(newest "$d/pmc_sq/vm/*_counter_collection.csv") profiles/$R/${pre}pmc_sq.csv
  cp $d/summary.json profiles/$R/$sum.json
  python tools/update_traffic.py $d/summary.json $e:$n:device profiles/$R/${pre}pmc_fetch_size.csv profiles/$R/${pre}pmc_write_size.csv | grep '"hbm_bytes_per_launch"'
  head -3 profiles/$R/$stats.csv | cut -c1-140
done
