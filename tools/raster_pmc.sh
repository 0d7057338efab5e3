#!/bin/bash
# Runs on the GPU box: SQ counters of a frame env's raster kernel under two workgroup shapes.
# usage: tools/raster_pmc.sh <env> <GYMX_FRAME_RASTER or GYMX_PONG_RASTER value> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ENV=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  tag=$(echo $v | tr ',' '_')
  OUT=$R/gpurun_out/raster_pmc/${ENV}_$tag
  mkdir -p $OUT
  export GYMX_FRAME_RASTER=$v GYMX_PONG_RASTER=$v
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_SMEM --output-format csv -d $OUT -- python3 $R/bench.py --env $ENV --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/log.txt 2>&1 || exit 1
  python3 - $OUT $v <<'P'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "raster" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v)) for k, v in sorted(agg.items())})
P
done
