#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats of the default bench run, then the
# two HBM-traffic counter passes, each in its own rocprofv3 run.  Output under gpurun_out/prof/.
# usage: [PROF_DIR=name] tools/collect_profiles.sh [bench args...]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof}
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq  # a run's files only
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
SHORT="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra $*"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $SHORT > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $SHORT > $OUT/pmc_write.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/pmc_sq -- $SHORT > $OUT/pmc_sq.log 2>&1
python3 $R/tools/summarize_profiles.py $OUT
