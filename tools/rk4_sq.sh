#!/bin/bash
# SQ busy / VALU-active counters of the fused RK4/binary64 rollout kernel, under gpurun_out/rk4_sq/.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/rk4_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $R/bench.py --env acrobot --rk4 --rollout 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU --output-format csv -d $OUT/sq -- $SHORT > $OUT/sq.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/sq/*/*counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "Acrobot64" in r["Kernel_Name"] and "step" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, sum(v) / len(v), len(v))
PY
