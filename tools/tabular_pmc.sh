#!/bin/bash
# SQ counters of the tabular step kernel for one bench workload: tools/tabular_pmc.sh <tag> <bench args...>
# Output: gpurun_out/tab_pmc_<tag>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
OUT=$R/gpurun_out/tab_pmc_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra $*"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -- $SHORT > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- $SHORT > $OUT/sq2.log 2>&1
python3 - <<PY > $R/gpurun_out/tab_pmc_$tag.txt
import csv, glob, collections
for d in ("sq", "sq2"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "tabular_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        v = v[2:]
        print("$tag", k, "%.4g" % (sum(v) / len(v)), "per launch over", len(v), "launches")
PY
cat $R/gpurun_out/tab_pmc_$tag.txt
