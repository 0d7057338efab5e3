#!/bin/bash
# SQ counters of the EXACT Acrobot single-step kernel (four passes) in the stationary regime of its policy: 300
# settling steps, the averages are over the last 12 launches.  Output under gpurun_out/acro_pmc/.
# usage: tools/acrobot_pmc.sh [bench args, e.g. --policy random]
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/acro_pmc
rm -rf $OUT/sq $OUT/fp $OUT/ws $OUT/ms
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $R/bench.py --env acrobot --settle 300 --steps 10 --warmup 2 --no-cpu-baseline --no-extra $*"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/sq -- $SHORT > $OUT/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_SALU --output-format csv -d $OUT/fp -- $SHORT > $OUT/fp.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $OUT/ws -- $SHORT > $OUT/ws.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/ms -- $SHORT > $OUT/ms.log 2>&1
python3 - <<PY
import csv, glob, collections
print("bench args: $*")
for d in ("sq", "fp", "ws", "ms"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "Acrobot" in r["Kernel_Name"] and "step" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, v in sorted(acc.items()):
        last = [x for _, x in sorted(v)[-12:]]
        print(d, k, sum(last) / len(last), "per launch; per four-env trip (65 536 trips): %.1f" % (sum(last) / len(last) / 65536))
PY
