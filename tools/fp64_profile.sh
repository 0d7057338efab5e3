#!/bin/bash
# FP64 VALU counters of the fused RK4 rollout kernel (runs on the GPU box through gpurun).
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/fp64
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc -- python3 $R/tools/rollout_driver.py 16777216 64 2 > $OUT/pmc.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/rollout_driver.py 16777216 64 3 > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "Acrobot64" in r["Kernel_Name"] and "step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
dur = None
for f in glob.glob("$OUT/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "Acrobot64" in r["Name"] and "step_kernel" in r["Name"]:
            dur = float(r["AverageNs"]) * 1e-9
flops = (m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + 2 * m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0)) * 64
out = {"counters_per_launch": m, "kernel_seconds": dur, "fp64_flops_per_launch": flops,
       "fp64_tflops": flops / dur / 1e12 if dur else None, "frac_of_78.6_tflops": flops / dur / 78.6e12 if dur else None,
       "f64_instr_share_of_valu": (m.get("SQ_INSTS_VALU_ADD_F64", 0) + m.get("SQ_INSTS_VALU_MUL_F64", 0) + m.get("SQ_INSTS_VALU_FMA_F64", 0) + m.get("SQ_INSTS_VALU_TRANS_F64", 0)) / max(m.get("SQ_INSTS_VALU", 1), 1)}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -1 $OUT/trace.log
