#!/bin/bash
# Runs on the GPU box: FETCH_SIZE and WRITE_SIZE (separate passes) of the step kernel of one bench workload.
# usage: tools/pmc_traffic_of.sh <tag> <bench args...>   -> gpurun_out/traffic_<tag>.txt
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
OUT=$R/gpurun_out/traffic_$tag
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra $*"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $SHORT > $OUT/fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $SHORT > $OUT/write.log 2>&1
python3 - <<PY > $R/gpurun_out/traffic_$tag.txt
import csv, glob, collections
res = {}
for d in ("fetch", "write"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % d)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[2:]  # past the warm-up launches
        res.setdefault(k, {})[d] = sum(v) / len(v) * 1024 * (2 if d == "fetch" else 1)  # KiB units; FETCH_SIZE x2 on gfx950
for k, v in res.items():
    print("$tag", k, "read %.1f MB  written %.1f MB  total %.1f MB per launch" % (v.get("fetch", 0) / 1e6, v.get("write", 0) / 1e6, (v.get("fetch", 0) + v.get("write", 0)) / 1e6))
PY
cat $R/gpurun_out/traffic_$tag.txt
