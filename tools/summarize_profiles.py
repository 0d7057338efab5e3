"""Condenses rocprofv3 output (kernel stats + PMC passes) into one JSON summary."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
summary = {"kernels": [], "pmc": {}}
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        summary["kernels"].append({"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                   "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])})
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for f in glob.glob(out + "/" + d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "step_kernel" not in k and "raster" not in k:
                continue
            for c, v in cs.items():
                summary["pmc"].setdefault(k, {})[c] = {"launches": len(v), "mean": sum(v) / len(v)}
# HBM traffic per launch of the dominant step kernel: FETCH_SIZE and WRITE_SIZE are in KiB;
# on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced stream, so it is
# doubled before it is compared with a byte count; WRITE_SIZE is exact for 16-B-per-lane stores.
for k, cs in summary["pmc"].items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        cs["hbm_read_bytes_per_launch"] = cs["FETCH_SIZE"]["mean"] * 1024 * 2
        cs["hbm_write_bytes_per_launch"] = cs["WRITE_SIZE"]["mean"] * 1024
        cs["hbm_bytes_per_launch"] = cs["hbm_read_bytes_per_launch"] + cs["hbm_write_bytes_per_launch"]
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: {c: v for c, v in cs.items() if c.startswith("hbm")} for k, cs in summary["pmc"].items()}, indent=1))
