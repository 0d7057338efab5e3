"""Writes one entry of profiles/traffic.json from a summary.json made by tools/collect_profiles.sh
(tools/summarize_profiles.py), stamped with the kernel name and the hash of the kernel sources the
counters were taken on; bench.py reports the entry only while that hash matches its build.
usage: python tools/update_traffic.py <summary.json> <env:num_envs:policy> <profiles/rN/...csv names for the source note>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gym_amd.build import csrc_hash  # noqa: E402

summary, key, note = json.load(open(sys.argv[1])), sys.argv[2], " ".join(sys.argv[3:])
rows = {k: v for k, v in summary["pmc"].items() if "hbm_bytes_per_launch" in v}
entry = {"hbm_bytes_per_launch": sum(v["hbm_bytes_per_launch"] for v in rows.values()),
         "hbm_read_bytes_per_launch": sum(v["hbm_read_bytes_per_launch"] for v in rows.values()),
         "hbm_write_bytes_per_launch": sum(v["hbm_write_bytes_per_launch"] for v in rows.values()),
         "kernel": sorted(rows), "csrc_hash": csrc_hash(),
         "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 per the gfx950 correction; " + note}
path = os.path.join(ROOT, "profiles", "traffic.json")
table = json.load(open(path)) if os.path.exists(path) else {}
table[key] = entry
json.dump(table, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
