#!/usr/bin/env python3
"""Per-kernel HBM traffic from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.
Units/corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in
KiB-like units of 1 KB as reported by rocprofv3 (kilobytes); FETCH_SIZE under-reports wide coalesced
streaming reads by exactly 2x (128-B requests tallied at 64 B) -> doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
evals = int(sys.argv[2]) if len(sys.argv) > 2 else 1   # evaluations the profiled command ran
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in sorted(glob.glob(os.path.join(out, "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
res = {}
for name, cs in agg.items():
    n = max(cnt[name].values())
    e = {"launches": n, "launches_per_eval": n / evals}
    if "FETCH_SIZE" in cs:
        e["fetch_bytes_per_launch_corrected"] = 2.0 * cs["FETCH_SIZE"] * 1024.0 / cnt[name]["FETCH_SIZE"]
    if "WRITE_SIZE" in cs:
        e["write_bytes_per_launch"] = cs["WRITE_SIZE"] * 1024.0 / cnt[name]["WRITE_SIZE"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE", 0) > 0:
        # MFMA busy cycles summed over 1024 SIMDs vs GUI-active cycles summed over 8 XCDs
        e["mfma_util"] = (cs["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (cs["GRBM_GUI_ACTIVE"] / 8.0)
    res[name] = e
json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -kv[1].get("fetch_bytes_per_launch_corrected", 0) * kv[1]["launches"]):
    print(k[:60].ljust(60), v)
