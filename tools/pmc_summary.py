#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs: per kernel name, the sum over dispatches."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in sorted(glob.glob(os.path.join(out, "*counter_collection.csv"))):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (f, r["Dispatch_Id"])
        if key not in seen and r["Counter_Name"] in ("SQ_WAVE_CYCLES", "FETCH_SIZE"):
            seen.add(key)
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for name, cs in agg.items():
        line = name + "\n" + "\n".join(f"    {k:32s} {v:.6g}" for k, v in sorted(cs.items()))
        print(line)
        fh.write(line + "\n")
