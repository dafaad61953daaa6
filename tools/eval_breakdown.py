#!/usr/bin/env python3
"""One evaluation of a rocprofv3 kernel trace broken down by kernel AND launch size (workgroups): which level of the
recursion the small products belong to.  usage: tools/eval_breakdown.py <run_kernel_trace.csv> <kernel substring that only
the wanted evaluation launches, e.g. 'trace_sym_mfma_kernel<3'> [base blocks per evaluation, e.g. 64]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "scale_inputs_kernel" in r["Kernel_Name"]]
ev = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 8] + [len(rows)]
want_base = int(sys.argv[3]) if len(sys.argv) > 3 else None
sel = None
for a, b in zip(ev[:-1], ev[1:]):
    cand = rows[a:b]
    if any(sys.argv[2] in r["Kernel_Name"] for r in cand) and \
       (want_base is None or sum("potrf_inv_base" in r["Kernel_Name"] for r in cand) == want_base):
        sel = cand
t0 = int(sel[0]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in sel:
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:44]
    a = agg.setdefault((nm, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])), [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
print(f"{len(sel)} dispatches, wall {(max(int(r['End_Timestamp']) for r in sel) - t0) / 1e3:.1f} us")
for (nm, wgs), (cnt, ns) in agg.items():
    print(f"{nm:46s} workgroups {wgs:>7d}  x{cnt:4d}  total {ns / 1e3:9.1f} us  avg {ns / cnt / 1e3:8.1f} us")
