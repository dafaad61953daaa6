#!/usr/bin/env python3
"""Timeline of ONE evaluation from a rocprofv3 kernel trace (run_kernel_trace.csv): every dispatch in
start order with its duration and the idle gap before it, plus busy / idle totals.  An evaluation is
delimited by a marker kernel (default scale_inputs_kernel, the first launch of every NLL call).
usage: tools/trace_timeline.py <run_kernel_trace.csv> [eval_index=-1] [marker] [--full]"""
import csv
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
marker = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith("--") else "scale_inputs_kernel"
full = "--full" in sys.argv
rows = [r for r in csv.DictReader(open(path)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
# consecutive markers closer than 8 dispatches belong to the same evaluation (X and U are both scaled)
evals = [s for k, s in enumerate(starts) if k == 0 or s - starts[k - 1] > 8]
lo = evals[which]
hi = evals[evals.index(lo) + 1] if evals.index(lo) + 1 < len(evals) else len(rows)
sel = rows[lo:hi]
# drop trailing dispatches that are separated from the evaluation by a long host pause
t0 = int(sel[0]["Start_Timestamp"])
busy = 0
gaps = []
prev_end = t0
agg = {}
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    gap = s - prev_end
    gaps.append((gap, name))
    busy += e - s
    a = agg.setdefault(name, [0, 0, 0])
    a[0] += 1
    a[1] += e - s
    a[2] += max(gap, 0)
    if full:
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  gap {gap / 1e3:8.1f} us  s{r.get('Stream_Id', '?'):>2s}  {name}")
    prev_end = max(prev_end, e)
wall = prev_end - t0
print(f"evaluation {which}: {len(sel)} dispatches, wall {wall / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(wall - busy) / 1e6:.3f} ms")
print(f"{'kernel':60s} {'calls':>6s} {'busy us':>10s} {'gap-before us':>14s}")
for name, (c, b, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{name:60s} {c:6d} {b / 1e3:10.1f} {g / 1e3:14.1f}")
