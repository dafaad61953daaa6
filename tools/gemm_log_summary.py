#!/usr/bin/env python3
"""Summarise the per-launch GEMM log that GPX_GEMM_LOG=1 makes bench.py print to stderr (shape, flags, ms, TFLOP/s)."""
import collections
import re
import sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r'gpx gemm M=(\d+) N=(\d+) K=(\d+) akc=(\d) bkc=(\d) flags=(\d+)\s+([\d.]+) ms\s+([\d.]+)', l)
    if m:
        rows.append(tuple(float(x) for x in m.groups()))
tot = sum(r[6] for r in rows)
print(len(rows), 'launches', round(tot, 1), 'ms')
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = tuple(int(v) for v in r[:6])
    agg[k][0] += 1
    agg[k][1] += r[6]
    agg[k][2] += r[6] * r[7]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('M=%d N=%d K=%d akc=%d bkc=%d flags=%d' % k, ' x%d' % v[0], '%.1f ms %.1f%%  %.1f TF' % (v[1], 100 * v[1] / tot, v[2] / v[1]))
