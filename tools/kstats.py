#!/usr/bin/env python3
"""Prints a rocprofv3 kernel_stats.csv as a table: name, calls, total ms, avg us, %."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
for r in rows:
    print(r["Name"][:78].ljust(78), r["Calls"].rjust(7), f"{float(r['TotalDurationNs']) / 1e6 / div:11.3f} ms",
          f"{float(r['AverageNs']) / 1e3:10.1f} us", f"{float(r['Percentage']):6.2f}%")
