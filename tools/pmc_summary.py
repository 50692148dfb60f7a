#!/usr/bin/env python
"""Reduce a rocprofv3 --pmc counter_collection CSV to per-kernel totals (the raw CSVs are tens of MB).
   python tools/pmc_summary.py <counter_collection.csv> <out.csv>"""
import collections
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(lambda: [0, 0.0])
counter = None
with open(src, newline="") as f:
    for r in csv.DictReader(f):
        counter = r["Counter_Name"]
        k = (r["Kernel_Name"], counter)
        tot[k][0] += 1
        tot[k][1] += float(r["Counter_Value"])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Counter_Sum", "Counter_Avg_Per_Dispatch"])
    for (k, c), (n, v) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, c, n, f"{v:.1f}", f"{v / n:.3f}"])
print("wrote", dst, len(tot), "rows")
