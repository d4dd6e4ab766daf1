#!/usr/bin/env python3
"""Sum a rocprofv3 --stats kernel summary: total GPU-busy time and the top kernels.
Usage: tools/stats_sum.py <dir with *kernel_stats.csv> [top]"""
import csv, glob, os, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
files = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(files[-1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%s: %d kernels, %.3f ms GPU-busy, %d launches" % (files[-1], len(rows), tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print("%6.2f%% calls=%6s avg=%9.1fus %s" % (100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
