#!/usr/bin/env python3
"""GPU time per roctx range from a `DVSG_ROCTX=1 rocprofv3 --kernel-trace --marker-trace` run of bench.py:
    python tools/marker_summary.py gpurun_out/prof_markers profiles/r03_marker_ranges.csv
rocprofv3 gives every kernel dispatch the correlation id of the roctx range it was launched in, so the join is exact.
Per range name: calls, kernels per call, GPU microseconds per call (kernel durations summed; the ranges themselves time
the host-side enqueue only)."""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def newest(pattern):
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def main(src, dst):
    markers = list(csv.DictReader(open(newest(src + "/**/*_marker_api_trace.csv"))))
    kernels = list(csv.DictReader(open(newest(src + "/**/*_kernel_trace.csv"))))
    by_corr = defaultdict(list)
    for k in kernels:
        by_corr[k["Correlation_Id"]].append((int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e3)
    rows = OrderedDict()
    for m in markers:
        d = by_corr.get(m["Correlation_Id"], [])
        r = rows.setdefault(m["Function"], {"calls": 0, "kernels": 0, "gpu_us": 0.0})
        r["calls"] += 1
        r["kernels"] += len(d)
        r["gpu_us"] += sum(d)
    total = sum(r["gpu_us"] / r["calls"] for r in rows.values())
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["range", "calls", "kernels_per_call", "gpu_us_per_call", "share_of_step"])
        for name, r in rows.items():
            w.writerow([name, r["calls"], "%.1f" % (r["kernels"] / r["calls"]), "%.1f" % (r["gpu_us"] / r["calls"]),
                        "%.4f" % (r["gpu_us"] / r["calls"] / total)])
        w.writerow(["(sum of ranges)", "", "", "%.1f" % total, "1.0"])
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
