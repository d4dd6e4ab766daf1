#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv (+ kernel_trace.csv for durations) per
kernel symbol for the LAST bench step: usage  pmc_summary.py <dir with *_counter_collection.csv>"""
import csv
import glob
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"dvsg::\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    return n[:48]


def main(d):
    cc = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    rows = list(csv.DictReader(open(cc)))
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for r in rows:
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    names = sorted({r["Counter_Name"] for r in rows})
    print("%-48s %5s %10s " % ("kernel", "n", "us_total") + " ".join("%22s" % n for n in names))
    for k in sorted(per, key=lambda k: -sum(dur.get(i, 0) for i in disp[k])):
        if "at::" in k or "rocclr" in k:
            continue
        t = sum(dur.get(i, 0) for i in disp[k])
        print("%-48s %5d %10.1f " % (k, len(disp[k]), t) + " ".join("%22.4g" % per[k][n] for n in names))


if __name__ == "__main__":
    main(sys.argv[1])
