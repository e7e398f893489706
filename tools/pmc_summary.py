#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into the CSV format committed under profiles/ (and read by bench.py at run time):
pass,kernel,counter,launches,mean_per_launch,mean_duration_us.
usage: pmc_summary.py out.csv name=dir [name=dir ...]   (dir = what was given to rocprofv3 -d for that pass)"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = name.split("(")[0].replace("void ", "").strip()
    return name


out = sys.argv[1]
rows = []
for arg in sys.argv[2:]:
    tag, d = arg.split("=", 1)
    cc = glob.glob(d + "/*/*counter_collection.csv")
    if not cc:
        print("no counter_collection.csv under", d)
        continue
    dur = collections.defaultdict(list)
    for kt in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(kt)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(cc[0])):
        k = (short(r["Kernel_Name"]), r["Counter_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    for (kern, ctr), (n, v) in sorted(agg.items()):
        if not re.search(r"larva::(conv3x3|wgrad|head_conv3|bicubic4)", kern):
            continue
        d_us = sum(dur[kern]) / len(dur[kern]) if dur.get(kern) else float("nan")
        rows.append((tag, kern, ctr, n, "%.1f" % (v / n), "%.2f" % d_us))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["pass", "kernel", "counter", "launches", "mean_per_launch", "mean_duration_us"])
    w.writerows(rows)
print("wrote %d rows to %s" % (len(rows), out))
