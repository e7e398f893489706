#!/usr/bin/env python3
"""Per-kernel breakdown of ONE captured training step from a rocprofv3 kernel trace.
usage: step_kernels.py <dir given to rocprofv3 -d>   (the run must be bench.py, N = 1)"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
step = rows[marks[-3] + 1:marks[-2] + 1]
span = (int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("at::native::", "")[:70]
    agg[name][0] += 1
    agg[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%d kernels, span %.1f us, sum of durations %.1f us" % (len(step), span, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %3d %8.1f" % (k, v[0], v[1]))
