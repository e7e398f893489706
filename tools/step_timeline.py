#!/usr/bin/env python3
"""Timeline of ONE captured training step from a rocprofv3 kernel trace: every kernel with start / end
relative to the step's first kernel, its stream (queue), and how long the chip ran 1 / 2 / 3+ kernels
at once.  usage: step_timeline.py <dir given to rocprofv3 -d> [--rows]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw" in r["Kernel_Name"]]
step = rows[marks[-3] + 1:marks[-2] + 1]
t0 = int(step[0]["Start_Timestamp"])
ev = []
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("larva::", "")[:60]
    agg[name][0] += 1
    agg[name][1] += e - s
    ev.append((s, e, name, r.get("Queue_Id", "?"), r.get("Workgroup_Size", ""), r.get("Grid_Size", "")))
    if "--rows" in sys.argv:
        print("%9.1f %9.1f %7.1f  q%-3s grid %-8s %s" % (s, e, e - s, ev[-1][3], ev[-1][5], name))
pts = sorted([(s, 1) for s, e, *_ in ev] + [(e, -1) for s, e, *_ in ev])
depth, last, busy = 0, 0.0, collections.defaultdict(float)
for t, d in pts:
    busy[min(depth, 3)] += t - last
    last = t
    depth += d
span = max(e for _, e, *_ in ev)
print("%d kernels, span %.1f us, sum of durations %.1f us; time with 0 / 1 / 2 / 3+ kernels running: %.1f / %.1f / %.1f / %.1f us"
      % (len(ev), span, sum(e - s for s, e, *_ in ev), busy[0], busy[1], busy[2], busy[3]))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %3d %8.1f  avg %6.2f" % (k, v[0], v[1], v[1] / v[0]))
