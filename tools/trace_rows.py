#!/usr/bin/env python3
"""Last N kernel rows of a rocprofv3 kernel trace: start / end (us, relative), duration, queue, name."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%9.1f %9.1f %6.1f q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("larva::", "")[:50]))
