#!/usr/bin/env python3
"""Timeline of the LAST host-pointer call in a rocprofv3 --kernel-trace --memory-copy-trace run: start/end of every copy
and kernel relative to the first, to see what overlaps.  usage: timeline_summary.py <dir with *_kernel_trace.csv, *_memory_copy_trace.csv>"""
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_find_range" in r["Kernel_Name"]:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "kernel"))
for f in glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy")))
ev.sort()
# the last burst: events after the last gap > 50 ms
last = 0
for i in range(1, len(ev)):
    if ev[i][0] - ev[i - 1][1] > 50_000_000:
        last = i
ev = ev[last:]
t0 = ev[0][0]
for s, e, what in ev[:80]:
    print(f"{(s - t0) / 1e6:9.3f} .. {(e - t0) / 1e6:9.3f} ms  ({(e - s) / 1e6:7.3f})  {what}")
print("events in burst:", len(ev), " span ms:", (max(e for _, e, _ in ev) - t0) / 1e6)
