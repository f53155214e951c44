#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter), averaged per dispatch.
usage: summarize_pmc.py <dir with p*/.../*_counter_collection.csv>"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: [0.0, 0, 0.0])
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    per_dispatch = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("rbg::(anonymous namespace)::", "rbg::").split("(")[0].replace("void ", "")
        per_dispatch[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, _d), v in per_dispatch.items():
        acc[(k, c)][0] += v
        acc[(k, c)][1] += 1
        acc[(k, c)][2] = max(acc[(k, c)][2], v)
kern = sorted({k for k, _ in acc})
for k in kern:
    print(f"## {k}")
    for (kk, c), (v, n, mx) in sorted(acc.items()):
        if kk == k:
            print(f"  {c:34s} per-dispatch avg = {v / n:18.1f}   max = {mx:18.1f}   (dispatches {n})")
