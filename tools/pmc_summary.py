#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per-kernel mean of each counter (and durations from the kernel trace)."""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(root + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in agg:
    print(k, "calls(pmc rows)", max(len(v) for v in agg[k].values()), "dur_us mean", sum(dur[k]) / max(len(dur[k]), 1))
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} mean {sum(v)/len(v):16.1f}  median {sorted(v)[len(v)//2]:16.1f}  n={len(v)}")
