#!/usr/bin/env python3
"""Print the per-kernel summary of a rocprofv3 --kernel-trace --stats run: python3 tools/kstats.py <dir> [filter]"""
import csv, glob, os, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for p in sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)):
    print("#", os.path.relpath(p, d))
    for r in csv.DictReader(open(p)):
        name = r["Name"].split("(")[0].replace("void ", "").replace("rt::", "")
        if flt and flt not in name:
            continue
        print(f"{name[:48]:48s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
