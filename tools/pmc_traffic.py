#!/usr/bin/env python3
"""HBM traffic of trace_kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has
4 counter slots, MI355X_MICROARCH.md "rocprofv3 PMC slots") -> profiles/trace_traffic.json, the figure bench.py attaches
as roofline.traffic for this exact workload.

    python3 tools/pmc_traffic.py <dir with the FETCH_SIZE pass> <dir with the WRITE_SIZE pass> <out.json> [label]

gfx950 correction (same guide, "HBM"): FETCH_SIZE reports half the bytes of 16-byte-per-lane loads -> x2; WRITE_SIZE is exact.
Both counters are in KiB-like units of 1024 bytes as rocprofv3 derives them (FETCH_SIZE = TCC_EA0_RDREQ * 64 / 1024)."""
import collections
import csv
import glob
import json
import os
import sys


def mean_counter(root, name, kernel="trace_kernel"):
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    best = max(vals.items(), key=lambda kv: len(kv[1]))
    v = sorted(best[1])
    return best[0], v[len(v) // 2], len(v)


fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
label = sys.argv[4] if len(sys.argv) > 4 else ""
kf, fetch_kb, nf = mean_counter(fetch_dir, "FETCH_SIZE")
kw, write_kb, nw = mean_counter(write_dir, "WRITE_SIZE")
rec = {
    "kernel": kf.split("(")[0],
    "workload": ("grid_mesh(2237,1) = 10M triangles" if "config4" in label else "grid_mesh(708,1)") + ", 1920x1080, camera A, kDepth, 1 spp, LBVH "
                "(python3 bench.py " + ("--preset config4 " if "config4" in label else "") + "--inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline)",
    "FETCH_SIZE_KB_raw_median": fetch_kb, "WRITE_SIZE_KB_raw_median": write_kb, "launches_sampled": [nf, nw],
    "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B/lane loads -> x2 (MI355X_MICROARCH.md, HBM); "
                  "WRITE_SIZE exact; separate --pmc passes (tools/collect.sh, pmc section)",
    "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {label}".strip(", "),
}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
