#!/usr/bin/env python3
"""tools/collect.sh run <round> pmc output -> the `address_path_utilisation` object of the bench line:
python3 tools/pmc_l1_json.py <pmcl1 dir> <pmc1 dir> <out.json> <date> [kernel regex]
TA utilisation = TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs); L1 hit = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES;
L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS).  Per-launch medians over the serial launches of the pass (trace_kernel<1>)."""
import collections, csv, glob, json, re, sys

l1dir, cdir, out, date = sys.argv[1:5]
rx = re.compile(sys.argv[5] if len(sys.argv) > 5 else r"trace_kernel")
CUS, XCDS = 256, 8   # GRBM_GUI_ACTIVE is summed over the 8 XCDs (40.7 M for a 2.16 ms launch = 8 x 5.1 M cycles at 2.36 GHz)


def medians(root):
    agg = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if rx.search(r["Kernel_Name"]):
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


a, na = medians(l1dir)
c, nc = medians(cdir)
rec = {"source": f"rocprofv3 --kernel-trace --pmc passes of `bench.py --inflight 1 --steps 3 --warmup 1 --no-extras --no-cpu-baseline` "
                 f"({date}; tools/collect.sh pmc section; per-launch medians over all trace_kernel launches of a pass)",
       "launches_per_pass": max(list(na.values()) + [0])}
if "TA_TA_BUSY_sum" in a and "GRBM_GUI_ACTIVE" in a:
    rec["ta_busy_frac"] = round(a["TA_TA_BUSY_sum"] / (a["GRBM_GUI_ACTIVE"] / XCDS * CUS), 4)
    rec["ta_busy_is"] = "TA_TA_BUSY_sum / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs): share of the kernel's cycles the per-CU texture-address units are busy"
if "TA_TOTAL_WAVEFRONTS_sum" in a and "TA_TA_BUSY_sum" in a:
    rec["ta_cycles_per_wave_instruction"] = round(a["TA_TA_BUSY_sum"] / a["TA_TOTAL_WAVEFRONTS_sum"], 2)
for k in ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"):
    if k in a and "GRBM_GUI_ACTIVE" in a:
        rec[k.lower().replace("_sum", "") + "_frac"] = round(a[k] / (a["GRBM_GUI_ACTIVE"] / XCDS * CUS), 4)
if "TCP_TOTAL_CACHE_ACCESSES_sum" in c and "TCP_TCC_READ_REQ_sum" in c:
    rec["l1_hit"] = round(1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    rec["l1_hit_is"] = "1 - TCP_TCC_READ_REQ_sum / TCP_TOTAL_CACHE_ACCESSES_sum"
if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
    rec["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
