#!/usr/bin/env python3
"""Counters of the LAST dispatch of every render kernel in a rocprofv3 --pmc output directory (earlier dispatches are the
warm-up frames that leave tile costs and the heavy-first order):  python tools/pmc_last.py DIR [--json]"""
import csv
import glob
import json
import os
import sys

out = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith(("rt_trace", "rt_sched"))]
    last = {}
    for r in rows:
        k = r["Kernel_Name"]
        last[k] = max(last.get(k, -1), int(r["Dispatch_Id"]))
    for r in rows:
        if int(r["Dispatch_Id"]) == last[r["Kernel_Name"]]:
            d = out.setdefault(r["Kernel_Name"], {})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            d["_launches_seen"] = len({int(q["Dispatch_Id"]) for q in rows if q["Kernel_Name"] == r["Kernel_Name"]})
for k, d in out.items():
    if "SQ_INSTS_VALU" in d and "SQ_THREAD_CYCLES_VALU" in d:
        d["active_lane_frac"] = round(d["SQ_THREAD_CYCLES_VALU"] / (64.0 * d["SQ_INSTS_VALU"]), 4)
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac"] = round(d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"], 4)
if "--json" in sys.argv:
    print(json.dumps(out))
else:
    for k in sorted(out):
        for c in sorted(out[k]):
            v = out[k][c]
            print(f"{k:32s} {c:26s} {v:.6g}" if isinstance(v, float) else f"{k:32s} {c:26s} {v}")
