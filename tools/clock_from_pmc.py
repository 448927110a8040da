#!/usr/bin/env python3
"""Effective shader clock per kernel = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md 'DVFS give-back').

    python3 tools/clock_from_pmc.py <dir with *_counter_collection.csv and *_kernel_trace.csv>
"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
acc = defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    ns, name = dur[r["Dispatch_Id"]]
    short = name.replace("void ", "").replace("ake_k::", "").split("(")[0][:50]
    acc[short].append((float(r["Counter_Value"]) / 8.0 / ns, ns))
for k, v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    ghz = [x[0] for x in v]
    print(f"{k:52s} n={len(v):4d} avg_us={sum(x[1] for x in v)/len(v)/1e3:8.1f}  clock GHz mean={sum(ghz)/len(ghz):.2f} min={min(ghz):.2f} max={max(ghz):.2f}")
