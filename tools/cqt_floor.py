#!/usr/bin/env python3
"""What bounds the CQT stage, from counters (VERDICT r2 item 2's fallback deliverable): per kernel the measured time next to two floors,

  * vector issue:  SQ_INSTS_VALU wave-instructions x 4 cycles (a 64-wide wave on a 16-lane SIMD) / (1024 SIMDs x clock)
  * HBM traffic:   (FETCH_SIZE + WRITE_SIZE bytes, gfx950-corrected as tools/pmc_traffic.py does) / 5.5 TB/s (the copy rate this part sustains)

and the issue picture (share of wave cycles issuing / stalled on an instruction dependency / parked at a waitcnt or barrier).

    python3 tools/cqt_floor.py <pmc dir A> <pmc dir B> <pmc_traffic.json> [out.md]
    A: rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace
    B: rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
    (both on `python3 tools/cqt_only.py 3`: 256 clips of 15 s, the standalone transform)"""
import csv, glob, json, sys
from collections import defaultdict


def load(d):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    per = defaultdict(lambda: defaultdict(dict))
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"]
        if "cqt_" not in name:
            continue
        short = name.replace("void ", "").replace("ake_k::", "").split("(")[0].split("<")[0]
        per[short][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for k, disp in per.items():
        n = len(disp)
        m = defaultdict(float)
        for v in disp.values():
            for c, x in v.items():
                m[c] += x / n
        m["_us"] = sum(dur.get(i, 0) for i in disp) / n / 1e3
        out[k] = m
    return out


A, B = load(sys.argv[1]), load(sys.argv[2])
traffic = json.load(open(sys.argv[3]))
per_kernel = {}
for k, v in traffic.get("kernels", {}).items():
    per_kernel[k.split("|")[0].split("<")[0].replace("ake_k::", "")] = v
lines = ["| kernel | us (under the profiler) | VALU wave-instructions | vector-issue floor @ 2.4 / 2.0 GHz | HBM bytes (PMC) | traffic floor @ 5.5 TB/s | waves issuing / dependency-stalled / parked | LDS / vector-memory instructions | LDS array cycles (conflicts) |",
         "|---|---|---|---|---|---|---|---|---|"]
tot = [0.0, 0.0, 0.0, 0.0]
for k in sorted(A, key=lambda k: -A[k]["_us"]):
    a, b = A[k], B.get(k, defaultdict(float))
    valu = a["SQ_INSTS_VALU"]
    f24, f20 = valu * 4 / (1024 * 2.4e3), valu * 4 / (1024 * 2.0e3)
    byt = None
    for kk, v in per_kernel.items():
        if kk.startswith(k) or k.startswith(kk):
            byt = v.get("hbm_bytes")
    hb = byt / 5.5e6 if byt else float("nan")
    wq = b["SQ_WAVE_CYCLES"] or 1.0
    lines.append(f"| `{k}` | {a['_us']:.1f} | {valu / 1e6:.1f} M | {f24:.0f} / {f20:.0f} us | {byt / 1e6:.0f} MB | {hb:.0f} us | "
                 f"{b['SQ_ACTIVE_INST_ANY'] / wq:.2f} / {b['SQ_WAIT_INST_ANY'] / wq:.2f} / {b['SQ_WAIT_ANY'] / wq:.2f} | "
                 f"{a['SQ_INSTS_LDS'] / 1e6:.2f} M / {(a['SQ_INSTS_VMEM_RD'] + a['SQ_INSTS_VMEM_WR']) / 1e6:.2f} M | {b['SQ_LDS_IDX_ACTIVE'] / 1e6:.1f} M ({b['SQ_LDS_BANK_CONFLICT'] / 1e6:.1f} M) |"
                 if byt else f"| `{k}` | {a['_us']:.1f} | {valu / 1e6:.1f} M | {f24:.0f} / {f20:.0f} us | - | - | - | - | - |")
    if byt:
        tot[0] += a["_us"]; tot[1] += f24; tot[2] += hb; tot[3] += max(f24, hb)
text = "\n".join(lines)
text += (f"\n\nSum over the kernels: measured {tot[0]:.0f} us; vector-issue floors {tot[1]:.0f} us; traffic floors {tot[2]:.0f} us; "
         f"max(floor) per kernel, summed: {tot[3]:.0f} us = the least this TWO-KERNEL structure with THIS instruction count and THIS traffic could take; "
         "the stage's target (0.40 of 8 TB/s on 361 MB algorithmic) is 113 us.\n")
print(text)
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as f:
        f.write("# CQT stage: counter-based floors (rocprofv3 --pmc, two passes + the FETCH / WRITE passes of the bench)\n\n" + (__doc__ or "") + "\n\n" + text)
