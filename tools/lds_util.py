#!/usr/bin/env python3
"""LDS / VALU activity per kernel from one rocprofv3 --pmc pass (SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE) taken together with --kernel-trace:

    python3 tools/lds_util.py <dir with */*_counter_collection.csv and */*_kernel_trace.csv> [out.md]

SQ_LDS_IDX_ACTIVE = cycles the LDS arrays work (summed over the CUs), SQ_LDS_BANK_CONFLICT = the part of them that is conflict replay;
LDS busy = SQ_LDS_IDX_ACTIVE / (launch duration x the clock the launch ran at x 256 CUs) is not computed (the clock under the profiler
is not known per launch): the table gives the raw sums per launch and the ratios that need no clock."""
import csv, glob, sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
per = defaultdict(lambda: defaultdict(dict))
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"]
    if "ake_k::" not in name and "cqt_" not in name:
        continue
    short = name.replace("void ", "").replace("ake_k::", "").split("(")[0][:48]
    per[short + "|grid=" + r["Grid_Size"]][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
rows = []
for k, disp in per.items():
    n = len(disp)
    mean = lambda c: sum(v.get(c, 0.0) for v in disp.values()) / n
    ns = sum(dur.get(i, 0) for i in disp) / n
    if ns < 8000:
        continue
    idx, conf = mean("SQ_LDS_IDX_ACTIVE"), mean("SQ_LDS_BANK_CONFLICT")
    rows.append((ns, k, n, mean("SQ_INSTS_LDS"), idx, conf / idx if idx else 0.0, idx / mean("SQ_INSTS_LDS") if mean("SQ_INSTS_LDS") else 0.0,
                 mean("SQ_INSTS_VALU"), idx / (ns * 1e-9 * 256) / 1e9))
rows.sort(reverse=True)
out = ["| kernel (grid) | launches | avg us (under the profiler) | LDS instructions | LDS array cycles | of them bank-conflict replay | cycles per LDS instruction | VALU instructions | LDS array G-cycles per second and CU |",
       "|---|---|---|---|---|---|---|---|---|"]
for ns, k, n, il, idx, cf, cpi, iv, rate in rows:
    out.append(f"| `{k}` | {n} | {ns / 1e3:.1f} | {il / 1e6:.2f} M | {idx / 1e6:.1f} M | {cf:.3f} | {cpi:.1f} | {iv / 1e6:.2f} M | {rate:.2f} |")
text = "\n".join(out)
print(text)
if len(sys.argv) > 2:
    with open(sys.argv[2], "w") as f:
        f.write("# LDS activity of the network kernels (rocprofv3 --pmc, one pass)\n\n"
                "Command: `rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU "
                "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 tools/net_only.py 3`\n"
                "(256 clips x 76 frames).  The last column divides the LDS array cycles by the launch duration and the 256 CUs: at the ~2 GHz these\n"
                "kernels hold, a value near 2 means the LDS arrays never rest.\n\n" + text + "\n")
