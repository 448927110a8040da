#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch.

    python3 tools/pmc_summary.py <..._counter_collection.csv> [substring filter]
"""
import csv
import sys
from collections import defaultdict


def main(path, flt=""):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if flt and flt not in name:
                continue
            short = name.replace("void ", "").replace("ake_k::", "").split("(")[0][:60]
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, ctrs in acc.items():
        n = max(len(v) for v in ctrs.values())
        print(f"== {k}  ({n} dispatches)")
        for c, v in sorted(ctrs.items()):
            print(f"   {c:32s} mean {sum(v) / len(v):16.1f}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
