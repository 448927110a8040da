#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command,
as the TCC counter slots require).  Units and gfx950 corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes:
both counters are in KiB; FETCH_SIZE counts 128-B requests as 64 B on gfx950 -> x2 (calibrated here on a known-size copy,
tools/probe/copy_probe.hip: x2.00 for 4-B and 16-B per lane loads); WRITE_SIZE is exact.

    python3 tools/pmc_traffic.py <FETCH counter_collection.csv> <WRITE counter_collection.csv> out.json

Launches are keyed by kernel name + grid size (the three 7x7 pitch convolutions share a template instance with the
pitch-class convolutions; the grid tells them apart)."""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            if not ("ake_k::" in name or name.startswith("cqt_") or "cqt_" in name.split("(")[0]):
                continue
            short = name.replace("void ", "").replace("ake_k::", "").split("(")[0]
            acc[f"{short}|grid={r['Grid_Size']}"].append(float(r["Counter_Value"]))
    return acc


def main(fetch_csv, write_csv, out):
    fe, wr = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, [0.0]); w = wr.get(k, [0.0])
        res[k] = {"dispatches": len(f), "fetch_bytes": round(2 * 1024 * sum(f) / len(f)), "write_bytes": round(1024 * sum(w) / len(w))}
        res[k]["hbm_bytes"] = res[k]["fetch_bytes"] + res[k]["write_bytes"]
    from bench import kernel_set_hash                        # the kernel sources these counters belong to: bench.py refuses a stale profile
    json.dump({"units": "bytes per launch (mean); fetch = 2 x FETCH_SIZE KiB (gfx950), write = WRITE_SIZE KiB",
               "kernel_set": kernel_set_hash(), "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k:60s} x{v['dispatches']:3d}  fetch {v['fetch_bytes']/1e6:9.2f} MB  write {v['write_bytes']/1e6:9.2f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
