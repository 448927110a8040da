#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats kernel_stats.csv to the library's own kernels (torch's data-generation
kernels are dropped, names shortened) and write it next to the bench line it was collected with.

    python3 tools/summarize_prof.py gpurun_out/<run>/<host>/<pid>_kernel_stats.csv bench.log profiles/<name>.md
"""
import csv
import sys


def main(stats_csv, bench_log, out_md):
    rows = []
    with open(stats_csv) as f:
        for r in csv.DictReader(f):
            n = r["Name"]
            if "ake_k::" in n or "cqt_" in n.split("(")[0] or "fill_i64" in n or "adam_step" in n:
                short = n.replace("void ", "").replace("ake_k::", "").split("(")[0]
                rows.append((short, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                             float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    tot = sum(r[2] for r in rows)
    bench = [l for l in open(bench_log) if l.startswith("{")]
    with open(out_md, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary ({stats_csv.split('/')[-1]})\n\n")
        f.write("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --sustained-seconds 0 --pipelined-streams 0 --spinup-seconds 0`\n")
        f.write("(library kernels only; torch kernels that synthesise the input clips are omitted; 12 forward passes = 5 timed + 5 all-kernel-events + 2 warm-up)\n\n")
        f.write("| kernel | calls | total ms | avg us | min us | max us | % of library time |\n|---|---|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -r[2]):
            f.write(f"| `{r[0]}` | {r[1]} | {r[2]:.3f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {100 * r[2] / tot:.1f} |\n")
        f.write(f"\nlibrary kernel time total: {tot:.3f} ms\n\n")
        if bench:
            f.write("bench.py line of the same run:\n\n```json\n" + bench[-1].strip() + "\n```\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
