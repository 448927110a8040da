#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats of `bench.py --train` -> per-kernel table (library kernels) with per-step times.

    python3 tools/summarize_train_prof.py <dir>/<host>/<pid>_kernel_stats.csv <bench.log> profiles/<tag>_train_kernel_stats.md [spinup warmup steps]
"""
import csv, json, sys


def main(stats_csv, bench_log, out_md, spin=100, warm=5, steps=20):
    rows = []
    for r in csv.DictReader(open(stats_csv)):
        n = r["Name"]
        short = n.replace("void ", "").replace("ake_k::", "").split("(")[0]
        rows.append((short, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                     "ake_k::" in n or "adam_step" in n or "general_step" in n))
    bench = [l for l in open(bench_log, errors="ignore") if l.startswith("{")][-1]
    total_steps = spin + warm + 2 * steps       # spin-up + warm-up + timed + the in-bench pass with the event timers
    lib = [r for r in rows if r[4]]
    with open(out_md, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary of the training bench ({stats_csv.split('/')[-1]})\n\n")
        f.write(f"Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --train --steps {steps} --warmup {warm} --no-cpu-baseline`\n")
        f.write(f"({total_steps} training steps in the process: {spin} spin-up + {warm} warm-up + {steps} timed + {steps} with the in-bench event timers; per-step = total / {total_steps})\n\n")
        f.write("| kernel | calls | total ms | avg us | ms per step |\n|---|---|---|---|---|\n")
        for r in sorted(lib, key=lambda r: -r[2])[:45]:
            f.write(f"| `{r[0]}` | {r[1]} | {r[2]:.2f} | {r[3]:.1f} | {r[2] / total_steps:.3f} |\n")
        tl = sum(r[2] for r in lib)
        to = sum(r[2] for r in rows if not r[4])
        f.write(f"\nlibrary kernels: {tl:.1f} ms = {tl / total_steps:.3f} ms per step; torch / runtime kernels (zero_grad, loss scaling, copies, data synthesis): "
                f"{to:.1f} ms = {to / total_steps:.3f} ms per step\n\n")
        f.write("Bench line of the same (profiled) run:\n\n```\n" + bench.strip() + "\n```\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], *[int(v) for v in sys.argv[4:7]])
