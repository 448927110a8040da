#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace sqlite database (rocprofv3 7.x writes <dir>/<host>/<pid>_results.db):
    python3 tools/prof_db_summary.py gpurun_out/c3prof/*/*_results.db [rows]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, count(*), sum(end-start)/1e6 from kernels group by name order by 3 desc"))
print(f"kernels {sum(r[1] for r in rows)}  total {sum(r[2] for r in rows):.2f} ms")
for name, n, ms in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{n:7d} {ms:9.2f} ms {ms / n * 1000:8.1f} us  {name[:100]}")
