#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace result database:
    prof_summary.py <results.db> <time steps the run made>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch_")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol_")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
nm = "display_name" if "display_name" in cols else "kernel_name"
rows = db.execute(f"select s.{nm}, count(*), sum(d.end-d.start), min(d.start), max(d.end) from {kd} d join {ks} s "
                  f"on d.kernel_id=s.id group by s.{nm} order by 3 desc").fetchall()
t0, t1 = min(r[3] for r in rows), max(r[4] for r in rows)
print(f"span {(t1 - t0) / 1e3 / steps:.1f} us/step, kernel time {sum(r[2] for r in rows) / 1e3 / steps:.1f} us/step, "
      f"{sum(r[1] for r in rows) / steps:.1f} launches/step")
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r[0])[:64]
    print(f"  {n:64s} {r[1] / steps:5.1f}/step  avg {r[2] / r[1] / 1e3:7.2f} us  {r[2] / 1e3 / steps:7.1f} us/step")
