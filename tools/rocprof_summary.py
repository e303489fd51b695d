#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd databases (kernel trace and/or PMC passes) as text.

  python tools/rocprof_summary.py gpurun_out/k4/k4_results.db [more.db ...] > profiles/xyz.txt
FETCH_SIZE / WRITE_SIZE are reported raw (KB) and corrected for gfx950: FETCH_SIZE counts
64 B per 128-byte request of a wide streaming read, i.e. half the bytes; WRITE_SIZE is exact.
"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    cur = db.cursor()
    print(f"== {path}")
    rows = cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        "from kernels group by name order by sum(end-start) desc limit 12").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'%':>6}  kernel")
    for n, c, s, a, mn, mx in rows:
        print(f"{c:6d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*s/tot:6.1f}  {n[:110]}")
    try:
        rows = cur.execute(
            "select kernel_name, counter_name, avg(value), count(*), max(vgpr_count), max(accum_vgpr_count), "
            "max(sgpr_count), max(lds_block_size) from counters_collection group by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        rows = []
    for kn, cn, v, c, vg, ag, sg, lds in rows:
        extra = ""
        if cn == "FETCH_SIZE":
            extra = f"  -> {2*v*1024/1e6:.1f} MB read per launch (x2 gfx950 correction)"
        if cn == "WRITE_SIZE":
            extra = f"  -> {v*1024/1e6:.1f} MB written per launch"
        print(f"  PMC {cn:14s} avg {v:16.1f} over {c} launches{extra}  [vgpr {vg} agpr {ag} sgpr {sg} lds {lds}]  {kn[:70]}")
