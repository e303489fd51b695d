#!/usr/bin/env python3
"""Turns rocprofv3 PMC passes over tools/prof_poisson.py into the entry of profiles/poisson_traffic.json
that bench.py reads (HBM bytes and vector-ALU instructions per launch of the multi-pass Poisson
kernel), stamped with the hash of the kernel source and build flags it was measured on.

  python tools/make_traffic.py --k 6 --grid 2559 4095 fetch.db write.db sq.db

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-byte request of a
wide streaming read, so it is doubled; WRITE_SIZE is exact."""
import argparse
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from navierstokes_amd._lib import poisson_kernel_hash

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, required=True)
ap.add_argument("--grid", type=int, nargs=2, required=True)
ap.add_argument("--round", type=int, default=3)
ap.add_argument("dbs", nargs="+")
a = ap.parse_args()

# the launch of K passes is k_poisson<K, ...> or, with its tasks in pairs, k_poisson_pairs<K, ...>
vals, kernel = {}, None
for form in ("k_poisson_pairs", "k_poisson"):
    like = f"%{form}<{a.k},%"
    for path in a.dbs:
        cur = sqlite3.connect(path).cursor()
        for name, avg, n in cur.execute(
                "select counter_name, avg(value), count(*) from counters_collection "
                "where kernel_name like ? group by counter_name", (like,)):
            vals[name] = (avg, n)
    if vals:
        kernel = form
        break
if kernel is None:
    raise SystemExit(f"no launch of k_poisson<{a.k}, ...> or k_poisson_pairs<{a.k}, ...> in the given databases")
# the kernel's duration in the pass that counted SQ_BUSY_CYCLES: cycles / duration = the clock it ran at
launch_us = None
for path in a.dbs:
    cur = sqlite3.connect(path).cursor()
    has = cur.execute("select count(*) from counters_collection where counter_name = 'SQ_BUSY_CYCLES'").fetchone()[0]
    if has:
        row = cur.execute("select avg(end - start) from kernels where name like ?", (like,)).fetchone()
        if row and row[0]:
            launch_us = row[0] / 1e3
need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU")
missing = [c for c in need if c not in vals]
if missing:
    raise SystemExit(f"counters missing from the given databases: {missing}")
entry = {"read_bytes": int(2 * vals["FETCH_SIZE"][0] * 1024), "write_bytes": int(vals["WRITE_SIZE"][0] * 1024),
         "insts_valu": int(vals["SQ_INSTS_VALU"][0]), "launches": vals["SQ_INSTS_VALU"][1],
         "grid": list(a.grid), "kernel_hash": poisson_kernel_hash(), "round": a.round}
if launch_us:
    entry["launch_us"] = round(launch_us, 3)
for extra in ("SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES"):
    if extra in vals:
        entry[extra.lower()] = int(vals[extra][0])
out = os.path.join(ROOT, "profiles", "poisson_traffic.json")
with open(out) as f:
    data = json.load(f)
data[f"{kernel}<{a.k}>"] = entry
with open(out, "w") as f:
    json.dump(data, f, indent=1)
print(json.dumps(entry))
