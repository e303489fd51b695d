#!/usr/bin/env python3
"""Where a task of the tiled lexicographic SOR spends its time.  Needs a debug build of the library
(bubble_sor_grid.hip compiled with -DNSK_SOR_GRID_TRACE, see tools/trace_sor_grid.sh) named by NSK_LIB:
every workgroup sums the time of each phase of its tasks (100 MHz clock); this prints the averages per task.

    NSK_LIB=tools/dbg/libnsk_sortrace.so python tools/trace_sor_grid.py [n] [sweeps] [rows]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd.bubble import DNS, DNSParams

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 0
q = DNSParams(nx=n, ny=n, dt=0.00125 * (32 / n) ** 2, pressure_solver="redblack", maxit=50, maxerr=0.0)
S = DNS(q)
S.step()
k = S.ctx
k.set_option("sor_grid_rows", rows)
sw = C.c_int(0)
for it in (2, sweeps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k.bub_solve_pressure(S.p.ptr, S.tmp1.ptr, S.tmp2.ptr, S.rhot.ptr, q.beta, q.dx, q.dy, n, n, it, 0.0, C.byref(sw), None)
    dt = time.perf_counter() - t0
print(f"{n} x {n}, {sweeps} sweeps: {dt*1e3:.2f} ms = {dt/sweeps*1e6:.1f} us per sweep")
nb = 2048
out = (C.c_ulonglong * (8 * nb))()
fn = k.lib.nsk_sor_grid_trace
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
assert fn(k.handle, out, nb) == 0
a = np.array(out[:], dtype=np.float64).reshape(nb, 8)
a = a[a[:, 7] > 0]
tasks = a[:, 7].sum()
names = ["between tasks (counter, skipped numbers)", "coefficients -> LDS", "waits", "acquire + unknown -> LDS", "sweep over the tile",
         "stores + maximum + drain", "flag / judgement"]
print(f"{len(a)} workgroups ran {int(tasks)} tasks; per task (us), average over all workgroups:")
for i, nm in enumerate(names):
    print(f"  {nm:42s} {a[:, i].sum() / tasks / 100.0:8.2f}")
print(f"  {'total':42s} {a[:, :7].sum() / tasks / 100.0:8.2f}")
