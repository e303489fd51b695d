#!/usr/bin/env python3
"""Timing of the DNS (code1) kernels at a scaled grid (default 1024 x 1024): per-kernel launch
time, achieved GB/s on the arrays each kernel must move, and the same kernels of the CPU
oracle for comparison.  Prints a table; rank-0 single GPU."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd.bubble import DNS, DNSParams

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--cpu", action="store_true", help="also time the oracle's kernels (single thread)")
a = ap.parse_args()
n = a.n
q = DNSParams(nx=n, ny=n, dt=0.00125 * (32 / n) ** 2, pressure_solver="redblack", maxit=50, maxerr=0.0)
sim = DNS(q)
k, nx, ny = sim.ctx, n, n
cells = (n + 2) * (n + 2)


def timed(fn, reps=a.reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


sim.step()
rows = []
S = sim
sw = C.c_int(0)
# (name, callable, arrays moved: reads + writes in units of one (n+2)^2 array)
K = [
    ("set_tangential", lambda: k.bub_set_tangential(S.u.ptr, S.v.ptr, 0.0, 0.0, 0.0, 0.0, nx, ny), 0),
    ("temp_uvel+temp_vvel", lambda: k.bub_temp_vel(S.ut.ptr, S.vt.ptr, S.u.ptr, S.v.ptr, S.rho.ptr, q.dt, q.dx, q.dy, q.gx, q.gy, q.m0, nx, ny), 5),
    ("set_boundary", lambda: k.bub_set_boundary(S.rhot.ptr, S.rho.ptr, 1000.0, nx, ny), 2),
    ("compute_sources", lambda: k.bub_compute_sources(S.tmp1.ptr, S.tmp2.ptr, S.rhot.ptr, S.ut.ptr, S.vt.ptr, q.dt, q.dx, q.dy, nx, ny), 5),
    ("red+black sweep (1 it)", lambda: k.bub_solve_pressure_rb(S.p.ptr, S.tmp1.ptr, S.tmp2.ptr, S.rhot.ptr, q.beta, q.dx, q.dy, nx, ny, 8, 0.0, C.byref(sw), None), 5 * 8),
    ("pseudo-transient (1 it)", lambda: k.bub_pt_pressure(S.p.ptr, S.dPdtau.ptr, S.tmp1.ptr, S.rhot.ptr, q.dtau, q.damp, q.dx, q.dy, nx, ny, 8, 0.0, C.byref(sw), None), 6 * 8),
    ("correct_uvel+vvel", lambda: k.bub_correct_vel(S.u.ptr, S.v.ptr, S.ut.ptr, S.vt.ptr, S.p.ptr, S.rho.ptr, q.dt, q.dx, q.dy, nx, ny), 6),
    ("advect_density", lambda: k.bub_advect_density(S.rho.ptr, S.rho_o.ptr, S.u.ptr, S.v.ptr, q.dt, q.dx, q.dy, q.m0, nx, ny), 4),
    ("velocity (corner avg)", lambda: k.bub_velocity(S.uu.ptr, S.vv.ptr, S.u.ptr, S.v.ptr, nx, ny), 4),
]
print(f"DNS kernels at {n} x {n} (one (n+2)^2 array = {cells*8/1e6:.1f} MB), {a.reps} repetitions")
print(f"{'kernel':28s} {'us/call':>10s} {'arrays':>7s} {'GB/s':>9s}")
for name, fn, arrays in K:
    reps = a.reps if "it)" not in name else max(2, a.reps // 8)
    us = timed(fn, reps)
    per = us / 8 if "it)" in name else us
    arr = arrays / 8 if "it)" in name else arrays
    gbs = arr * cells * 8 / per / 1e3 if arr else float("nan")
    print(f"{name:28s} {per:10.1f} {arr:7.1f} {gbs:9.0f}")
# lexicographic SOR (the reference's order): tiles on the whole chip with several sweeps in flight, against one
# workgroup that sweeps the anti-diagonals one sweep at a time
def sor(sweeps):
    k.bub_solve_pressure(S.p.ptr, S.tmp1.ptr, S.tmp2.ptr, S.rhot.ptr, q.beta, q.dx, q.dy, nx, ny, sweeps, 0.0, C.byref(sw), None)


for rows in (16, 32, 64):
    k.set_option("sor_grid_rows", rows)
    sor(2)                                            # the ring of copies is allocated on the first call
    for sweeps in (4, 50, 200):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sor(sweeps)
        dt_ = time.perf_counter() - t0
        print(f"{'lexicographic SOR sweep':28s} {dt_/sweeps*1e6:10.1f}   (tiles of 64 x {rows} on the chip, {sweeps} sweeps in one solve: {dt_*1e3:.2f} ms)")
k.set_option("sor_grid_rows", 0)
k.set_option("sor_grid", 0)
sor(1)
torch.cuda.synchronize()
t0 = time.perf_counter()
sor(4)
print(f"{'lexicographic SOR sweep':28s} {(time.perf_counter()-t0)/4*1e6:10.1f}   (one workgroup, anti-diagonal wavefront)")
k.set_option("sor_grid", 1)
if a.cpu:
    from oracle.bubble_oracle import OracleDNS
    o = OracleDNS(q)
    o.step()
    bo = o.o

    def cpu(fn, reps=3):
        fn()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t) / reps * 1e6
    print("CPU oracle (1 thread):")
    print(f"{'temp_uvel+temp_vvel':28s} {cpu(lambda: (bo.temp_uvel(o.ut, o.u, o.v, o.rho, q.dt, q.dx, q.dy, q.gx, q.m0, nx, ny), bo.temp_vvel(o.vt, o.u, o.v, o.rho, q.dt, q.dx, q.dy, q.gy, q.m0, nx, ny))):10.1f}")
    print(f"{'compute_sources':28s} {cpu(lambda: bo.compute_sources(o.tmp1, o.tmp2, o.rhot, o.ut, o.vt, q.dt, q.dx, q.dy, nx, ny)):10.1f}")
    print(f"{'red+black sweep (1 it)':28s} {cpu(lambda: bo.solve_pressure_rb(o.p, o.tmp1, o.tmp2, o.rhot, q.beta, q.dx, q.dy, nx, ny, 2, 0.0, None))/2:10.1f}")
    print(f"{'lexicographic SOR sweep':28s} {cpu(lambda: bo.solve_pressure(o.p, o.tmp1, o.tmp2, o.rhot, q.beta, q.dx, q.dy, nx, ny, 2, 0.0, None))/2:10.1f}")
    print(f"{'advect_density':28s} {cpu(lambda: bo.advect_density(o.rho, o.rho_o, o.u, o.v, q.dt, q.dx, q.dy, q.m0, nx, ny)):10.1f}")
