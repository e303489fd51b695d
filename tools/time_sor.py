#!/usr/bin/env python3
"""Time the lexicographic SOR pressure solve alone: fixed sweep counts (tolerance 0) on a random
32 x 32 problem, microseconds per solve and per barrier step of the several-sweeps-in-flight kernel."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import _lib

nx = ny = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.cuda.set_device(0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(1)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda()
shape = (nx + 2, ny + 2)
ptr = lambda t: C.c_void_p(t.data_ptr())
p0 = dev(rng.standard_normal(shape))
# coefficients of a real solve (the iteration contracts): densities 1..2 with the large value on the
# border, sources from random temporary velocities
r, rt = dev(1.0 + rng.random(shape)), dev(np.zeros(shape))
t1, t2 = dev(np.zeros(shape)), dev(np.zeros(shape))
ut, vt = dev(rng.standard_normal((nx + 1, ny + 2))), dev(rng.standard_normal((nx + 2, ny + 1)))
ctx.bub_set_boundary(ptr(rt), ptr(r), 1000.0, nx, ny)
ctx.bub_compute_sources(ptr(t1), ptr(t2), ptr(rt), ptr(ut), ptr(vt), 0.00125, 1.0 / nx, 1.0 / ny, nx, ny)
for pipe in (1, 0):
    ctx.set_option("sor_pipeline", pipe)
    for sweeps in (8, 50, 400):
        p = p0.clone()
        run = lambda: ctx.bub_solve_pressure(ptr(p), ptr(t1), ptr(t2), ptr(rt), 1.2, 1.0 / nx, 1.0 / ny, nx, ny, sweeps, 0.0, None, None)
        run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        steps = nx + ny - 1 + 2 * (sweeps - 1) if pipe else sweeps * (nx + ny - 1)
        print(f"{nx}x{ny} pipeline={pipe} {sweeps:4d} sweeps: {us:8.1f} us per solve, {us / steps:6.3f} us per barrier step ({steps} steps)")
