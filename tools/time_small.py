#!/usr/bin/env python3
"""Whole time steps of NavierStokes.jl at its default size (153 x 255, the reference's own
CPU-runnable configuration): seconds per step and µs per Poisson pass, for several passes per
launch; the oracle (1 thread) beside it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes

p = ns_params("NavierStokes")
for k, small, halo, rows in ((0, 1, 0, 0), (0, 1, 12, 16), (0, 1, 14, 16), (0, 1, 16, 16), (0, 1, 12, 20), (0, 1, 16, 20), (0, 1, 16, 24),
                            (0, 1, 20, 20), (0, 1, 12, 12), (0, 0, 0, 0)):
    sim = NavierStokes(p)
    sim.ctx.set_option("poisson_k", k)
    sim.ctx.set_option("poisson_small", small)      # whole runs of passes in one launch
    sim.ctx.set_option("poisson_small_halo", halo)
    sim.ctx.set_option("poisson_small_rows", rows)
    sim.ctx.set_option("poisson_small_cols", rows)
    sim.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    its = [sim.step() for _ in range(3)]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"K={k} one-launch={small} halo={halo} rows={rows}: {dt/3*1e3:8.2f} ms/step, passes {its}, {dt/sum(its)*1e6:6.2f} us/pass")
if "--cpu" in sys.argv:
    from oracle.ns_oracle import OracleSim
    ref = OracleSim(p)
    ref.step()
    t0 = time.perf_counter()
    its = [ref.step() for _ in range(2)]
    dt = time.perf_counter() - t0
    print(f"oracle: {dt/2*1e3:8.2f} ms/step, passes {its}, {dt/sum(its)*1e6:6.2f} us/pass")
