#!/usr/bin/env python3
"""Whole time steps at 2559 x 4095 with the Poisson loop cut to one check interval, for a
rocprofv3 kernel trace: shows what a step costs outside the Poisson loop -- two steps from the
reference's initial condition (fluid at rest: most dividends of the step kernels are zero) and two
from a moving field (random velocities)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes

p = ns_params("NavierStokes_Re6", n=16)
p = p.with_(niter=p.nchk)
sim = NavierStokes(p)
for _ in range(2):
    it = sim.step()
torch.cuda.synchronize()
print("from rest: iters", it, "err", sim.err_evo)
g = torch.Generator(device="cpu").manual_seed(3)
for k in ("Vx", "Vy"):
    t = getattr(sim, k).t
    t.copy_(torch.randn(t.shape, generator=g, dtype=torch.float64) * 0.3)
for _ in range(2):
    it = sim.step()
torch.cuda.synchronize()
print("moving field: iters", it)
