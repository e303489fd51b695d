#!/usr/bin/env python3
"""Two whole time steps at 2559 x 4095 with the Poisson loop cut to one check interval, for a
rocprofv3 kernel trace: shows what a step costs outside the Poisson loop."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes

p = ns_params("NavierStokes_Re6", n=16)
p = p.with_(niter=p.nchk)
sim = NavierStokes(p)
for _ in range(2):
    it = sim.step()
torch.cuda.synchronize()
print("iters", it, "err", sim.err_evo)
