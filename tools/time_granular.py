#!/usr/bin/env python3
"""µs per Poisson pass when the reference's three call sites (update_dPrdτ!, update_Pr!,
set_bc_Pr!) are issued one launch each, as a kernel-by-kernel port of the driver would, on the
benchmark grid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes

p = ns_params("NavierStokes_Re6", n=16)
sim = NavierStokes(p, mode="granular")
k, f, nx, ny = sim.ctx, sim, p.nx, p.ny


def passes(n):
    for _ in range(n):
        k.update_dPrdtau(f.Pr.ptr, f.dPrdtau.ptr, f.divV.ptr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, nx, ny)
        k.update_Pr(f.Pr.ptr, f.dPrdtau.ptr, p.dtau, nx, ny)
        k.set_bc_Pr(f.Pr.ptr, 0.0, nx, ny, int(p.full_rows), 0, ny)


passes(50)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
passes(300)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 300
print(f"kernel-granular: {us:.1f} us per pass, T_eff {40.0*nx*ny/us/1e3:.0f} GB/s")
