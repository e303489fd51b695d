#!/usr/bin/env python3
"""Steady-state cost of a time step of the two bubble drivers at the reference sizes: mean sweeps
per lexicographic SOR solve and microseconds per step, after the start-up transient."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd.bubble import Code3, Code3Params, DNS, DNSParams


def run(sim, warm, steps):
    for _ in range(warm):
        sim.step()
    torch.cuda.synchronize()
    t0, sweeps = time.time(), []
    for _ in range(steps):
        sweeps.append(sim.step())
    torch.cuda.synchronize()
    return np.mean(np.array(sweeps, dtype=float), axis=0), (time.time() - t0) / steps * 1e6


sw, us = run(DNS(DNSParams()), 20, 200)
print(f"DNS.jl 32x32: {sw:.1f} sweeps per pressure solve = {63 + 2 * (sw - 1):.0f} barrier steps of the pipeline, {us:.1f} us per time step")
sw, us = run(Code3(Code3Params()), 20, 100)
print(f"code3 32x32, Nf=100: sweeps per solve (pressure, density, pressure, density) {np.round(sw, 1)}, {us:.1f} us per time step")
for name, sim, n in (("DNS.jl", DNS(DNSParams()), 200), ("code3", Code3(Code3Params()), 100)):
    sim.run(20)
    torch.cuda.synchronize()
    t0 = time.time()
    sim.run(n)
    torch.cuda.synchronize()
    print(f"{name}: {(time.time() - t0) / n * 1e6:.1f} us per time step as a run of {n} steps in one call")
