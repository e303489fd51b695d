#!/usr/bin/env python3
"""Writes tests/golden/ns_oracle_21x37_3steps.npz: the state after 3 time steps of
scripts/NavierStokes.jl's problem on a 21 x 37 grid, computed by THIS repository's CPU oracle
(oracle/ns_oracle.c).  The reference cannot run here (Julia + ParallelStencil.jl absent) and
ships no outputs, so this fixture pins the oracle against itself over time, not against the
reference."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from navierstokes_amd import ns_params  # noqa: E402
from oracle.ns_oracle import OracleSim  # noqa: E402

nx, ny, steps = 21, 37, 3
s = OracleSim(ns_params("NavierStokes", nx=nx, ny=ny))
iters = [s.step() for _ in range(steps)]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ns_oracle_21x37_3steps.npz"),
                    nx=nx, ny=ny, steps=steps, iters=np.array(iters), Pr=s.Pr, Vx=s.Vx, Vy=s.Vy, C=s.C,
                    dPrdtau=s.dPrdtau)
print("iters", iters)
