#!/usr/bin/env python3
"""How often the FMA division of the Poisson kernel hands over to IEEE division in a real run:
whole reference time steps from the initial condition (needs a library built with
POISSON_FLAGS+=-DNSK_POISSON_TRACE=4096)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import NavierStokes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = ns_params("NavierStokes_Re6", n=n)
sim = NavierStokes(p)
sim.ctx.set_option("poisson_k", 5)
fn = _lib.load().nsk_debug_counters_k5
buf = (C.c_ulonglong * 4)()
fn(buf)
for s in range(steps):
    it = sim.step()
    torch.cuda.synchronize()
    assert fn(buf) == 0
    tasks, redone = buf[0], buf[1]
    print(f"step {s+1}: {it} passes; tasks {tasks}, run again with IEEE division {redone} "
          f"({100.0*redone/max(tasks,1):.4f} %)", flush=True)
