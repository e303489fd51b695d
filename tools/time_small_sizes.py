#!/usr/bin/env python3
"""The one-launch Poisson passes against the launch-per-K-passes kernel on grids of several sizes: us per pass
for runs of 120 passes (what lies between two residual checks), for the default tiles and for chosen frame
widths / tile sizes.    python tools/time_small_sizes.py [nx ny ...]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params

a = [int(x) for x in sys.argv[1:]]
grids = list(zip(a[::2], a[1::2])) or [(63, 105), (153, 255), (159, 255), (319, 511), (479, 767), (639, 1023), (799, 1279), (255, 2047)]
k = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
count, reps = int(os.environ.get('COUNT', 120)), 20
for nx, ny in grids:
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    ld = _pitch(nx)
    Pr, dP, dv = (torch.rand(ny, ld, dtype=torch.float64, device="cuda", generator=g) * 1e-2 for _ in range(3))
    cp = c_params(p, ld_pr=ld, dp_embedded=1)
    line = f"{nx:5d} x {ny:5d}:"
    for small, halo, tile in ((0, 0, 0), (1, 0, 0), (1, 4, 0), (1, 6, 0), (1, 8, 0), (1, 10, 0), (1, 12, 0), (1, 14, 0), (1, 16, 0)):
        for name, v in (("poisson_small", small), ("poisson_small_halo", halo), ("poisson_small_cols", tile), ("poisson_small_rows", tile)):
            k.set_option(name, v)
        k.ns_poisson_iters(Pr.data_ptr(), dP.data_ptr(), dv.data_ptr(), C.byref(cp), count)
        k.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            k.ns_poisson_iters(Pr.data_ptr(), dP.data_ptr(), dv.data_ptr(), C.byref(cp), count)
        k.sync()
        us = (time.perf_counter() - t0) / (reps * count) * 1e6
        one = k.get_option("poisson_k") == count
        what = "launch per K" if not small else (f"H={halo or 'auto'} tile {k.get_option('poisson_rows_used')}" if one else f"H={halo}: n/a")
        line += f"  {what}: {us:5.2f}"
    print(line, flush=True)
