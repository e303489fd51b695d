#!/usr/bin/env python3
"""Times the multi-pass Poisson kernel at 2559 x 4095 for every passes-per-launch K and a few
rows-per-task settings.  Prints one line per setting: µs per pass and T_eff."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--ks", default="1,2,3,4,5,6,7,8")
ap.add_argument("--rows", default="0")
ap.add_argument("--launches", type=int, default=200)
ap.add_argument("--div", type=int, default=-1)
ap.add_argument("--edge", default="0")
ap.add_argument("--xcd", default="1")
ap.add_argument("--ny", type=int, default=0, help="grid rows (default: the script's 256n-1)")
ap.add_argument("--dense", action="store_true", help="dense (unaligned) storage instead of pitched")
a = ap.parse_args()

p = ns_params("NavierStokes_Re6", n=a.n)
if a.ny:    # another height with the scalars of the script's grid (same divisors, same division form)
    p = p.with_(ny=a.ny)
nx, ny = p.nx, p.ny
ld = nx if a.dense else _pitch(nx)
torch.cuda.set_device(0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cpu").manual_seed(1)
def rnd():
    t = torch.zeros((ny, ld), dtype=torch.float64)
    t[:, :nx] = torch.randn((ny, nx), generator=g, dtype=torch.float64) * 1e-2
    return t.cuda()
Pr, dv = rnd(), rnd()
if a.dense:
    dP = torch.zeros((ny - 2, nx - 2), dtype=torch.float64, device="cuda")
    cp = c_params(p)
else:
    dP = torch.zeros((ny, ld), dtype=torch.float64, device="cuda")
    cp = c_params(p, ld_pr=ld, dp_embedded=1)
ptr = lambda t: C.c_void_p(t.data_ptr())
# the clocks of an idle GPU take a while to come up: about 0.2 s of work before any timing
ctx.ns_poisson_iters(ptr(Pr), ptr(dP), ptr(dv), C.byref(cp), 8000)
torch.cuda.synchronize()
for k in [int(x) for x in a.ks.split(",")]:
    for rows, edge, xcd in [(int(x), int(e), int(c)) for x in a.rows.split(",") for e in a.edge.split(",")
                            for c in a.xcd.split(",")]:
        ctx.set_option("poisson_k", k)
        ctx.set_option("poisson_edge_percent", edge)
        ctx.set_option("poisson_xcd", xcd)
        ctx.set_option("poisson_rows", rows)
        ctx.set_option("poisson_div", a.div)
        npass = a.launches * (k or 5)      # K=0: the library chooses; time a fixed number of passes
        ctx.ns_poisson_iters(ptr(Pr), ptr(dP), ptr(dv), C.byref(cp), 2 * (k or 5))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.ns_poisson_iters(ptr(Pr), ptr(dP), ptr(dv), C.byref(cp), npass)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / npass
        ku, ru = ctx.get_option("poisson_k"), ctx.get_option("poisson_rows_used")
        print(f"K={k} rows={rows:4d} edge={edge:3d} xcd={xcd} (used K={ku} rows={ru})  {us:8.2f} us/pass  launch {us*ku:8.1f} us  T_eff {40.0*nx*ny/us/1e3:9.1f} GB/s", flush=True)
