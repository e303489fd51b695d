#!/usr/bin/env python3
"""How often does a call of one-pass launches on a 127 x 1574 grid of subnormal values (every task run twice)
differ from the oracle?  repro_rerun.py [reps] [count] [name=value options ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from navierstokes_amd import _lib, ns_params
from oracle.ns_oracle import Oracle
from navierstokes_amd.ns import _pitch, c_params
from util import Dev

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
count = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nx, ny = int(os.environ.get("NX", 127)), int(os.environ.get("NY", 1574))
scale = float(os.environ.get("SCALE", 1e-310))
p = ns_params("NavierStokes", nx=nx, ny=ny)
rng = np.random.default_rng(903)
def field(n, m, s):
    f = np.asfortranarray(rng.standard_normal((n, m)) * s)
    f[rng.random((n, m)) < 0.3] = 0.0
    return f
Pr0, dP0, dv = field(nx, ny, scale), field(nx - 2, ny - 2, scale), field(nx, ny, scale * 10)
oracle = Oracle()
Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
with np.errstate(all="ignore"):
    for _ in range(count):
        oracle.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
        oracle.update_Pr(Pr, dP, p.dtau)
        oracle.set_bc_Pr(Pr, 0.0, True)
torch.cuda.set_device(0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
ctx.set_option("poisson_k", int(os.environ.get("K", 1)))
ctx.set_option("poisson_small", 0)
for o in sys.argv[3:]:
    n, v = o.split("=")
    ctx.set_option(n, int(v))
ld = _pitch(nx)
cp = c_params(p.with_(full_rows=True), ld_pr=ld, dp_embedded=1, ny=ny)
emb = np.full((nx, ny), np.nan, order="F")
emb[1:-1, 1:-1] = dP0
bad, rows_hist = 0, {}
for rep in range(reps):
    dPr, dDv, dDp = Dev(Pr0, ld, pad_value=np.nan), Dev(dv, ld, pad_value=np.nan), Dev(emb, ld, pad_value=np.nan)
    ctx.ns_poisson_iters(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count)
    g = dPr.get()
    d = np.argwhere(np.ascontiguousarray(g).view(np.uint64) != np.ascontiguousarray(Pr).view(np.uint64))
    if len(d):
        bad += 1
        for j in sorted(set(int(j) for _, j in d)):
            cols = [int(i) for i, jj in d if jj == j]
            key = "strip1" if min(cols) >= 122 else ("strip0" if max(cols) <= 125 else "both")
            rows_hist.setdefault(key, []).append(j)
print(f"{bad} of {reps} calls differ; K={ctx.get_option('poisson_k')} rows={ctx.get_option('poisson_rows_used')} div={ctx.get_option('poisson_div_used')}; "
      + "; ".join(f"{k}: {len(v)} rows, e.g. {v[:12]}" for k, v in rows_hist.items()))
