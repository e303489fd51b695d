#!/usr/bin/env python3
"""Upper bound on what overlapping consecutive launches could gain: two independent Poisson problems of the benchmark
size on two streams of one GPU (their launches may overlap freely) against the same two run one after the other."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
p = ns_params("NavierStokes_Re6", n=int(os.environ.get("AB_N", "16")))
nx, ny, ld = p.nx, p.ny, _pitch(p.nx)
torch.cuda.set_device(0)
g = torch.Generator(device="cpu").manual_seed(1)
def rnd():
    t = torch.zeros((ny, ld), dtype=torch.float64)
    t[:, :nx] = torch.randn((ny, nx), generator=g, dtype=torch.float64) * 1e-2
    return t.cuda()
cp = c_params(p, ld_pr=ld, dp_embedded=1)
probs = []
for _ in range(2):
    st = torch.cuda.Stream()
    ctx = _lib.Context(0, st.cuda_stream)
    ctx.set_option("poisson_k", 6)
    Pr, dv = rnd(), rnd()
    probs.append((st, ctx, Pr, torch.zeros_like(Pr), dv))
torch.cuda.synchronize()
def run(which, n):
    st, ctx, Pr, dP, dv = probs[which]
    ctx.ns_poisson_iters(C.c_void_p(Pr.data_ptr()), C.c_void_p(dP.data_ptr()), C.c_void_p(dv.data_ptr()), C.byref(cp), 6 * n)
for w in (0, 1):
    run(w, 10)
torch.cuda.synchronize()
import time
def timed(f):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
def sequential():
    run(0, launches)
    torch.cuda.synchronize()
    run(1, launches)
def together():                       # chunks of 20 launches alternately, so both queues stay filled
    for _ in range(launches // 20):
        run(0, 20)
        run(1, 20)
for name, f in (("one after the other", sequential), ("on two streams", together), ("one after the other", sequential), ("on two streams", together)):
    dt = timed(f)
    print(f"{name}: {dt / (2 * launches * 6) * 1e6:.2f} us per pass of one problem ({nx} x {ny})", flush=True)
