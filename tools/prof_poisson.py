#!/usr/bin/env python3
"""A short run of the multi-pass Poisson kernel for rocprofv3 (kernel trace or PMC passes)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=6)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--xcd", type=int, default=1)
ap.add_argument("--launches", type=int, default=20)
ap.add_argument("--copy", action="store_true", help="also run a plain device copy of the same footprint")
a = ap.parse_args()
p = ns_params("NavierStokes_Re6", n=16)
nx, ny = p.nx, p.ny
ld = _pitch(nx)
torch.cuda.set_device(0)
ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cpu").manual_seed(1)
def rnd():
    t = torch.zeros((ny, ld), dtype=torch.float64)
    t[:, :nx] = torch.randn((ny, nx), generator=g, dtype=torch.float64) * 1e-2
    return t.cuda()
Pr, dv = rnd(), rnd()
dP = torch.zeros((ny, ld), dtype=torch.float64, device="cuda")
cp = c_params(p, ld_pr=ld, dp_embedded=1)
ptr = lambda t: C.c_void_p(t.data_ptr())
ctx.set_option("poisson_k", a.k)
ctx.set_option("poisson_rows", a.rows)
ctx.set_option("poisson_xcd", a.xcd)
ctx.ns_poisson_iters(ptr(Pr), ptr(dP), ptr(dv), C.byref(cp), a.k * a.launches)
torch.cuda.synchronize()
if a.copy:
    src = torch.randn(int(2.5 * ny * ld), dtype=torch.float64, device="cuda")
    dst = torch.empty_like(src)
    for _ in range(3):
        dst.copy_(src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"copy {src.numel()*8/1e6:.0f} MB: {ms*1e3:.1f} us  {2*src.numel()*8/ms/1e6:.0f} GB/s (read+write)")
print("done")
