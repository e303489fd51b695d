#!/usr/bin/env python3
"""Where a workgroup of the one-launch Poisson passes spends its time.  Needs a debug build of the library
(ns_poisson_small.hip compiled with -DNSK_SMALL_TRACE, see tools/trace_small.sh) named by NSK_LIB: every
workgroup sums the time of each phase (100 MHz clock); this prints the averages per pass and per exchange.

    NSK_LIB=tools/dbg/libnsk_smalltrace.so python tools/trace_small.py [nx ny passes halo rows]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params

a = [int(x) for x in sys.argv[1:]]
nx, ny, passes, halo, rows = (a + [153, 255, 152, 0, 0][len(a):])[:5]
p = ns_params("NavierStokes", nx=nx, ny=ny)
ld = _pitch(nx)
k = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
k.set_option("poisson_small_halo", halo)
k.set_option("poisson_small_rows", rows)
k.set_option("poisson_small_cols", rows)
k.set_option("poisson_div", int(os.environ.get("DIV", "2")))
g = torch.Generator(device="cuda").manual_seed(1)
Pr, dP, dv = (torch.rand(ny, ld, dtype=torch.float64, device="cuda", generator=g) for _ in range(3))
cp = c_params(p, ld_pr=ld, dp_embedded=1)
for rep in range(3):
    Pr.uniform_(generator=g), dP.uniform_(generator=g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k.ns_poisson_iters(Pr.data_ptr(), dP.data_ptr(), dv.data_ptr(), C.byref(cp), passes)
    k.sync()
    dt = time.perf_counter() - t0
print(f"{nx} x {ny}, {passes} passes in one launch: {dt*1e6:.1f} us = {dt/passes*1e6:.2f} us per pass; "
      f"tile rows {k.get_option('poisson_rows_used')}")
out = (C.c_ulonglong * (8 * 256))()
fn = k.lib.nsk_small_trace
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
n = C.c_int(0)
assert fn(k.handle, out, 256, C.byref(n)) == 0
t = np.array(out[: 8 * n.value], dtype=np.float64).reshape(n.value, 8) / 100.0     # us, last call
names = ["inputs -> LDS / registers", "passes", "outer cells out, drain, word", "wait for the neighbours", "halo cells in", "results out"]
print(f"{n.value} workgroups; per call (us): average, minimum, maximum over the workgroups")
for i, nm in enumerate(names):
    print(f"  {nm:30s} {t[:, i].mean():9.2f} {t[:, i].min():9.2f} {t[:, i].max():9.2f}")
print(f"  {'total':30s} {t[:, :6].sum(axis=1).mean():9.2f}")
