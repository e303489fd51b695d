#!/usr/bin/env python3
"""Time the multi-pass Poisson kernel of any build of libnsk.so (raw ctypes, only entry points every
build has), for A/B comparisons on one box:  ab_time.py path/to/libnsk.so [launches]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import ns_params
from navierstokes_amd.ns import _pitch, c_params

lib = C.CDLL(sys.argv[1])
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 300
p = ns_params("NavierStokes_Re6", n=int(os.environ.get("AB_N", "16")))
K = int(os.environ.get("AB_K", "6"))
nx, ny, ld = p.nx, p.ny, int(os.environ.get("AB_LD", "0")) or _pitch(p.nx)   # AB_LD = row pitch in doubles
torch.cuda.set_device(0)
h = C.c_void_p()
assert lib.nsk_create(0, C.byref(h)) == 0
assert lib.nsk_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
assert lib.nsk_set_option(h, b"poisson_k", K) == 0
ROWS = int(os.environ.get("AB_ROWS", "0"))
for opt in os.environ.get("AB_OPTS", "").split():        # AB_OPTS="name=value ..." (options this build has)
    name, v = opt.split("=")
    assert lib.nsk_set_option(h, name.encode(), int(v)) == 0, opt
assert lib.nsk_set_option(h, b"poisson_rows", ROWS) == 0
g = torch.Generator(device="cpu").manual_seed(1)
def rnd():
    t = torch.zeros((ny, ld), dtype=torch.float64)
    t[:, :nx] = torch.randn((ny, nx), generator=g, dtype=torch.float64) * 1e-2
    return t.cuda()
Pr, dv = rnd(), rnd()
dP = torch.zeros_like(Pr)
cp = c_params(p, ld_pr=ld, dp_embedded=1)
ptr = lambda t: C.c_void_p(t.data_ptr())
run = lambda n: lib.nsk_ns_poisson_iters(h, ptr(Pr), ptr(dP), ptr(dv), C.byref(cp), n)
assert run(60) == 0
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
assert run(6 * launches) == 0
e1.record()
torch.cuda.synchronize()
def used(name):
    v = C.c_int(-1)
    return v.value if lib.nsk_get_option(h, name, C.byref(v)) else v.value       # (-1: this build has no such option)
print(f"{os.path.basename(sys.argv[1])} {nx}x{ny} K={K or 'auto'} rows={ROWS or 'auto'} ld={ld} {os.environ.get('AB_OPTS', '')}: {e0.elapsed_time(e1) * 1e3 / (6 * launches):.2f} us/pass"
      f"   [ran K={used(b'poisson_k')} rows={used(b'poisson_rows_used')} pairs={used(b'poisson_pairs_used')}]")
