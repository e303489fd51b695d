#!/usr/bin/env python3
"""Per-task timeline of one launch of the multi-pass Poisson kernel (needs a library built with
POISSON_FLAGS+=-DNSK_POISSON_TRACE=4096): where the tasks ran, when they started and ended."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=5)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--edge", type=int, default=0)
ap.add_argument("--tall", type=int, default=0, help="poisson_tall_percent")
ap.add_argument("--pairs", type=int, default=1, help="poisson_pairs: tasks in pairs that walk away from a common row")
ap.add_argument("--n", type=int, default=16, help="grid: nx = 160 n - 1, ny = 256 n - 1")
a = ap.parse_args()
p = ns_params("NavierStokes_Re6", n=a.n)
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
ctx.set_option("poisson_edge_percent", a.edge)
ctx.set_option("poisson_tall_percent", a.tall)
ctx.set_option("poisson_pairs", a.pairs)
ctx.ns_poisson_iters(ptr(Pr), ptr(dP), ptr(dv), C.byref(cp), a.k * 6)
torch.cuda.synchronize()
lib = _lib.load()
fn = getattr(lib, f"nsk_debug_trace_k{a.k}")
n = 4096
buf = np.zeros((n, 4), dtype=np.uint64)
rc = fn(buf.ctypes.data_as(C.c_void_p), n)
assert rc == 0, rc
used = buf[:, 1] > 0
b = buf[used]
t0 = b[:, 0].astype(np.int64); t1 = b[:, 1].astype(np.int64)
base = t0.min()
st = (t0 - base) / 100.0; en = (t1 - base) / 100.0   # us
dur = en - st
kind = (b[:, 2] >> 32) & 0xFF; strip = (b[:, 2] >> 40) & 0xFFFF; first_row = b[:, 3] >> 32      # (filled in by the paired launch)
b[:, 2] &= 0xFFFFFFFF; b[:, 3] &= 0xFFFFFFFF
hw = b[:, 2]; xcc = b[:, 3] & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print(f"tasks {len(b)}  span {en.max():.1f} us  start: max {st.max():.1f}  dur: mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} p50 {np.median(dur):.1f} p90 {np.percentile(dur,90):.1f}")
key = (xcc.astype(np.int64) << 16) | (se.astype(np.int64) << 12) | (sh.astype(np.int64) << 8) | (cu.astype(np.int64) << 4) | simd.astype(np.int64)
u, cnt = np.unique(key, return_counts=True)
print(f"distinct SIMDs used {len(u)}; tasks per SIMD histogram {np.bincount(cnt)}")
cukey = key >> 4
u2, cnt2 = np.unique(cukey, return_counts=True)
print(f"distinct CUs used {len(u2)}; tasks per CU histogram {np.bincount(cnt2)}")
# duration by class
idx = np.nonzero(used)[0]
order = np.argsort(dur)
print("slowest 10:", [(int(idx[i]), round(float(dur[i]), 1), round(float(st[i]), 1)) for i in order[-10:]])
print("fastest 10:", [(int(idx[i]), round(float(dur[i]), 1)) for i in order[:10]])
percu = {}
for k_, e_ in zip(cukey, en):
    percu[k_] = max(percu.get(k_, 0), e_)
v = np.array(list(percu.values()))
print(f"per-CU finish: min {v.min():.1f} mean {v.mean():.1f} max {v.max():.1f}")
for x in range(8):
    m = xcc == x
    if m.any():
        print(f"xcc {x}: tasks {m.sum()} finish max {en[m].max():.1f} mean dur {dur[m].mean():.1f}")
nt = len(b)
print("edge tasks mean dur", dur[idx >= idx.max() - 200].mean() if nt > 300 else None)

# workgroups are handed out in the order of their numbers: how do the first and the second half fare?
half = (idx.max() + 1) // 2
for name, m in (("first half of the workgroup numbers", idx < half), ("second half", idx >= half)):
    if m.any():
        print(f"{name}: {m.sum()} tasks, start mean {st[m].mean():.1f}, duration mean {dur[m].mean():.1f} p10 {np.percentile(dur[m],10):.1f} "
              f"p90 {np.percentile(dur[m],90):.1f}, end mean {en[m].mean():.1f} max {en[m].max():.1f}")
# pairs on one SIMD: the one with the smaller workgroup number against the other
order_by_simd = {}
for i_, k_ in enumerate(key):
    order_by_simd.setdefault(int(k_), []).append(i_)
first, second = [], []
for v_ in order_by_simd.values():
    if len(v_) == 2:
        a_, b_ = sorted(v_, key=lambda i_: idx[i_])
        first.append(a_); second.append(b_)
if first:
    first, second = np.array(first), np.array(second)
    print(f"SIMDs with two tasks: {len(first)}; smaller number: duration mean {dur[first].mean():.1f}, end mean {en[first].mean():.1f}; "
          f"larger number: duration mean {dur[second].mean():.1f}, end mean {en[second].mean():.1f}; "
          f"pairs whose smaller number is in the first half and larger in the second: {int(((idx[first] < half) & (idx[second] >= half)).sum())}")

if kind.any():
    nstrips = int(strip.max()) + 1
    for name, m in (("plain, interior strips", (kind == 1) & (strip > 0) & (strip < nstrips - 1)), ("plain, edge strips", (kind == 1) & ((strip == 0) | (strip == nstrips - 1))),
                    ("pair, walks down", kind == 2), ("pair, walks up", kind == 3)):
        if m.any():
            print(f"{name}: {m.sum()} tasks, duration mean {dur[m].mean():.1f} p10 {np.percentile(dur[m],10):.1f} p90 {np.percentile(dur[m],90):.1f} max {dur[m].max():.1f}, "
                  f"end mean {en[m].mean():.1f} max {en[m].max():.1f}")
    slow = order[-12:]
    print("slowest 12 (kind, strip, first row, duration):", [(int(kind[i]), int(strip[i]), int(first_row[i]), round(float(dur[i]), 1)) for i in slow])
