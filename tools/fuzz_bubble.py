#!/usr/bin/env python3
"""Randomised parity sweep of the two bubble drivers (scripts/DNS.jl in the form of code1.m, and
scripts/code3_fun.jl) on the GPU against the CPU oracle (test infrastructure):

    fuzz_bubble.py [--cases N] [--seed S]

Each case draws a grid (sizes on both sides of what the several-sweeps-in-flight SOR kernel takes,
square and not), densities, viscosities, gravity, wall velocities, drop position and radius, the
SOR relaxation factor, tolerance and sweep limit, the pressure solver (DNS) or the number of front
points and the surface tension (code3), and runs a few steps; sweep counts, fields and the front
are compared with the oracle as bit patterns (NaN payloads aside)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

from navierstokes_amd import _lib
from navierstokes_amd.bubble import Code3, Code3Params, DNS, DNSParams
from oracle.bubble_oracle import FrontLeftTheBox, OracleCode3, OracleDNS
from fuzz_poisson import strict_same


def sweep(cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    failures, t_start = [], time.time()
    for case in range(cases):
        nx = int(rng.choice([8, 16, 24, 31, 32, 33, 40, 48, 64, 66, 70, 96]))
        ny = nx if rng.random() < 0.5 else int(rng.choice([8, 16, 24, 31, 32, 33, 40, 48, 64, 66, 70, 96]))
        common = dict(nx=nx, ny=ny, gy=float(rng.choice([-100.0, -30.0, 0.0])), gx=float(rng.choice([0.0, 0.0, 5.0])),
                      rho2=float(rng.uniform(1.2, 4.0)), unorth=float(rng.choice([0.0, 0.0, 1.0])),
                      usouth=float(rng.choice([0.0, 0.0, -0.5])), veast=float(rng.choice([0.0, 0.0, 0.3])),
                      vwest=float(rng.choice([0.0, 0.0, -0.3])), rad=float(rng.uniform(0.08, 0.2)),
                      maxit=int(rng.choice([1, 2, 5, 40, 200, 400])), beta=float(rng.uniform(1.0, 1.5)))
        steps = int(rng.integers(2, 7))
        ok, why = True, ""
        if rng.random() < 0.5:
            q = DNSParams(xcenter=float(rng.uniform(0.35, 0.65)), ycenter=float(rng.uniform(0.4, 0.75)),
                          m0=float(rng.choice([0.01, 0.05])), maxerr=float(rng.choice([1e-3, 1e-5, 1e-1])),
                          dt=0.00125 * (32 / max(nx, ny)) ** 2,
                          pressure_solver=str(rng.choice(["sor", "sor", "sor", "redblack", "pt"])), **common)
            g, o = DNS(q), OracleDNS(q)
            g.fused = bool(rng.integers(0, 2))
            g.ctx.set_option("bub_merge_launches", int(rng.integers(0, 2)))
            if g.fused and rng.random() < 0.5:     # the same steps as one run (one C call)
                if g.run(steps) != [o.step() for _ in range(steps)]:
                    ok, why = False, "sweep counts of a run"
            else:
                for _ in range(steps):
                    if g.step() != o.step():
                        ok, why = False, "sweep count"
            for k in ("rho", "u", "v", "p", "uu", "vv"):
                if ok and not strict_same(getattr(g, k).get(), getattr(o, k)):
                    ok, why = False, k
            desc = f"case {case}: DNS {nx}x{ny} steps={steps} solver={q.pressure_solver} maxit={q.maxit} tol={q.maxerr}"
        else:
            Nf = int(rng.choice([20, 57, 100, 180, 300]))
            q = Code3Params(xc=float(rng.uniform(0.35, 0.65)), yc=float(rng.uniform(0.4, 0.75)), Nf=Nf,
                            sigma=float(rng.choice([0.0, 1.0, 10.0])), m2=float(rng.choice([0.01, 0.05])),
                            maxError=float(rng.choice([1e-3, 1e-5, 1e-1])), dt0=0.00125 * (32 / max(nx, ny)) ** 2, **common)
            g, o = Code3(q), OracleCode3(q)
            g.fused = bool(rng.integers(0, 2))
            g.ctx.set_option("bub_merge_launches", int(rng.integers(0, 2)))
            left, as_run, sos = False, g.fused and rng.random() < 0.5, []
            for _ in range(steps):
                try:
                    so = o.step()
                except FrontLeftTheBox:      # the run ends there in the reference too
                    left = True
                    break
                if as_run:
                    sos.append(list(so))
                elif g.step() != so or g.Nf != o.Nf:
                    ok, why = False, "sweep counts / front size"
                    break
            if as_run and not left:          # the same steps as one run (one C call)
                try:
                    if g.run(steps) != sos or g.Nf != o.Nf:
                        ok, why = False, "sweep counts / front size of a run"
                except _lib.NskError as e:
                    ok, why = False, f"run failed: {e}"
            if left:
                if verbose:
                    print(f"   case {case}: code3 {nx}x{ny} maxit={q.maxit}: the front left the box, case dropped", flush=True)
                continue
            for k in ("r", "m", "u", "v", "p"):
                if ok and not strict_same(getattr(g, k).get(), getattr(o, k)):
                    ok, why = False, k
            if ok:
                (gx_, gy_), (ox_, oy_) = g.front(), o.front()
                if not (strict_same(gx_, ox_) and strict_same(gy_, oy_)):
                    ok, why = False, "front"
            desc = f"case {case}: code3 {nx}x{ny} Nf={Nf}->{o.Nf} steps={steps} maxit={q.maxit} tol={q.maxError} sigma={q.sigma}"
        if not ok:
            failures.append(why + ": " + desc)
            print("MISMATCH", why, desc, flush=True)
        elif verbose and case % 10 == 0:
            print("ok", desc, f"[{time.time() - t_start:.0f} s]", flush=True)
    if verbose:
        print(f"{cases} cases, {len(failures)} mismatches, {time.time() - t_start:.0f} s")
    return failures


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    sys.exit(1 if sweep(a.cases, a.seed) else 0)
