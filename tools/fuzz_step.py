#!/usr/bin/env python3
"""Randomised parity sweep of whole time steps (scripts/NavierStokes.jl:77-100) through nsk_ns_step
against the CPU oracle (test infrastructure):

    fuzz_step.py [--cases N] [--seed S] [--max-n 260]

Each case draws a grid, the bc_x! row rule, physical scalars (viscosity, density, gravity, time
step scale, ellipse position / axes / rotation -- including an ellipse outside the box and one that
covers it), random initial fields, a short Poisson loop (niter, nchk) and whether the three fused
kernels or the one-launch-per-call-site sequence runs; after two steps every array the reference
keeps is compared with the oracle as bit patterns (signed zeros included, NaN payloads not)."""
import argparse
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes
from oracle.ns_oracle import OracleSim
from fuzz_poisson import strict_same


def sweep(cases, seed, max_n, verbose=True):
    rng = np.random.default_rng(seed)
    interesting = [3, 4, 5, 7, 8, 9, 63, 64, 65, 66, 127, 128, 129, 130]
    failures, t_start = [], time.time()
    for case in range(cases):
        def dim():
            return int(rng.choice(interesting)) if rng.random() < 0.4 else int(rng.integers(3, max_n + 1))
        nx, ny = dim(), dim()
        full = bool(rng.integers(0, 2))
        fused = int(rng.integers(0, 2))
        nchk = int(rng.integers(1, 8))
        niter = int(rng.integers(1, 4)) * nchk + int(rng.integers(0, 3))
        beta = float(rng.uniform(-math.pi, math.pi))
        a_ly, b_ly = float(rng.uniform(0.01, 0.3)), float(rng.uniform(0.01, 0.3))
        kw = dict(niter=niter, nchk=nchk, full_rows=full, mu=float(10.0 ** rng.uniform(-6, -1)), rho=float(rng.uniform(0.5, 2.0)),
                  g=float(rng.choice([0.0, 0.3, -1.7])), a2=a_ly * a_ly, b2=b_ly * b_ly, ox=float(rng.uniform(-0.4, 0.4)),
                  oy=float(rng.uniform(-0.7, 0.7)), sinb=math.sin(beta), cosb=math.cos(beta))
        if rng.random() < 0.1:
            kw.update(a2=4.0, b2=4.0)      # the ellipse covers the whole box
        p = ns_params("NavierStokes", nx=nx, ny=ny)
        p = p.with_(dt=p.dt * float(rng.uniform(0.2, 3.0)), **kw)
        sim, ref = NavierStokes(p), OracleSim(p)
        sim.ctx.set_option("step_fused", fused)
        # 5: back-traces of several cells, clamped at the walls; tiny and huge scales as well
        vscale = float(rng.choice([0.0, 0.1, 1.0, 5.0, 1e-268, 1e-290, 1e-305, 1e280]))
        for k, shape in (("Vx", (nx + 1, ny)), ("Vy", (nx, ny + 1)), ("C", (nx, ny)), ("Pr", (nx, ny))):
            arr = np.asfortranarray(rng.standard_normal(shape) * (vscale if k != "Pr" else (1e-3 * vscale if 0.0 < vscale < 0.1 else 1e-3)))
            getattr(sim, k).set(arr)
            getattr(ref, k)[...] = arr
        ok, why = True, ""
        with np.errstate(all="ignore"):
            for _ in range(2):
                same_iters = sim.step() == ref.step() and sim.iter_evo == ref.iter_evo
                # residuals as bit patterns: a NaN residual (which ends the loop) must compare equal
                if not same_iters or not strict_same(np.array(sim.err_evo), np.array(ref.err_evo)):
                    ok, why = False, "iteration count / error history"
        got = sim.host_fields()
        for k in ("Pr", "Vx", "Vy", "C"):
            if ok and not strict_same(got[k], getattr(ref, k)):
                ok, why = False, k
        for k in ("Vx_o", "Vy_o", "C_o", "divV"):
            if ok and not strict_same(getattr(sim, k).get(), getattr(ref, k)):
                ok, why = False, k
        if ok and not strict_same(sim.dPrdtau_host(), ref.dPrdtau):
            ok, why = False, "dPrdtau"
        desc = f"case {case}: nx={nx} ny={ny} full={full} fused={fused} niter={niter} nchk={nchk} vscale={vscale} {kw}"
        if not ok:
            failures.append(why + ": " + desc)
            print("MISMATCH", why, desc, flush=True)
        elif verbose and case % 20 == 0:
            print("ok", desc[:160], f"[{time.time() - t_start:.0f} s]", flush=True)
        sim.close() if hasattr(sim, "close") else None
    if verbose:
        print(f"{cases} cases, {len(failures)} mismatches, {time.time() - t_start:.0f} s")
    return failures


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=260)
    a = ap.parse_args()
    sys.exit(1 if sweep(a.cases, a.seed, a.max_n) else 0)
