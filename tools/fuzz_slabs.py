#!/usr/bin/env python3
"""Randomised parity sweep of the slab groups behind the C ABI (nsk_slabs_*; all slabs of a group on
cuda:0, halo rows by device copies) against the single-domain CPU oracle (test infrastructure):

    fuzz_slabs.py [--cases N] [--seed S]

Poisson cases draw the number of slabs, their height, the halo width (1 .. the slab height), the
grid width, the passes per launch and a pass count that is not a multiple of the halo width; time
step cases draw slabs, halo >= 4, scalars and a moving initial field, and run two whole steps.
Gathered global arrays are compared with the oracle's as bit patterns (NaN payloads aside)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import c_params, init_vprof
from oracle.ns_oracle import Oracle, OracleSim
from fuzz_poisson import strict_same


def sweep(oracle, cases, seed, verbose=True):
    rng = np.random.default_rng(seed)
    failures, t_start, refused = [], time.time(), 0
    for case in range(cases):
        world = int(rng.integers(2, 7))
        step_case = rng.random() < 0.35
        ny_loc = int(rng.integers(6 if step_case else 3, 48))
        halo = int(rng.integers(4 if step_case else 1, ny_loc + 1))
        nx = int(rng.choice([3, 5, 33, 70, 116, 117, 130, 233, 260])) if rng.random() < 0.5 else int(rng.integers(3, 300))
        nyg = world * ny_loc
        if step_case:   # dτ follows dy alone (NavierStokes.jl:54): with dx well below dy the loop diverges
            nx = int(rng.integers(3, max(4, int(0.6 * nyg)) + 1))
        full = bool(rng.integers(0, 2))
        ok, why = True, ""
        if not step_case:
            count = int(rng.integers(1, 3 * halo + 8))
            p = ns_params("NavierStokes", nx=nx, ny=nyg, full_rows=full)
            Pr0 = np.asfortranarray(rng.standard_normal((nx, nyg)))
            dP0 = np.asfortranarray(rng.standard_normal((nx - 2, nyg - 2)))
            dv = np.asfortranarray(rng.standard_normal((nx, nyg)) * 10.0)
            g = _lib.Slabs(world, 0, [0] * world, c_params(p), ny_loc, halo, transport=2)
            try:
                g.set_option("edges_first", int(rng.integers(0, 2)))
                for name, arr in (("Pr", Pr0), ("dPrdtau", dP0), ("divV", dv)):
                    g.upload(name, arr)
                g.poisson_iters(count)
                res = g.residual()
                gPr, gdP = np.full((nx, nyg), np.nan, order="F"), np.full((nx - 2, nyg - 2), np.nan, order="F")
                g.download("Pr", gPr)
                g.download("dPrdtau", gdP)
                exch = g.info()["exchanges"]
            finally:
                g.close()
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            for _ in range(count):
                oracle.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
                oracle.update_Pr(Pr, dP, p.dtau)
                oracle.set_bc_Pr(Pr, 0.0, full)
            Rp = np.zeros((nx - 2, nyg - 2), order="F")
            oracle.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
            if not (strict_same(gPr, Pr) and strict_same(gdP, dP)):
                ok, why = False, "fields"
            elif res != oracle.maxabs(Rp):
                ok, why = False, "residual"
            elif exch != -(-count // halo):
                ok, why = False, f"{exch} exchanges"
            desc = f"case {case}: poisson world={world} ny_loc={ny_loc} halo={halo} nx={nx} full={full} count={count}"
        else:
            nchk = int(rng.integers(2, 12))
            p = ns_params("NavierStokes", nx=nx, ny=nyg, full_rows=full, niter=nchk * int(rng.integers(1, 4)), nchk=nchk,
                          oy=float(rng.uniform(-0.4, 0.4)), mu=float(10.0 ** rng.uniform(-5, -2)))
            ref = OracleSim(p)
            # at most (halo - 3) / 1 rows of back-trace: |Vy| dt / dy + 2 <= halo - 1
            vmax = min(2.0, (halo - 3.5)) * p.dy / p.dt
            # smooth fields plus a little noise: a rough field has a divergence that the short Poisson
            # loop turns into velocities beyond any halo
            def smooth(n1, n2):
                x, y = np.meshgrid(np.linspace(0, 1, n1), np.linspace(0, 1, n2), indexing="ij")
                kx, ky, ph = rng.integers(1, 4), rng.integers(1, 4), rng.uniform(0, 6.28)
                return np.sin(kx * 3.1 * x + ph) * np.cos(ky * 2.9 * y) + 0.02 * rng.standard_normal((n1, n2))
            Vx = np.asfortranarray(smooth(nx + 1, nyg) * vmax)
            Vy = np.asfortranarray(smooth(nx, nyg + 1) * vmax * 0.45)
            Cc = np.asfortranarray(rng.random((nx, nyg)))
            g = _lib.Slabs(world, 0, [0] * world, c_params(p), ny_loc, halo, transport=2)
            try:
                g.init_flow(init_vprof(p))
                for name, arr in (("Vx", Vx), ("Vy", Vy), ("C", Cc)):
                    g.upload(name, arr)
                    getattr(ref, name)[...] = arr
                try:
                    for _ in range(2):
                        it = ref.step()
                        if g.step() != (it, ref.iter_evo, ref.err_evo):
                            ok, why = False, "iteration count / error history"
                    shapes = {"Pr": (nx, nyg), "Vx": (nx + 1, nyg), "Vy": (nx, nyg + 1), "C": (nx, nyg), "dPrdtau": (nx - 2, nyg - 2)}
                    for k, shp in shapes.items():
                        out = np.full(shp, np.nan, order="F")
                        g.download(k, out)
                        if ok and not strict_same(out, getattr(ref, k)):
                            ok, why = False, k
                except _lib.NskError as e:
                    if "raise the halo width" not in str(e) and "with a wider halo" not in str(e):   # (on entry / after the corrector)
                        raise
                    why = "(reach check refused the step: " + str(e) + ")"   # legitimate: the field accelerated past the halo
            finally:
                g.close()
            desc = f"case {case}: step world={world} ny_loc={ny_loc} halo={halo} nx={nx} full={full} niter={p.niter} nchk={nchk} {why}"
        if not ok:
            failures.append(why + ": " + desc)
            print("MISMATCH", why, desc, flush=True)
        elif verbose and (case % 20 == 0 or "refused" in why):
            print("ok", desc, f"[{time.time() - t_start:.0f} s]", flush=True)
        refused += "refused" in why
    if verbose:
        print(f"{cases} cases, {len(failures)} mismatches, {refused} steps refused by the reach check, {time.time() - t_start:.0f} s")
    return failures


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    sys.exit(1 if sweep(Oracle(), a.cases, a.seed) else 0)
