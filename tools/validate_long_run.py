#!/usr/bin/env python3
"""Many whole time steps from the reference's initial condition, GPU against the CPU oracle, compared
as bit patterns after every step (test infrastructure; the pytest suite runs 2-3 steps per case):

    validate_long_run.py [steps_small] [steps_mid] [steps_large]

small = scripts/NavierStokes.jl as shipped (153 x 255, Re = 1e4); mid = scripts/NavierStokes_Re6.jl
with n = 2 (319 x 511, Re = 1e6), OpenMP oracle; large = the same script with n = 8 (1279 x 2047), whose
Poisson passes run as launches of K passes with the tasks in pairs (the fields do not fit on chip)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes
from oracle.ns_oracle import OracleSim
from fuzz_poisson import strict_same


def run(name, p, steps, omp):
    sim, ref = NavierStokes(p), OracleSim(p, omp=omp)
    t0, total = time.time(), 0
    for it in range(1, steps + 1):
        a, b = sim.step(), ref.step()
        total += b
        ok = a == b and sim.err_evo == ref.err_evo and sim.iter_evo == ref.iter_evo
        got = sim.host_fields()
        for k in ("Pr", "Vx", "Vy", "C"):
            ok = ok and strict_same(got[k], getattr(ref, k))
        ok = ok and strict_same(sim.dPrdtau_host(), ref.dPrdtau)
        if not ok:
            print(f"{name}: MISMATCH at step {it} ({a} vs {b} passes)", flush=True)
            return False
        if it % 10 == 0 or it == steps or steps <= 10:
            print(f"{name}: step {it:4d} ok, {total} passes so far, {time.time() - t0:.0f} s", flush=True)
    return True


if __name__ == "__main__":
    n_small = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    n_mid = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    ok = run("NavierStokes.jl 153x255", ns_params("NavierStokes"), n_small, False)
    ok = run("NavierStokes_Re6.jl n=2 319x511", ns_params("NavierStokes_Re6", n=2), n_mid, True) and ok
    n_large = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if n_large:
        n = int(os.environ.get("LARGE_N", "8"))              # 12: 1919 x 3071, paired tasks by the planner's choice
        p = ns_params("NavierStokes_Re6", n=n)
        ok = run(f"NavierStokes_Re6.jl n={n} {p.nx}x{p.ny}", p, n_large, True) and ok
    sys.exit(0 if ok else 1)
