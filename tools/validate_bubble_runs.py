#!/usr/bin/env python3
"""The two bubble drivers over their whole reference runs, GPU against the CPU oracle, compared as bit
patterns (test infrastructure; the pytest suite runs a dozen steps per case):

    validate_bubble_runs.py [dns_steps] [code3_steps]

DNS.jl (form of code1.m): 32 x 32, the reference runs 100 steps; code3_fun.jl: 32 x 32 with a front
of 100 points, the reference runs 300 steps.  The GPU advances in runs of 25 steps per C call
(nsk_bub_dns_run / nsk_bub_code3_run); after every run the sweep counts of all its steps, the
fields and the front are compared."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from navierstokes_amd.bubble import Code3, Code3Params, DNS, DNSParams
from oracle.bubble_oracle import FrontLeftTheBox, OracleCode3, OracleDNS
from fuzz_poisson import strict_same

CHUNK = 25


def dns(steps):
    q = DNSParams()
    g, o = DNS(q), OracleDNS(q)
    t0, done, sweeps = time.time(), 0, 0
    while done < steps:
        n = min(CHUNK, steps - done)
        ref = [o.step() for _ in range(n)]
        ok = g.run(n) == ref
        for k in ("rho", "u", "v", "p", "uu", "vv"):
            ok = ok and strict_same(getattr(g, k).get(), getattr(o, k))
        done += n
        sweeps += sum(ref)
        if not ok:
            print(f"DNS.jl: MISMATCH in steps {done - n + 1}..{done}", flush=True)
            return False
        print(f"DNS.jl 32x32: step {done:4d} ok, {sweeps} sweeps so far, {time.time() - t0:.1f} s", flush=True)
    return True


def code3(steps):
    q = Code3Params()
    g, o = Code3(q), OracleCode3(q)
    t0, done, sweeps = time.time(), 0, 0
    while done < steps:
        n = min(CHUNK, steps - done)
        try:
            ref = [list(o.step()) for _ in range(n)]
        except FrontLeftTheBox:
            print(f"code3: the front leaves the box after step {done}; the reference's run ends there too", flush=True)
            return True
        ok = g.run(n) == ref and g.Nf == o.Nf
        for k in ("r", "m", "u", "v", "p"):
            ok = ok and strict_same(getattr(g, k).get(), getattr(o, k))
        (gx, gy), (ox, oy) = g.front(), o.front()
        ok = ok and strict_same(gx, ox) and strict_same(gy, oy)
        done += n
        sweeps += sum(sum(r) for r in ref)
        if not ok:
            print(f"code3: MISMATCH in steps {done - n + 1}..{done}", flush=True)
            return False
        print(f"code3 32x32: step {done:4d} ok, front of {g.Nf} points, {sweeps} sweeps so far, {time.time() - t0:.1f} s", flush=True)
    return True


if __name__ == "__main__":
    a = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    sys.exit(0 if dns(a) and code3(b) else 1)
