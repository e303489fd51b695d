#!/usr/bin/env python3
"""Runs code3 time steps (one launch per step) with the library named by NSK_LIB and prints, per step,
the four sweep counts next to the oracle's, then whether the fields agree.  Used with debug builds of
bubble_kernels.hip (NSK_SOR_DEBUG prints every judged sweep, NSK_SOR_NOINLINE makes the solver a call)
to compare the inlined and the called form of sor_pipe; see tools/debug_sor_call.sh."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from navierstokes_amd.bubble import Code3, Code3Params
from oracle.bubble_oracle import OracleCode3

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
q = Code3Params()
g, o = Code3(q), OracleCode3(q)
bad = 0
for k in range(steps):
    sg, so = g.step(), o.step()
    print(f"STEP {k+1} gpu {sg} oracle {so} Nf {g.Nf} {o.Nf}", flush=True)
    same = all(np.array_equal(getattr(g, n).get().view(np.uint64), getattr(o, n).view(np.uint64)) for n in ("r", "m", "u", "v", "p"))
    print(f"STEP {k+1} fields_equal {same}", flush=True)
    bad += (sg != so) or not same
sys.exit(1 if bad else 0)
