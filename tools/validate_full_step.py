#!/usr/bin/env python3
"""One whole reference time step at the benchmark size (2559 x 4095, NavierStokes_Re6.jl, from its
initial condition: tens of thousands of Poisson passes until the residual test passes) on the GPU
and with the OpenMP oracle on the host cores; compares pass count, residual history and fields bit
for bit.  Takes minutes (the oracle), so it is a one-off check, not part of the test suite."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes
from oracle.ns_oracle import OracleSim

p = ns_params("NavierStokes_Re6", n=int(sys.argv[1]) if len(sys.argv) > 1 else 16)
sim = NavierStokes(p)
t0 = time.perf_counter()
it_gpu = sim.step()
torch.cuda.synchronize()
print(f"GPU: {it_gpu} passes in {time.perf_counter()-t0:.2f} s, history {list(zip(sim.iter_evo, sim.err_evo))}", flush=True)
ref = OracleSim(p, omp=True)
res = {}
th = threading.Thread(target=lambda: res.setdefault("it", ref.step()))
t0 = time.perf_counter()
th.start()
while th.is_alive():
    th.join(30.0)
    print(f"  oracle running, {time.perf_counter()-t0:.0f} s", flush=True)
print(f"oracle: {res['it']} passes in {time.perf_counter()-t0:.1f} s on {os.environ.get('OMP_NUM_THREADS', '?')} threads", flush=True)
ok = res["it"] == it_gpu and sim.iter_evo == ref.iter_evo and sim.err_evo == ref.err_evo
got = sim.host_fields()
for k in ("Pr", "Vx", "Vy", "C"):
    same = np.array_equal(got[k].view(np.uint64), getattr(ref, k).view(np.uint64))
    print(f"{k}: {'bit-identical' if same else 'DIFFERENT'}")
    ok &= same
same = np.array_equal(sim.dPrdtau_host().view(np.uint64), ref.dPrdtau.view(np.uint64))
print(f"dPrdtau: {'bit-identical' if same else 'DIFFERENT'}")
print("RESULT", "OK" if ok and same else "MISMATCH")
sys.exit(0 if ok and same else 1)
