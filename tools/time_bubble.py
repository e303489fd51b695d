#!/usr/bin/env python3
"""Time steps per second of the two bubble drivers at the reference sizes (32 x 32 cells), GPU
against the CPU oracle.  At this size the GPU path is launch- and latency-bound (one workgroup
does the SOR solves); the scalable solvers are timed by tools/time_dns.py."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd.bubble import Code3, Code3Params, DNS, DNSParams
from oracle.bubble_oracle import OracleCode3, OracleDNS


def rate(sim, n, sync):
    sim.step()
    sync()
    t0 = time.perf_counter()
    for _ in range(n):
        sim.step()
    sync()
    return n / (time.perf_counter() - t0)


def run_rate(sim, n):
    """the same steps as one run (one C call, one wait)"""
    sim.run(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.run(n)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


gs = torch.cuda.synchronize
print(f"DNS.jl   (32x32, SOR to 1e-3): GPU {rate(DNS(DNSParams()), 100, gs):7.1f} steps/s, oracle {rate(OracleDNS(DNSParams()), 100, lambda: None):7.1f} steps/s")
print(f"code3    (32x32, Nf=100)     : GPU {rate(Code3(Code3Params()), 50, gs):7.1f} steps/s, oracle {rate(OracleCode3(Code3Params()), 50, lambda: None):7.1f} steps/s")
print(f"as runs of 100 / 50 steps in one call: DNS {run_rate(DNS(DNSParams()), 100):7.1f} steps/s, code3 {run_rate(Code3(Code3Params()), 50):7.1f} steps/s")
for n in (256, 512):
    q = DNSParams(nx=n, ny=n, dt=0.00125 * (32 / n) ** 2, maxit=200, pressure_solver="redblack")
    print(f"DNS {n}x{n} red/black: GPU {rate(DNS(q), 10, gs):7.1f} steps/s, oracle {rate(OracleDNS(q), 3, lambda: None):7.2f} steps/s")
