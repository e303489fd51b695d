#!/usr/bin/env python3
"""Writes tests/golden/bubble_oracle.npz with THIS repository's CPU oracle (oracle/bubble_oracle.c):
20 steps of the code1/DNS drop (32 x 32, maxit 200) and 10 steps of code3 (32 x 32, Nf = 100).
The reference stores no outputs and cannot run here (no Julia/MATLAB), so the fixture guards the
oracle against regressions; it does not pin it to the reference."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from navierstokes_amd.bubble import Code3Params, DNSParams  # noqa: E402
from oracle.bubble_oracle import OracleCode3, OracleDNS  # noqa: E402

d = OracleDNS(DNSParams(maxit=200))
dns_sweeps = [d.step() for _ in range(20)]
c = OracleCode3(Code3Params())
c3_sweeps = [c.step() for _ in range(10)]
xf, yf = c.front()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bubble_oracle.npz"),
                    dns_sweeps=np.array(dns_sweeps), dns_rho=d.rho, dns_u=d.u, dns_v=d.v, dns_p=d.p,
                    c3_sweeps=np.array(c3_sweeps), c3_Nf=c.Nf, c3_r=c.r, c3_m=c.m, c3_u=c.u, c3_v=c.v, c3_p=c.p,
                    c3_xf=xf, c3_yf=yf)
print(dns_sweeps, c3_sweeps, c.Nf)
