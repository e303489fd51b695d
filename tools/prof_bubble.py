#!/usr/bin/env python3
"""Run N steps of one bubble driver (for rocprofv3 --kernel-trace --stats):
    prof_bubble.py dns|code3 [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd.bubble import Code3, Code3Params, DNS, DNSParams

which = sys.argv[1] if len(sys.argv) > 1 else "dns"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sim = DNS(DNSParams()) if which == "dns" else Code3(Code3Params())
for _ in range(steps):
    sim.step()
torch.cuda.synchronize()
