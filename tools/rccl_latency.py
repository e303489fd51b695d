#!/usr/bin/env python3
"""What one halo exchange costs in RCCL software alone, measured on ONE GPU: a one-rank communicator
sends row blocks of the size of a 60-row halo (1.2 MB) to itself through ncclSend / ncclRecv in one
group on the communication stream, ordered against the kernel stream with events exactly as a real
exchange (nsk_slabs_rccl_self_exchange).  No xGMI is involved, so this is the launch and protocol
overhead a real exchange pays before any wire time.

  python tools/rccl_latency.py [rows] [repeats]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import c_params

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 60
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
torch.cuda.set_device(0)
p = ns_params("NavierStokes_Re6", n=16)
g = _lib.Slabs(1, 0, [0], c_params(p), p.ny, 0, transport=1, unique_id=_lib.rccl_unique_id())
w = g.window(0)
for r in (rows, 1):
    for _ in range(10):
        g.call("nsk_slabs_rccl_self_exchange", r)
    g.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.call("nsk_slabs_rccl_self_exchange", r)
        g.sync()                      # each exchange waited for: latency, not throughput
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        g.call("nsk_slabs_rccl_self_exchange", r)
    g.sync()
    dq = (time.perf_counter() - t0) / reps
    mb = r * w["ld"] * 8 / 1e6
    print(f"self send+recv of {r} rows ({mb:.2f} MB): {dt*1e6:7.1f} us each when waited for, {dq*1e6:7.1f} us each back to back")
g.close()
