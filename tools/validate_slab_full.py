#!/usr/bin/env python3
"""Slab-decomposed whole time steps at the benchmark size against the single-domain run on the
same GPU: three ranks (3 x 1365 rows of the 2559 x 4095 grid) share cuda:0 under gloo, halo rows
staged through host memory.  Compares pass counts, residual history and the owned rows of every
field bit for bit.  A one-off check of the multi-rank logic at full size (RCCL itself needs one GPU
per rank)."""
import os
import socket
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from navierstokes_amd import ns_params

WORLD, HALO, STEPS = 3, 60, int(sys.argv[1]) if len(sys.argv) > 1 else 1


def worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from navierstokes_amd import _lib
        from navierstokes_amd.dist import HipBackend, SlabNavierStokes, make_slab
        from navierstokes_amd.ns import _pitch, init_vprof
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        p = ns_params("NavierStokes_Re6", n=16)
        assert p.ny % WORLD == 0
        slab = make_slab(rank, WORLD, p.ny // WORLD, HALO)
        ld = _pitch(p.nx)
        ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
        sim = SlabNavierStokes(p, slab, HipBackend(ctx, p, ld), dev, ld, init_vprof(p))
        t0 = time.perf_counter()
        iters = []
        for _ in range(STEPS):
            iters.append(sim.step())
            if rank == 0:
                print(f"  slab step done: {iters[-1]} passes, {time.perf_counter()-t0:.1f} s", flush=True)
        torch.cuda.synchronize()
        res = {k: sim.owned(k)[:, : {"Vx": p.nx + 1}.get(k, p.nx)].cpu() for k in ("Pr", "Vx", "Vy", "C", "dP")}
        res["iters"], res["hist"] = iters, sim.hist
        torch.save(res, os.path.join(out, f"s{rank}.pt"))
        ctx.close()
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tempfile.mkdtemp()
    mp.spawn(worker, args=(port, out), nprocs=WORLD, join=True)
    from navierstokes_amd.ns import NavierStokes
    p = ns_params("NavierStokes_Re6", n=16)
    sim = NavierStokes(p)
    iters = [sim.step() for _ in range(STEPS)]
    parts = [torch.load(os.path.join(out, f"s{r}.pt")) for r in range(WORLD)]
    ok = all(q["iters"] == iters for q in parts) and all([h[1] for h in q["hist"]] == sim.err_evo for q in parts)
    print(f"single domain: passes {iters}; slabs: {parts[0]['iters']}; histories equal: {ok}")
    ref = sim.host_fields()
    for k in ("Pr", "Vx", "Vy", "C"):
        got = torch.cat([q[k] for q in parts]).numpy().T
        same = np.array_equal(got.view(np.uint64), np.ascontiguousarray(ref[k]).view(np.uint64)) or np.array_equal(got, ref[k])
        print(f"{k}: {'identical' if same else 'DIFFERENT'}")
        ok &= same
    got = torch.cat([q["dP"] for q in parts]).numpy().T
    same = np.array_equal(got[1:-1, 1:-1], sim.dPrdtau_host())
    print(f"dPrdtau: {'identical' if same else 'DIFFERENT'}")
    print("RESULT", "OK" if ok and same else "MISMATCH")
    sys.exit(0 if ok and same else 1)
