#!/usr/bin/env python3
"""The torch.distributed/RCCL branch of every slab schedule on ONE GPU.

A process group of one rank on the `nccl` backend (= RCCL) is as far as one GPU goes: the interior
rank of a three-slab decomposition runs its Poisson passes with every P2P operation addressed to
itself, so each send pairs with the receive queued after it -- the low halo gets the rank's own
first owned rows, the high halo its last ones.  That is what the loopback mode of SlabPoisson does
with device copies, so the two runs must agree bit for bit; what is exercised is the product code's
RCCL branch as it runs on a node: init_process_group with a device id, batched isend/irecv of row
views issued under the edge stream, the main stream (not the host) waiting on the
requests, the MAX all-reduce of the residual's bit pattern on the device, the barrier.  Also prints
the time per exchange the main stream could not hide.

  python tools/rccl_self_rehearsal.py [nx ny_loc halo passes]      exit status 0 = all schedules agree"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from navierstokes_amd import _lib, ns_params
from navierstokes_amd import dist as nsd
from navierstokes_amd.dist import HipBackend, Slab, SlabPoisson, allreduce_maxabs
from navierstokes_amd.ns import _pitch


def fields(slab, nx, ld, dev):
    g = torch.Generator(device="cpu").manual_seed(11)
    def rnd():
        t = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
        t[:, :nx] = torch.randn((slab.ny_win, nx), generator=g, dtype=torch.float64) * 1e-2
        return t.to(dev)
    Pr, divV = rnd(), rnd()
    return Pr, torch.zeros_like(Pr), divV


def main():
    nx, ny_loc, halo, passes = (int(v) for v in (sys.argv[1:5] or (319, 200, 12, 60)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    real_op = dist.P2POp
    nsd.dist.P2POp = lambda op, tensor, peer: real_op(op, tensor, 0)       # every peer is this rank
    slab = Slab(rank=1, world=3, ny_loc=ny_loc, halo=halo, nyg=3 * ny_loc)
    p = ns_params("NavierStokes", nx=nx, ny=3 * ny_loc)
    ld = _pitch(nx)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    bad = 0
    for schedule in ("plain", "edges"):
        out = {}
        for mode in ("loopback", "rccl"):
            Pr, dP, divV = fields(slab, nx, ld, dev)
            sp = SlabPoisson(p, slab, HipBackend(ctx, p, ld), Pr, dP, divV, schedule=schedule)
            sp.loopback = mode == "loopback"
            sp.iterate(halo)                                   # warm-up block (communicator set-up)
            sp.timing = True
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sp.iterate(passes)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            exposed = sp.exposed_us()
            err = sp.residual() if mode == "rccl" else allreduce_maxabs(sp.be.residual_max(Pr, divV, slab), dev)
            out[mode] = (Pr.cpu(), dP.cpu(), err)
            print(f"{schedule:6s} {mode:8s} {passes} passes {dt*1e3:8.2f} ms, exposed {exposed:7.1f} us per exchange, "
                  f"max|Rp| {err!r}", flush=True)
        same = all(torch.equal(a.view(torch.int64), b.view(torch.int64))
                   for a, b in zip(out["loopback"][:2], out["rccl"][:2])) and out["loopback"][2] == out["rccl"][2]
        print(f"{schedule:6s} RCCL-to-self == loopback: {same}", flush=True)
        bad += not same
    dist.barrier()
    nan = allreduce_maxabs(float("nan"), dev)
    print("all-reduce keeps NaN:", nan != nan, flush=True)
    bad += not (nan != nan)
    ctx.close()
    dist.destroy_process_group()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
