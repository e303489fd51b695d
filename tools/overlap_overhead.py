#!/usr/bin/env python3
"""Compute-side cost of the two slab schedules on ONE GPU: an interior rank (two neighbours) of a
2559 x 4095-per-rank decomposition runs `passes` Poisson passes with its halo rows refilled by
device copies instead of peers (loopback), so everything but the wire is timed: redundant halo
rows, the boundary bands of the overlapped schedule and how they disturb the window's launches,
the install copies.  Prints us per pass against the single-domain loop.

  python tools/overlap_overhead.py [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.dist import HipBackend, SlabPoisson, make_slab
from navierstokes_amd.ns import _pitch


def run(p, halo, overlap, passes, world=3, rank=1):
    dev = torch.device("cuda", 0)
    slab = make_slab(rank, world, p.ny, halo)
    pg = p.with_(ny=p.ny * world, ly=p.ly * world) if world > 1 else p
    ld = _pitch(p.nx)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(7)
    def rnd():
        t = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
        t[:, :p.nx] = torch.randn((slab.ny_win, p.nx), generator=g, dtype=torch.float64) * 1e-2
        return t.to(dev)
    Pr, divV = rnd(), rnd()
    dP = torch.zeros_like(Pr)
    sp = SlabPoisson(pg, slab, HipBackend(ctx, pg, ld), Pr, dP, divV, overlap=overlap)
    sp.loopback = True
    sp.iterate(2 * max(halo, 6))
    sp.timing = True
    if os.environ.get("NSK_DEBUG_POLL"):
        streams = {"main": torch.cuda.current_stream()}
        if sp.overlap:
            streams.update(band=sp.band_stream, comm=sp.comm_stream)
        for i in range(8):
            state = {k: v.query() for k, v in streams.items()}
            print("poll", i, state, flush=True)
            if all(state.values()):
                break
            time.sleep(1.0)
        else:
            print("stuck: streams", {k: (hex(v.cuda_stream), v.priority) for k, v in streams.items()}, flush=True)
            os._exit(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sp.iterate(passes)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ex = sp.exposed_us()
    return dt / passes * 1e6, ex


def main():
    import faulthandler
    faulthandler.dump_traceback_later(45, exit=True)     # a stuck run says where and ends
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
    cases = [(60, False), (120, False), (24, True), (36, True), (60, True)]
    if len(sys.argv) > 2:
        cases = [(int(h), o == "o") for h, o in (c.split(":") for c in sys.argv[2:])]
    p = ns_params("NavierStokes_Re6", n=16)
    base, _ = run(p, 0, False, passes, world=1, rank=0)
    print(f"single domain                       {base:7.2f} us/pass", flush=True)
    for halo, overlap in cases:
        faulthandler.dump_traceback_later(45, exit=True)
        t, ex = run(p, halo, overlap, passes)
        print(f"halo {halo:3d} {'overlapped' if overlap else 'plain     '}  {t:7.2f} us/pass  "
              f"{(t / base - 1) * 100:5.1f} % over single domain, {ex:6.1f} us per exchange after the last pass", flush=True)


if __name__ == "__main__":
    main()
