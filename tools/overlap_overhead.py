#!/usr/bin/env python3
"""Compute-side cost of the slab schedules on ONE GPU: an interior rank (two neighbours) of a
2559 x 4095-per-rank decomposition runs `passes` Poisson passes with its halo rows refilled by
device copies instead of peers (loopback), so everything but the wire is timed: redundant halo
rows and the cut last launch of the edges-first schedule.  With a wire time (us) a spin
kernel of that length stands where the transfer would run, which shows what each schedule hides.
Prints us per pass against the single-domain loop.

  python tools/overlap_overhead.py [passes] [wire_us] [halo:schedule ...]    schedule p | e"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.dist import HipBackend, SlabPoisson, make_slab
from navierstokes_amd.ns import _pitch


def spin_cycles(us):
    """torch.cuda._sleep cycles for `us` microseconds (calibrated on the spot)."""
    if us <= 0:
        return 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10_000_000
    torch.cuda._sleep(n)
    torch.cuda.synchronize()
    e0.record()
    torch.cuda._sleep(n)
    e1.record()
    torch.cuda.synchronize()
    return int(n * us / (e0.elapsed_time(e1) * 1e3))


def run(p, halo, schedule, passes, world=3, rank=1, wire=0):
    dev = torch.device("cuda", 0)
    slab = make_slab(rank, world, p.ny, halo)
    pg = p.with_(ny=p.ny * world, ly=p.ly * world) if world > 1 else p
    ld = _pitch(p.nx)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("poisson_edge_task_rows", int(os.environ.get("NSK_EDGE_TASK_ROWS", "0")))
    g = torch.Generator(device="cpu").manual_seed(7)
    def rnd():
        t = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
        t[:, :p.nx] = torch.randn((slab.ny_win, p.nx), generator=g, dtype=torch.float64) * 1e-2
        return t.to(dev)
    Pr, divV = rnd(), rnd()
    dP = torch.zeros_like(Pr)
    sp = SlabPoisson(pg, slab, HipBackend(ctx, pg, ld), Pr, dP, divV, schedule=schedule)
    sp.loopback = True
    sp.loopback_wire = wire
    sp.iterate(2 * max(halo, 6))
    sp.timing = True
    if os.environ.get("NSK_DEBUG_POLL"):
        streams = {"main": torch.cuda.current_stream()}
        if sp.schedule == "edges":
            streams.update(edge=sp.edge_stream)
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
    wire_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    cases = [(60, "plain"), (60, "edges"), (120, "plain"), (120, "edges")]
    if len(sys.argv) > 3:
        names = {"p": "plain", "e": "edges"}
        cases = [(int(h), names[o]) for h, o in (c.split(":") for c in sys.argv[3:])]
    p = ns_params("NavierStokes_Re6", n=16)
    torch.cuda.set_device(0)
    wire = spin_cycles(wire_us)
    base, _ = run(p, 0, "plain", passes, world=1, rank=0)
    print(f"single domain                       {base:7.2f} us/pass   (stand-in for the transfer: {wire_us:.0f} us per exchange)", flush=True)
    for halo, schedule in cases:
        faulthandler.dump_traceback_later(45, exit=True)
        t, ex = run(p, halo, schedule, passes, wire=wire)
        print(f"halo {halo:3d} {schedule:6s}  {t:7.2f} us/pass  "
              f"{(t / base - 1) * 100:5.1f} % over single domain, {ex:6.1f} us per exchange after the last pass", flush=True)


if __name__ == "__main__":
    main()
