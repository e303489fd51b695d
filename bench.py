#!/usr/bin/env python3
"""Benchmark of the hot path: the pseudo-transient pressure-Poisson loop of
scripts/NavierStokes_Re6.jl (2559 x 4095, Float64) on N MI355X.

One bench "step" = one residual-check interval of the reference's Poisson loop
(NavierStokes_Re6.jl:84-95): nchk = 4*(nx-1) = 10232 passes of update_dPrdτ! + update_Pr! +
set_bc_Pr! followed by compute_res! + max|Rp| -- the smallest unit of Poisson work a time step
performs (a time step runs 1..12 of them).  T_eff follows the ParallelStencil convention:
algorithmic bytes = (2 + 2 + 1) fields x 8 B = 40 B per cell and pass.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

N > 1 is weak scaling: every rank owns a 2559 x 4095 slab of a 2559 x (4095 N) grid; halos
travel over RCCL (navierstokes_amd/dist.py).  Prints ONE JSON line on rank 0.

The `roofline` block: `achieved`/`frac` price the ALGORITHMIC bytes (40 B per cell and pass, K
passes per launch) against the HBM peak -- with K passes per sweep over the fields this exceeds 1
by design.  The physical fractions: `fabric_frac` prices the bytes the L2 <-> fabric counters
(FETCH_SIZE / WRITE_SIZE: requests that leave L2, Infinity-Cache hits included, so an upper bound
on the DRAM bytes) saw per launch, `valu_frac` the issue cycles of the vector ALU instructions the
launch executed, at the 2.4 GHz peak clock and (`valu_frac_sustained`) at the clock the part held
in the counter run (SQ_BUSY_CYCLES); `frac_of_bound` is the larger of the two, `bound` its name.
`launch_us` is the device time of the timed loop itself (events on the launch stream around it)
divided by its passes / K: the residual checks and a remainder launch are in it, so it bounds the
kernel's average duration from above and launch_us x launches <= ms_per_step.

N > 1: the first exchange and all-reduce run under a watchdog (--first-contact-s): if they do not
complete, every rank prints where it was (rank, device, peers, bytes, schedule) and exits non-zero.
"""
from __future__ import annotations

import argparse
import ctypes as C
import faulthandler
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak
SIMDS = 256 * 4            # 256 compute units x 4 SIMDs
CLOCK_GHZ = 2.4            # peak engine clock
VALU_CYCLES_F64 = 4        # issue cycles of one wave-wide fp64 vector instruction on a SIMD16


def poisson_counters(kernel: str, nx: int, ny_win: int):
    """HBM bytes and vector-ALU instructions per launch of `kernel`, measured with rocprofv3 PMC
    passes (tools/make_traffic.py) -- only if they were taken from this very kernel source and build
    flags on this grid; anything else would be a stale number, so None then."""
    from navierstokes_amd._lib import poisson_kernel_hash
    try:
        with open(os.path.join(ROOT, "profiles", "poisson_traffic.json")) as f:
            t = json.load(f).get(kernel)
    except (OSError, ValueError):
        return None
    if not t or t.get("kernel_hash") != poisson_kernel_hash() or t.get("grid") != [nx, ny_win]:
        return None
    return t


class Watchdog:
    """Ends the process (exit status 3, never a re-exec) when a phase does not finish in time: a host-side
    deadline for work whose waits sit inside libraries (the first RCCL exchange and all-reduce of a
    multi-GPU run).  `describe` returns what to print."""

    def __init__(self, seconds: float, describe):
        self.seconds, self.describe = seconds, describe
        self.timer = None

    def _expired(self):
        try:
            sys.stderr.write(f"bench.py: no completion within {self.seconds:.0f} s -- {self.describe()}\n")
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            sys.stderr.flush()
        finally:
            os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            self.timer = threading.Timer(self.seconds, self._expired)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def cpu_baseline(p, seconds_target=15.0):
    """The oracle's OpenMP build (same loops, outer index threaded like ParallelStencil's Threads
    backend) timed on this host: a bounded number of Poisson passes on the same 2559 x 4095 fields."""
    from oracle.ns_oracle import Oracle, farray
    # every core this process may use: the launcher's OMP_NUM_THREADS where it sets one (a GPU box
    # gives one GPU's job a share of the host, not the whole machine), else the affinity mask
    present = os.cpu_count() or 1
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else present
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or allowed
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = Oracle(omp=True)
    nx, ny = p.nx, p.ny
    rng = np.random.default_rng(1234)
    Pr = np.asfortranarray(rng.standard_normal((nx, ny)) * 1e-2)
    dv = np.asfortranarray(rng.standard_normal((nx, ny)) * 1e-2)
    dP = farray(nx - 2, ny - 2)

    def passes(n):
        t0 = time.perf_counter()
        for _ in range(n):
            o.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
            o.update_Pr(Pr, dP, p.dtau)
            o.set_bc_Pr(Pr, 0.0, p.full_rows)
        return time.perf_counter() - t0

    t1 = passes(2)                                   # warm-up + calibration
    n = max(4, min(6000, int(seconds_target / max(t1 / 2, 1e-4))))
    t = passes(n)
    t_iter = t / n
    return {"value": 40.0 * nx * ny / t_iter / 1e9, "unit": "GB/s", "cores": threads, "cores_present": present,
            "cores_allowed": allowed, "kind": "port",
            "sample": f"{n} Poisson passes at {nx}x{ny} (oracle, OpenMP over iy, {threads} threads of "
                      f"{present} host CPUs), {t_iter*1e3:.2f} ms/pass",
            "steps_per_s_at_nchk": 1.0 / (t_iter * p.nchk)}


def group_k(group) -> int:
    """Passes per launch the library chose for the slab group's windows."""
    return group.get_option("poisson_k")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", dest="n", type=int, default=16, help="runme(n=...): nx=160n-1, ny=256n-1 per GPU")
    ap.add_argument("--halo", type=int, default=60, help="halo rows = passes between exchanges (N>1)")
    ap.add_argument("--passes", type=int, default=0, help="passes per step (default nchk)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU work of the baseline leg (bounded sample)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    ap.add_argument("--one-gpu", action="store_true",
                    help="rehearsal: all ranks share cuda:0 (needs --backend gloo; not a measurement)")
    ap.add_argument("--schedule", choices=["plain", "edges"], default=None,
                    help="N>1: plain = passes, then the exchange; edges (default) = the rows to be sent come first in "
                         "the last launch of a block and travel beside the rest of it")
    ap.add_argument("--abi-slabs", action="store_true",
                    help="run the slabs through the C ABI (nsk_slabs_*: RCCL called from the library, one slab "
                         "per process) instead of the torch.distributed host path")
    ap.add_argument("--first-contact-s", type=float, default=240.0,
                    help="N>1: deadline for process-group set-up, the first exchange and the first all-reduce")
    ap.add_argument("--time-steps", type=int, default=5,
                    help="additionally run this many whole reference time steps from the IC (N=1)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from navierstokes_amd import _lib, ns_params
    from navierstokes_amd.dist import HipBackend, SlabPoisson, exchange_rows, make_slab
    from navierstokes_amd.ns import _pitch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path)")
    if a.one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    where = {"phase": "process group set-up", "exchanges": 0}

    def describe():
        peers = [r for r in (rank - 1, rank + 1) if 0 <= r < world]
        return (f"rank {rank} of {world} on cuda:{local} ({torch.cuda.get_device_name(local)}), backend {a.backend}, "
                f"peers {peers}, phase: {where['phase']}, exchanges completed: {where['exchanges']}, "
                f"{a.halo} halo rows = {a.halo * 2560 * 8} bytes per field, neighbour and direction, "
                f"schedule {'abi-slabs' if a.abi_slabs else (a.schedule or 'edges')}")

    if world > 1:
        if os.environ.get("NSK_BENCH_TEST_STALL") == str(rank):
            # (test of the watchdog: this rank never arrives -- and stays silent, like a hung peer; with a watchdog
            #  of its own it could leave first and have the launcher end the waiting rank before that one reports)
            time.sleep(10 * a.first_contact_s)
        with Watchdog(a.first_contact_s, describe):
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group(a.backend)

    p1 = ns_params("NavierStokes_Re6", n=a.n)                   # one GPU's slab
    nx, ny_loc = p1.nx, p1.ny
    slab = make_slab(rank, world, ny_loc, a.halo)
    # global problem: same scalars, ly grows with the number of slabs (dy, dt, dτ unchanged)
    p = p1.with_(ny=ny_loc * world, ly=p1.ly * world) if world > 1 else p1
    passes = a.passes or p1.nchk
    ld = _pitch(nx)
    ctx = _lib.Context(local, torch.cuda.current_stream().cuda_stream)

    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    def rnd():
        t = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
        t[:, :nx] = torch.randn((slab.ny_win, nx), generator=g, dtype=torch.float64) * 1e-2
        return t.to(dev)
    Pr, divV = rnd(), rnd()
    dP = torch.zeros((slab.ny_win, ld), dtype=torch.float64, device=dev)
    with Watchdog(a.first_contact_s if world > 1 else 0, describe):
        where["phase"] = "first halo exchange (Pr, divV)"
        exchange_rows((Pr, divV), slab)
        torch.cuda.synchronize()
        where["exchanges"] = 1
    be = HipBackend(ctx, p, ld)
    sp = SlabPoisson(p, slab, be, Pr, dP, divV, schedule=a.schedule)

    group = None
    if a.abi_slabs:
        # same windows, library-owned; the 128-byte RCCL id travels through torch.distributed once
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(_lib.rccl_unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            uid = uid.to(dev) if a.backend == "nccl" else uid
            dist.broadcast(uid, 0)
        from navierstokes_amd.ns import c_params
        group = _lib.Slabs(world, rank, [local], c_params(p), ny_loc, a.halo, transport=1,
                           unique_id=bytes(uid.cpu().numpy().tobytes()))
        w, f = group.window(0), group.fields(0)
        assert (w["ny_win"], w["ld"], w["iy0"]) == (slab.ny_win, ld, slab.iy0)
        n = slab.ny_win * ld * 8
        for name, t in (("Pr", Pr), ("dPrdtau", dP), ("divV", divV)):
            ctx.copy(C.c_void_p(f[name]), C.c_void_p(t.data_ptr()), slab.ny_win * ld)
        ctx.sync()

    def step():
        if group is not None:
            group.poisson_iters(passes)
            return group.residual()
        sp.iterate(passes)
        return sp.residual()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with Watchdog(a.first_contact_s if world > 1 else 0, describe):
        # first contact of the loop itself: one block of passes with its exchange, one all-reduced residual
        where["phase"] = "first block of Poisson passes and its exchange"
        if group is not None:
            group.poisson_iters(max(a.halo, 1))
        else:
            sp.iterate(max(a.halo, 1))
        torch.cuda.synchronize()
        where["exchanges"] += 1
        where["phase"] = "first all-reduce of the residual"
        group.residual() if group is not None else sp.residual()
        where["phase"] = "barrier after the first contact"
        fence()
    where["phase"] = "warm-up / timed loop"
    for _ in range(a.warmup):
        step()
    fence()
    sp.timing, ex0 = world > 1, sp.exchanges
    # device time of the timed loop, on the stream the library launches on (torch's current stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(a.steps):
        err = step()
    e1.record()
    fence()
    dt = time.perf_counter() - t0
    gpu_s = e0.elapsed_time(e1) * 1e-3
    exposed, n_exch = sp.exposed_us(), sp.exchanges - ex0
    sp.timing = False
    if world > 1:
        tt = torch.tensor([dt, exposed or 0.0], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, exposed = float(tt[0].item()), float(tt[1].item())

    # dominant kernel: its launches of K passes make up the timed loop but for the residual checks and a
    # remainder launch per block, so loop time / (passes / K) bounds its average duration from above
    kpl = ctx.get_option("poisson_k") if group is None else group_k(group)
    launches = a.steps * passes / kpl
    t_launch = gpu_s / launches
    alg_bytes = 40.0 * nx * slab.ny_win * kpl
    achieved = alg_bytes / t_launch / 1e9

    paired = group is None and ctx.get_option("poisson_pairs_used") == 1   # (the launch's tasks in pairs: its own kernel)
    kernel = f"k_poisson_pairs<{kpl}>" if paired else f"k_poisson<{kpl}>"
    if world > 1 and (ctx.get_option("poisson_pairs") if group is None else 1) == 1:
        # a slab block ends with the launch that is cut in three (plain tasks, the one the option above reports);
        # the launches before it run their tasks in pairs where the planner chooses that form
        kernel = f"k_poisson_pairs<{kpl}>, the last launch of a block k_poisson<{kpl}>"
    cnt = poisson_counters(kernel, nx, slab.ny_win)
    traffic = fabric_frac = valu_frac = valu_sus = clock_sus = None
    if cnt:
        traffic = cnt["read_bytes"] + cnt["write_bytes"]
        fabric_frac = traffic / t_launch / (HBM_PEAK_GBS * 1e9)
        valu_frac = cnt["insts_valu"] * VALU_CYCLES_F64 / (SIMDS * t_launch * CLOCK_GHZ * 1e9)
        if cnt.get("sq_busy_cycles"):
            # SQ_BUSY_CYCLES sums over the 32 shader engines: / 32 = cycles one launch took in the counter run
            cyc = cnt["sq_busy_cycles"] / 32.0
            valu_sus = cnt["insts_valu"] * VALU_CYCLES_F64 / (SIMDS * cyc)
            if cnt.get("launch_us"):
                clock_sus = cyc / cnt["launch_us"] * 1e-3
    fracs = {"fabric": fabric_frac or 0.0, "valu": max(valu_frac or 0.0, valu_sus or 0.0)}
    bound = max(fracs, key=fracs.get)

    cells = nx * ny_loc * world
    t_iter = dt / (a.steps * passes)
    teff = 40.0 * cells / t_iter / 1e9
    out = {
        "metric": "T_eff", "value": teff, "unit": "GB/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"NavierStokes_Re6.jl Poisson loop, {nx}x{ny_loc} cells per GPU, "
                               f"{passes} passes + 1 residual check per step",
                   "grid_per_gpu": [nx, ny_loc], "passes_per_step": passes,
                   "parallelism": f"slab{world}" if world > 1 else "single",
                   "halo_rows": slab.halo, "passes_per_launch": kpl},
        "t_iter_us": t_iter * 1e6,
        "t_eff_over_hbm_peak": teff / (HBM_PEAK_GBS * world),
        "time_steps_per_s_at_min_iters": 1.0 / (t_iter * p1.nchk),
        "last_err": err,
        "roofline": {"bound": "hbm" if bound == "fabric" else "valu",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_kind": "L2<->fabric bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE): Infinity-Cache hits "
                                     "included, an upper bound on DRAM bytes",
                     "fabric_frac": fabric_frac, "valu_frac": valu_frac, "valu_frac_sustained": valu_sus,
                     "sustained_clock_ghz": clock_sus,
                     "frac_of_bound": max(fracs.values()) if cnt else None,
                     "kernel": kernel, "launch_us": t_launch * 1e6, "launches_per_step": passes / kpl,
                     "launch_us_from": "events around the timed loop / (passes / K): residual checks and remainder "
                                       "launches included",
                     "algorithmic_bytes_per_launch": alg_bytes, "passes_per_launch": kpl,
                     "note": "achieved/frac price algorithmic bytes (K passes per sweep: > 1 by design); "
                             "fabric_frac = counter bytes / launch / 8 TB/s; valu_frac = SQ_INSTS_VALU x 4 cycles / "
                             "(1024 SIMDs x launch cycles at 2.4 GHz), valu_frac_sustained the same at the clock of the "
                             "counter run (SQ_BUSY_CYCLES / 32 per launch); null = no counters for this kernel build"},
    }
    if world > 1:
        out["schedule"] = "abi-slabs (RCCL from the library)" if group is not None else sp.schedule
        out["exchanges_timed"] = n_exch
        out["comm_exposed_us_per_exchange"] = exposed
        # who took part: one entry per rank, gathered from every rank
        props = torch.cuda.get_device_properties(local)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": f"cuda:{local}",
                "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")),
                "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"), "pid": os.getpid()}
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        out["world_size"] = dist.get_world_size()
        out["backend"] = dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")
        out["ranks"] = everyone
        out["distinct_devices"] = len({(e["uuid"], e["pci_bus_id"], e["device"]) for e in everyone})
    if rank == 0 and world == 1 and a.time_steps > 0:
        from navierstokes_amd.ns import NavierStokes
        sim = NavierStokes(p1, device=f"cuda:{local}")
        its, secs = [], []
        torch.cuda.synchronize()
        for _ in range(a.time_steps):
            t0 = time.perf_counter()
            its.append(sim.step())
            torch.cuda.synchronize()
            secs.append(time.perf_counter() - t0)
        tts = sum(secs)
        out["time_steps"] = {"steps": a.time_steps, "steps_per_s": a.time_steps / tts,
                             "poisson_iters": its, "seconds": secs,
                             "us_per_pass_whole_step": [s_ / i * 1e6 for s_, i in zip(secs, its)],
                             "from": "reference initial condition"}
        # the reference's own CPU-runnable configuration (scripts/NavierStokes.jl as shipped, 153 x 255): its
        # fields fit on chip, so the passes between two residual checks run as one launch; the same steps with
        # a launch per K passes beside it
        ps = ns_params("NavierStokes")
        small = {"grid": [ps.nx, ps.ny], "from": "reference initial condition", "steps": 4}
        for key, one_launch in (("one_launch_per_run_of_passes", 1), ("launch_per_k_passes", 0)):
            sim = NavierStokes(ps, device=f"cuda:{local}")
            sim.ctx.set_option("poisson_small", one_launch)
            sim.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            its = [sim.step() for _ in range(small["steps"])]
            torch.cuda.synchronize()
            ts = time.perf_counter() - t0
            small[key] = {"steps_per_s": small["steps"] / ts, "us_per_pass_whole_step": ts / sum(its) * 1e6, "poisson_iters": its}
        out["time_steps_default_grid"] = small
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(p1, a.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
