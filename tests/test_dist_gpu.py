"""The slab-decomposed time step with the HIP backend.  Only one GPU is available to the tests,
so two ranks share it: they run as two processes on cuda:0 under gloo (halo rows staged through
host memory).  This exercises what the CPU tests cannot: the window arguments (iy0, nyg) of the
real kernels -- mask coordinates, boundary conditions on physical edges only, the global clamp
of the back-trace, the multi-pass Poisson kernel on windows with halo edges -- against the
single-domain oracle, bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from navierstokes_amd import ns_params

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, halo, schedule, nx, ny_loc, steps, niter, nchk, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from navierstokes_amd import _lib
        from navierstokes_amd.dist import HipBackend, SlabNavierStokes, make_slab
        from navierstokes_amd.ns import _pitch, init_vprof
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        p = ns_params("NavierStokes", nx=nx, ny=ny_loc * world, niter=niter, nchk=nchk)
        slab = make_slab(rank, world, ny_loc, halo)
        ld = _pitch(nx)
        ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
        sim = SlabNavierStokes(p, slab, HipBackend(ctx, p, ld), dev, ld, init_vprof(p), schedule=schedule)
        assert sim.poisson.schedule == (schedule or "edges")
        iters = [sim.step() for _ in range(steps)]
        torch.cuda.synchronize()
        res = {k: sim.owned(k)[:, : {"Vx": nx + 1}.get(k, nx)].cpu() for k in ("Pr", "Vx", "Vy", "C", "dP")}
        res["iters"], res["hist"] = iters, sim.hist
        torch.save(res, os.path.join(out, f"s{rank}.pt"))
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,halo,schedule,nx,ny_loc", [(2, 8, "plain", 45, 40), (3, 15, "plain", 54, 30),
                                                          (2, 8, None, 45, 40), (3, 12, "edges", 54, 30),
                                                          (3, 14, "edges", 70, 44), (4, 5, "edges", 33, 24)])
def test_slab_time_steps_on_gpu(tmp_path, world, halo, schedule, nx, ny_loc):
    """plain: passes, then the exchange; edges (the default, None): the last launch of every
    block produces the rows to be sent first, on a second stream, and leaves the halo rows to the
    exchange.  Results must not change by a bit."""
    from oracle.ns_oracle import OracleSim
    steps, niter, nchk = 3, 400, 100
    mp.spawn(_worker, args=(world, _free_port(), halo, schedule, nx, ny_loc, steps, niter, nchk, str(tmp_path)),
             nprocs=world, join=True)
    p = ns_params("NavierStokes", nx=nx, ny=ny_loc * world, niter=niter, nchk=nchk)
    ref = OracleSim(p)
    iters = [ref.step() for _ in range(steps)]
    parts = [torch.load(os.path.join(tmp_path, f"s{r}.pt")) for r in range(world)]
    for q in parts:
        assert q["iters"] == iters
        assert [h[1] for h in q["hist"]] == ref.err_evo
    for k in ("Pr", "Vx", "Vy", "C"):
        got = torch.cat([q[k] for q in parts]).numpy().T
        assert np.array_equal(got, getattr(ref, k)), k
    got = torch.cat([q["dP"] for q in parts]).numpy().T
    assert np.array_equal(got[1:-1, 1:-1], ref.dPrdtau)


def test_rccl_branch_of_every_schedule_to_self():
    """The torch.distributed/RCCL branch of the two slab schedules as far as one GPU can take it: a
    one-rank `nccl` process group, every isend/irecv of an interior rank addressed to itself, against
    the same run with the halo rows refilled by device copies (tools/rccl_self_rehearsal.py).  Runs in
    a process of its own (the process group is process-wide) under a time limit, so that a stalled
    collective ends the test instead of the suite."""
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "rccl_self_rehearsal.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-u", tool, "95", "60", "8", "40"], env=env, timeout=100,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    for schedule in ("plain", "edges"):
        assert f"{schedule:6s} RCCL-to-self == loopback: True" in r.stdout, r.stdout[-3000:]
    assert "all-reduce keeps NaN: True" in r.stdout
