"""Slab decomposition of the Poisson loop (navierstokes_amd/dist.py) on 2 and 3 CPU ranks under
gloo.  The compute backend is a stand-in built on the oracle's loops, injected from here; what
is under test is the window/halo bookkeeping, the exchange and the NaN-keeping MAX all-reduce:
the gathered result must equal the single-domain oracle bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from navierstokes_amd import ns_params
from navierstokes_amd.dist import SlabPoisson, allreduce_maxabs, exchange_rows, make_slab


class OracleWindowBackend:
    """`count` passes of NavierStokes.jl:85-87 on a window, with the oracle's kernels."""

    def __init__(self, p):
        from oracle.ns_oracle import Oracle
        self.o, self.p = Oracle(), p

    @staticmethod
    def _f(t, nx):     # (ny_win, ld) torch tensor -> Fortran (nx, ny_win) copy
        return np.asfortranarray(t[:, :nx].numpy().T)

    def poisson_iters(self, Pr, dP, divV, slab, count):
        p, o, nx = self.p, self.o, self.p.nx
        a, d, v = self._f(Pr, nx), self._f(dP, nx), self._f(divV, nx)
        din = np.asfortranarray(d[1:-1, 1:-1])
        lim = (slab.nyg if p.full_rows else min(nx, slab.nyg)) - slab.iy0
        lim = max(0, min(lim, slab.ny_win))
        for _ in range(count):
            o.update_dPrdtau(a, din, v, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
            o.update_Pr(a, din, p.dtau)
            a[0, :lim] = a[1, :lim]            # bc_x! on global rows <= its launch range
            a[-1, :lim] = a[-2, :lim]
            if slab.iy0 == 0:
                a[:, 0] = a[:, 1]              # bc_yval!, physical edges only
            if slab.iy0 + slab.ny_win == slab.nyg:
                a[:, -1] = 0.0
        d[1:-1, 1:-1] = din
        Pr[:, :nx] = torch.from_numpy(np.ascontiguousarray(a.T))
        dP[:, :nx] = torch.from_numpy(np.ascontiguousarray(d.T))

    def residual_max(self, Pr, divV, slab):
        p, nx = self.p, self.p.nx
        a, v = self._f(Pr, nx), self._f(divV, nx)
        Rp = np.zeros((nx - 2, slab.ny_win - 2), order="F")
        self.o.compute_res(Rp, a, v, p.rho, p.dt, p.dx, p.dy)
        lo, hi = max(slab.own_lo, 1), min(slab.own_hi, slab.ny_win - 1)    # interior owned rows
        return float(np.abs(Rp[:, lo - 1:hi - 1]).max()) if hi > lo else 0.0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, halo, nx, ny_loc, niter, nchk, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nyg = ny_loc * world
        p = ns_params("NavierStokes", nx=nx, ny=nyg, niter=niter, nchk=nchk)
        slab = make_slab(rank, world, ny_loc, halo)
        rng = np.random.default_rng(42)                       # same global fields on every rank
        gdiv = rng.standard_normal((nyg, nx)) * 0.05
        gPr = rng.standard_normal((nyg, nx)) * 0.01
        ld = nx + 3
        def win(g):
            t = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
            t[:, :nx] = torch.from_numpy(g[slab.iy0:slab.iy0 + slab.ny_win])
            return t
        Pr, divV = win(gPr), win(gdiv)
        dP = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
        # halos start stale on purpose: the first exchange must repair them
        if slab.lo_halo:
            Pr[: slab.lo_halo] = 777.0
        if slab.hi_halo:
            Pr[slab.own_hi:] = -777.0
        exchange_rows((Pr,), slab)
        sp = SlabPoisson(p, slab, OracleWindowBackend(p), Pr, dP, divV)
        done, hist = sp.solve()
        own = slice(slab.own_lo, slab.own_hi)
        torch.save({"Pr": Pr[own, :nx].clone(), "dP": dP[own, :nx].clone(), "done": done, "hist": hist},
                   os.path.join(out, f"r{rank}.pt"))
        # NaN must survive the MAX all-reduce
        v = allreduce_maxabs(float("nan") if rank == world - 1 else 1.0, torch.device("cpu"))
        assert v != v
        v = allreduce_maxabs(float(rank + 1), torch.device("cpu"))
        assert v == float(world)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(2, 4), (2, 7), (3, 5)])
def test_slab_poisson_matches_single_domain(tmp_path, world, halo):
    from oracle.ns_oracle import Oracle
    nx, ny_loc, niter, nchk = 17, 12, 60, 20
    mp.spawn(_worker, args=(world, _free_port(), halo, nx, ny_loc, niter, nchk, str(tmp_path)), nprocs=world, join=True)
    nyg = ny_loc * world
    p = ns_params("NavierStokes", nx=nx, ny=nyg, niter=niter, nchk=nchk)
    rng = np.random.default_rng(42)
    gdiv = rng.standard_normal((nyg, nx)) * 0.05
    gPr = rng.standard_normal((nyg, nx)) * 0.01
    Pr, dv = np.asfortranarray(gPr.T), np.asfortranarray(gdiv.T)
    dP, Rp = np.zeros((nx - 2, nyg - 2), order="F"), np.zeros((nx - 2, nyg - 2), order="F")
    it, ih, eh = Oracle().poisson(Pr, dP, Rp, dv, p)
    parts = [torch.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    for part in parts:
        assert part["done"] == it
        assert [h[0] for h in part["hist"]] == ih and [h[1] for h in part["hist"]] == eh
    gotPr = torch.cat([q["Pr"] for q in parts]).numpy().T
    gotdP = torch.cat([q["dP"] for q in parts]).numpy().T
    assert np.array_equal(gotPr, Pr)
    assert np.array_equal(gotdP[1:-1, 1:-1], dP)
