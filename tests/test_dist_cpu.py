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

    # ---- whole-step kernels on windows (used by SlabNavierStokes) -------------------------
    def _run(self, tensors, widths, fn):
        """copy windows to Fortran arrays, call fn(*arrays), copy back"""
        arrs = [self._f(t, w) for t, w in zip(tensors, widths)]
        fn(*arrs)
        for t, w, a in zip(tensors, widths, arrs):
            t[:, :w] = torch.from_numpy(np.ascontiguousarray(a.T))

    def update_tau(self, txx, tyy, txy, Vx, Vy, s):
        p, nx = self.p, self.p.nx
        self._run((txx, tyy, txy, Vx, Vy), (nx, nx, nx - 1, nx + 1, nx),
                  lambda a, b, c, d, e: self.o.update_tau(a, b, c, d, e, p.mu, p.dx, p.dy))

    def predict_V(self, Vx, Vy, txx, tyy, txy, s):
        p, nx = self.p, self.p.nx
        self._run((Vx, Vy, txx, tyy, txy), (nx + 1, nx, nx, nx, nx - 1),
                  lambda a, b, c, d, e: self.o.predict_V(a, b, c, d, e, p.rho, p.g, p.dt, p.dx, p.dy))

    def set_sphere(self, Cc, Vx, Vy, s):
        p, nx = self.p, self.p.nx
        self._run((Cc, Vx, Vy), (nx, nx + 1, nx),
                  lambda a, b, c: self.o.set_sphere(a, b, c, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly,
                                                    p.dx, p.dy, iy0=s.iy0))

    def update_divV(self, divV, Vx, Vy, s):
        p, nx = self.p, self.p.nx
        self._run((divV, Vx, Vy), (nx, nx + 1, nx), lambda a, b, c: self.o.update_divV(a, b, c, p.dx, p.dy))

    def correct_V(self, Vx, Vy, Pr, s):
        p, nx = self.p, self.p.nx
        self._run((Vx, Vy, Pr), (nx + 1, nx, nx), lambda a, b, c: self.o.correct_V(a, b, c, p.dt, p.rho, p.dx, p.dy))

    def set_bc_Vel(self, Vx, Vy, Vprof, s):
        """NavierStokes.jl:179-184 on a window: only the physical edges, bc_x! on global rows."""
        p, nx = self.p, self.p.nx
        bottom, top = s.iy0 == 0, s.iy0 + s.ny_win == s.nyg
        if bottom:
            Vx[0, :nx + 1] = Vx[1, :nx + 1]
        if top:
            Vx[-1, :nx + 1] = Vx[-2, :nx + 1]
        lim = ((s.nyg + 1) if p.full_rows else min(nx, s.nyg + 1)) - s.iy0
        lim = max(0, min(lim, s.ny_win + 1))
        Vy[:lim, 0] = Vy[:lim, 1]
        Vy[:lim, nx - 1] = Vy[:lim, nx - 2]
        if bottom:
            Vy[0, :nx] = Vprof
        if top:
            Vy[-1, :nx] = Vy[-2, :nx]

    def advect(self, Vx, Vx_o, Vy, Vy_o, Cc, C_o, s):
        p, nx = self.p, self.p.nx
        self._run((Vx, Vx_o, Vy, Vy_o, Cc, C_o), (nx + 1, nx + 1, nx, nx, nx, nx),
                  lambda a, b, c, d, e, f: self.o.advect(a, b, c, d, e, f, p.dt, p.dx, p.dy))

    def maxabs(self, t):
        return float(t.abs().max())

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
        assert sp.schedule == "plain"          # the edges-first schedule needs the HIP backend
        done, hist = sp.solve()
        assert sp.exchanges == sum(-(-c // halo) for c in [nchk] * (done // nchk) + [done % nchk])
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


@pytest.mark.parametrize("world,halo", [(2, 4), (2, 7), (3, 5), (3, 6), (4, 3)])
def test_slab_poisson_matches_single_domain(tmp_path, world, halo):
    """Passes, then a blocking exchange; block lengths that do and do not divide the check interval;
    a rank that owns exactly 2*halo rows."""
    from oracle.ns_oracle import Oracle
    nx, ny_loc, niter, nchk = 17, 12, 60, 20
    mp.spawn(_worker, args=(world, _free_port(), halo, nx, ny_loc, niter, nchk, str(tmp_path)), nprocs=world,
             join=True)
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


def _step_worker(rank, world, port, halo, nx, ny_loc, steps, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from navierstokes_amd.dist import SlabNavierStokes
        from navierstokes_amd.ns import init_vprof
        nyg = ny_loc * world
        p = ns_params("NavierStokes", nx=nx, ny=nyg, niter=40, nchk=10)
        slab = make_slab(rank, world, ny_loc, halo)
        sim = SlabNavierStokes(p, slab, OracleWindowBackend(p), torch.device("cpu"), nx + 1, init_vprof(p))
        iters = [sim.step() for _ in range(steps)]
        res = {k: sim.owned(k)[:, : {"Vx": nx + 1}.get(k, nx)].clone() for k in ("Pr", "Vx", "Vy", "C", "dP")}
        res["iters"], res["hist"] = iters, sim.hist
        torch.save(res, os.path.join(out, f"s{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,halo", [(2, 6), (3, 5)])
def test_slab_time_steps_match_single_domain(tmp_path, world, halo):
    """Whole time steps (NavierStokes.jl:77-100) on slabs == the single-domain oracle."""
    from oracle.ns_oracle import OracleSim
    nx, ny_loc, steps = 13, 12, 3
    mp.spawn(_step_worker, args=(world, _free_port(), halo, nx, ny_loc, steps, str(tmp_path)), nprocs=world,
             join=True)
    p = ns_params("NavierStokes", nx=nx, ny=ny_loc * world, niter=40, nchk=10)
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


def test_slab_geometry_rejects_unusable_halos():
    """A zero halo with more than one slab could never advance the Poisson loop, and a slab time
    step needs halo rows for the back-trace of advect!."""
    from navierstokes_amd.dist import SlabNavierStokes, make_slab
    with pytest.raises(ValueError, match="at least 1 row"):
        make_slab(0, 2, 40, 0)
    with pytest.raises(ValueError, match="larger than"):
        make_slab(0, 2, 40, 41)
    assert make_slab(0, 1, 40, 0).halo == 0 and make_slab(0, 1, 40, 9).halo == 0
    p = ns_params("NavierStokes", nx=9, ny=80)
    with pytest.raises(ValueError, match="at least 4 halo rows"):
        SlabNavierStokes(p, make_slab(0, 2, 40, 3), None, "cpu", 9, np.zeros(9))
    z = torch.zeros((60, 9), dtype=torch.float64)
    with pytest.raises(ValueError, match="unknown schedule"):
        SlabPoisson(p, make_slab(0, 2, 40, 20), None, z, z, z, schedule="bands")
    with pytest.raises(ValueError, match="edges-first schedule needs"):
        SlabPoisson(p, make_slab(0, 2, 40, 20), None, z, z, z, schedule="edges")     # CPU tensors, no HIP backend
    assert SlabPoisson(p, make_slab(0, 2, 40, 20), None, z, z, z).schedule == "plain"
