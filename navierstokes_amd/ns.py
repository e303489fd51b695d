"""Host driver of the projection-method time step: the reference's ``runme()``.

Mirrors /root/reference/scripts/NavierStokes.jl:12-113 (and NavierStokes_Re6.jl, same lines):
same parameters, same allocation list, same call order.  Every ``@parallel f!(...)`` of the
reference is one call into libnsk (``mode="granular"``), or the whole loop body is the single
fused call ``nsk_ns_step`` (``mode="fused"``, default).  PyTorch only provides device memory
and the stream; all arithmetic happens in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np
import torch

from . import _lib
from .params import NSParams, ns_params


def _pitch(nx: int) -> int:
    """Leading dimension for the Poisson fields: even, > nx when nx is odd, whole 128-byte lines."""
    return (nx + 1 + 15) // 16 * 16


class Field:
    """Column-major (n1 x n2) Float64 device array A[ix,iy] with leading dimension ld >= n1."""

    def __init__(self, n1: int, n2: int, device, ld: Optional[int] = None):
        self.n1, self.n2, self.ld = n1, n2, ld or n1
        self.t = torch.zeros((n2, self.ld), dtype=torch.float64, device=device)

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.t.data_ptr())

    def set(self, host: np.ndarray) -> None:
        """`Data.Array(host)`: host is (n1, n2)."""
        assert host.shape == (self.n1, self.n2), (host.shape, self.n1, self.n2)
        self.t[:, : self.n1].copy_(torch.from_numpy(np.ascontiguousarray(host.T)))

    def get(self) -> np.ndarray:
        """`Array(A)`: returns an (n1, n2) Fortran-ordered host copy."""
        return np.asfortranarray(self.t[:, : self.n1].cpu().numpy().T)


def c_params(p: NSParams, *, ld_pr: int = 0, ld_dp: int = 0, dp_embedded: int = 0,
             iy0: int = 0, nyg: Optional[int] = None, ny: Optional[int] = None) -> _lib.CNSParams:
    ny_ = p.ny if ny is None else ny
    return _lib.CNSParams(
        nx=p.nx, ny=ny_, iy0=iy0, nyg=(p.ny if nyg is None else nyg), niter=p.niter, nchk=p.nchk,
        full_rows=int(p.full_rows), ld_pr=ld_pr, ld_dp=ld_dp, dp_embedded=dp_embedded,
        lx=p.lx, ly=p.ly, rho=p.rho, vin=p.vin, psc=p.psc, mu=p.mu, g=p.g, a2=p.a2, b2=p.b2,
        ox=p.ox, oy=p.oy, sinb=p.sinb, cosb=p.cosb, eps_it=p.eps_it, dx=p.dx, dy=p.dy, dt=p.dt,
        damp=p.damp, dtau=p.dtau)


def init_vprof(p: NSParams) -> np.ndarray:
    """NavierStokes.jl:72 -- inflow parabola on LinRange(0.5dx, lx-0.5dx, nx)."""
    a, b = 0.5 * p.dx, p.lx - 0.5 * p.dx
    out = np.empty(p.nx)
    for i in range(p.nx):
        t = i / (p.nx - 1) if p.nx > 1 else 0.0
        x = (1 - t) * a + t * b
        out[i] = 4 * p.vin * x / p.lx * (1.0 - x / p.lx)
    return out


class NavierStokes:
    """Device state + time stepping of one domain on one GPU."""

    HIST_CAP = 1024

    def __init__(self, p: NSParams, device: str = "cuda:0", mode: str = "fused",
                 pitched: bool = True):
        if not torch.cuda.is_available():
            raise _lib.NskError("no GPU visible: the NavierStokes driver has no CPU path")
        assert mode in ("fused", "granular")
        self.p, self.mode = p, mode
        self.dev = torch.device(device)
        torch.cuda.set_device(self.dev)
        self.ctx = _lib.Context(self.dev.index or 0, torch.cuda.current_stream(self.dev).cuda_stream)
        nx, ny = p.nx, p.ny
        pitched = pitched and mode == "fused"
        ld = _pitch(nx) if pitched else nx
        self.pitched = pitched
        d = self.dev
        # NavierStokes.jl:58-70
        self.Pr = Field(nx, ny, d, ld)
        self.dPrdtau = Field(nx, ny, d, ld) if pitched else Field(nx - 2, ny - 2, d)
        self.C = Field(nx, ny, d)
        self.C_o = Field(nx, ny, d)
        self.txx = Field(nx, ny, d)
        self.tyy = Field(nx, ny, d)
        self.txy = Field(nx - 1, ny - 1, d)
        self.Vx = Field(nx + 1, ny, d)
        self.Vy = Field(nx, ny + 1, d)
        self.Vx_o = Field(nx + 1, ny, d)
        self.Vy_o = Field(nx, ny + 1, d)
        self.divV = Field(nx, ny, d, ld)
        self.Rp = Field(nx - 2, ny - 2, d)
        # NavierStokes.jl:72-74
        vprof = init_vprof(p)
        self.Vprof = Field(nx, 1, d)
        self.Vprof.set(vprof.reshape(nx, 1))
        vy = np.zeros((nx, ny + 1), order="F")
        vy[:, 0] = vprof
        self.Vy.set(vy)
        if p.g != 0.0:
            yc = np.array([(1 - j / (ny - 1)) * (-(p.ly - p.dy) / 2) + j / (ny - 1) * ((p.ly - p.dy) / 2)
                           for j in range(ny)])
            self.Pr.set(np.asfortranarray(np.broadcast_to(-(yc - p.ly / 2) * p.rho * p.g, (nx, ny))))
        self.cp = c_params(p, ld_pr=ld if pitched else 0, dp_embedded=1 if pitched else 0)
        self.cs = _lib.CNSState(**{k: getattr(self, k).t.data_ptr() for k, _ in _lib.CNSState._fields_})
        self.it = 0
        self.err_evo: list[float] = []
        self.iter_evo: list[int] = []

    # -- the body of the time loop, NavierStokes.jl:78-100 -------------------------------
    def step(self) -> int:
        self.it += 1
        hist_e = (C.c_double * self.HIST_CAP)()
        hist_i = (C.c_int * self.HIST_CAP)()
        nh, iters = C.c_int(0), C.c_int(0)
        if self.mode == "fused":
            self.ctx.ns_step(C.byref(self.cs), C.byref(self.cp), C.byref(iters), hist_e, hist_i,
                             self.HIST_CAP, C.byref(nh))
            n = min(nh.value, self.HIST_CAP)
            self.err_evo = list(hist_e[:n])
            self.iter_evo = list(hist_i[:n])
            return iters.value
        p, k, f = self.p, self.ctx, self
        nx, ny = p.nx, p.ny
        k.update_tau(f.txx.ptr, f.tyy.ptr, f.txy.ptr, f.Vx.ptr, f.Vy.ptr, p.mu, p.dx, p.dy, nx, ny)
        k.predict_V(f.Vx.ptr, f.Vy.ptr, f.txx.ptr, f.tyy.ptr, f.txy.ptr, p.rho, p.g, p.dt, p.dx, p.dy, nx, ny)
        k.set_sphere(f.C.ptr, f.Vx.ptr, f.Vy.ptr, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly,
                     p.dx, p.dy, nx, ny, 0)
        k.update_divV(f.divV.ptr, f.Vx.ptr, f.Vy.ptr, p.dx, p.dy, nx, ny)
        self.err_evo, self.iter_evo = [], []
        done = 0
        for it in range(1, p.niter + 1):
            k.update_dPrdtau(f.Pr.ptr, f.dPrdtau.ptr, f.divV.ptr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, nx, ny)
            k.update_Pr(f.Pr.ptr, f.dPrdtau.ptr, p.dtau, nx, ny)
            k.set_bc_Pr(f.Pr.ptr, 0.0, nx, ny, int(p.full_rows), 0, ny)
            done = it
            if p.nchk > 0 and it % p.nchk == 0:
                k.compute_res(f.Rp.ptr, f.Pr.ptr, f.divV.ptr, p.rho, p.dt, p.dx, p.dy, nx, ny)
                m = C.c_double(0.0)
                k.maxabs(f.Rp.ptr, (nx - 2) * (ny - 2), C.byref(m))
                err = m.value * (p.ly * p.ly) / p.psc
                self.err_evo.append(err)
                self.iter_evo.append(it)
                if err < p.eps_it or not np.isfinite(err):
                    break
        k.correct_V(f.Vx.ptr, f.Vy.ptr, f.Pr.ptr, p.dt, p.rho, p.dx, p.dy, nx, ny)
        k.set_sphere(f.C.ptr, f.Vx.ptr, f.Vy.ptr, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly,
                     p.dx, p.dy, nx, ny, 0)
        k.set_bc_Vel(f.Vx.ptr, f.Vy.ptr, f.Vprof.ptr, nx, ny, int(p.full_rows), 0, ny)
        k.copy(f.Vx_o.ptr, f.Vx.ptr, (nx + 1) * ny)
        k.copy(f.Vy_o.ptr, f.Vy.ptr, nx * (ny + 1))
        k.copy(f.C_o.ptr, f.C.ptr, nx * ny)
        k.advect(f.Vx.ptr, f.Vx_o.ptr, f.Vy.ptr, f.Vy_o.ptr, f.C.ptr, f.C_o.ptr, p.dt, p.dx, p.dy, nx, ny, 0, ny)
        k.sync()
        return done

    def host_fields(self) -> dict:
        """The four arrays the reference saves (NavierStokes.jl:75,109)."""
        return {"Pr": self.Pr.get(), "Vx": self.Vx.get(), "Vy": self.Vy.get(), "C": self.C.get()}

    def dPrdtau_host(self) -> np.ndarray:
        a = self.dPrdtau.get()
        return np.asfortranarray(a[1:-1, 1:-1]) if self.pitched else a

    # -- restart files --------------------------------------------------------------------
    # What the next step reads from the previous one: Vx, Vy, C (advected), Pr, and dPrdτ,
    # which the time loop never resets (NavierStokes.jl:58,85).  The reference saves only
    # Pr, Vx, Vy, C (:109), so a run continued from its .mat files is not the same run;
    # this file is.
    RESTART_KEYS = ("Pr", "Vx", "Vy", "C", "dPrdtau")

    def save_restart(self, path: str) -> None:
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        p = self.p
        np.savez(path, **self.host_fields(), dPrdtau=self.dPrdtau_host(), it=self.it,
                 grid=np.array([p.nx, p.ny]), scalars=np.array([p.dx, p.dy, p.dt, p.dtau, p.damp, p.rho, p.mu]))

    def load_restart(self, path: str) -> None:
        with np.load(path, allow_pickle=False) as z:
            p = self.p
            if tuple(z["grid"]) != (p.nx, p.ny):
                raise ValueError(f"restart file is for a {tuple(z['grid'])} grid, this run is {(p.nx, p.ny)}")
            want = np.array([p.dx, p.dy, p.dt, p.dtau, p.damp, p.rho, p.mu])
            if not np.array_equal(z["scalars"], want):
                raise ValueError("restart file was written with other parameters (dx, dy, dt, dtau, damp, rho, mu)")
            for k in ("Pr", "Vx", "Vy", "C"):
                getattr(self, k).set(np.asfortranarray(z[k]))
            d = np.asfortranarray(z["dPrdtau"])
            if self.pitched:
                full = np.zeros((p.nx, p.ny), order="F")
                full[1:-1, 1:-1] = d
                d = full
            self.dPrdtau.set(d)
            self.it = int(z["it"])


def save_mat(path: str, fields: dict, dx: float, dy: float) -> None:
    """`matwrite("out_vis/step_N.mat", Dict("Pr","Vx","Vy","C","dx","dy"))` (NavierStokes.jl:75,109)."""
    from scipy.io import savemat
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    savemat(path, {**fields, "dx": dx, "dy": dy})


def runme(n: int = 1, do_vis: bool = True, do_save: bool = False, *, script: str = "NavierStokes_Re6",
          nt: Optional[int] = None, device: str = "cuda:0", out_dir: str = "out_vis",
          mode: str = "fused", verbose: bool = True, restart_from: Optional[str] = None,
          restart_every: int = 0, **overrides) -> NavierStokes:
    """`runme(; n, do_vis, do_save)` of the reference drivers.  do_vis is accepted for interface
    compatibility; plotting is left to vis/vis_navier_stokes.m on the saved .mat files."""
    p = ns_params(script, n=n, **overrides)
    sim = NavierStokes(p, device=device, mode=mode)
    if restart_from:
        sim.load_restart(restart_from)
    elif do_save:
        save_mat(os.path.join(out_dir, "step_0.mat"), sim.host_fields(), p.dx, p.dy)
    for it in range(sim.it + 1, (nt if nt is not None else p.nt) + 1):
        if verbose:
            print(f"#it = {it}")
        sim.step()
        if verbose:
            for i, e in zip(sim.iter_evo, sim.err_evo):
                print(f"  #iter = {i}, err = {e:1.3e}")
        if do_save and it % p.nsave == 0:
            save_mat(os.path.join(out_dir, f"step_{it}.mat"), sim.host_fields(), p.dx, p.dy)
        if restart_every and it % restart_every == 0:
            sim.save_restart(os.path.join(out_dir, "restart.npz"))
    return sim
