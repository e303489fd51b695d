"""Host drivers of the bubble / drop solvers on libnsk.

DNS       mirrors ``main(; do_vis, do_save)`` of /root/reference/scripts/DNS.jl:14-116 (parameters
          :16-41, allocation :52-62, loop :70-113).  DNS.jl does not run as written; the kernels
          called are the algorithm it transcribes, scripts/code1.m:33-101.
Code3     mirrors the script body of /root/reference/scripts/code3_fun.jl:296-494.
PyTorch provides device memory and the stream only.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


class DArr:
    """Column-major (n1 x n2) Float64 device array (n2 = 1 for vectors)."""

    def __init__(self, n1: int, n2: int = 1, device="cuda:0", fill: float = 0.0):
        self.n1, self.n2 = n1, n2
        self.t = torch.full((n2, n1), fill, dtype=torch.float64, device=device)

    @property
    def ptr(self):
        return C.c_void_p(self.t.data_ptr())

    def set(self, host: np.ndarray) -> None:
        host = np.asarray(host, dtype=np.float64).reshape(self.n1, self.n2, order="F")
        self.t.copy_(torch.from_numpy(np.ascontiguousarray(host.T)))

    def get(self) -> np.ndarray:
        a = np.asfortranarray(self.t.cpu().numpy().T)
        return a[:, 0].copy() if self.n2 == 1 else a

    def copy_from(self, other: "DArr") -> None:
        self.t.copy_(other.t)


@dataclass
class DNSParams:
    """DNS.jl:16-45 (code1.m:8-13 has maxit=200)."""
    Lx: float = 1.0
    Ly: float = 1.0
    gx: float = 0.0
    gy: float = -100.0
    rho1: float = 1.0
    rho2: float = 2.0
    m0: float = 0.01
    unorth: float = 0.0
    usouth: float = 0.0
    veast: float = 0.0
    vwest: float = 0.0
    rad: float = 0.15
    xcenter: float = 0.5
    ycenter: float = 0.7
    nx: int = 32
    ny: int = 32
    dt: float = 0.00125
    nsteps: int = 100
    nvis: int = 10
    maxit: int = 400
    maxerr: float = 1e-3
    beta: float = 1.2
    # "sor": lexicographic SOR (DNS.jl:83-93 / code1.m:68-78, the reference's result);
    # "redblack": red/black SOR (code3_perf_gpu.jl:129-183); "pt": damped pseudo-transient
    # iteration (code1_perf_gpu.jl:98-118, 228-242) with damp / dtau = cfl_tau*dy (:166,180)
    pressure_solver: str = "sor"
    damp: float = 0.72
    cfl_tau: float = 0.9 / 2 ** 0.5
    # cell-centre coordinates that set_density! tests against: "code1" = dx*(i-1.5) (code1.m:22),
    # "linrange" = LinRange(-0.5dx, Lx+0.5dx, nx+2) as DNS.jl:46-47 builds them (the two differ in the
    # last bit of some coordinates, which can move a cell across the drop's edge)
    grid: str = "code1"

    @classmethod
    def dns_jl(cls, **kw) -> "DNSParams":
        """scripts/DNS.jl:16-47 as written: maxit 400, LinRange coordinates."""
        return cls(**{"grid": "linrange", "maxit": 400, **kw})

    @classmethod
    def dns_loop_jl(cls, **kw) -> "DNSParams":
        """scripts/DNS_loop.jl:20-58: maxit 200 and the time step from its two CFL numbers,
        dt = min(CFL_visc*dy^2*rho1/m0, CFL_adv*dy/1.0) (:43-52), LinRange coordinates."""
        q = cls(**{"grid": "linrange", "maxit": 200, **kw})
        if "dt" not in kw:
            cfl_visc, cfl_adv = 1 / 4.1, 1.0
            q.dt = min(cfl_visc * (q.dy * q.dy) * q.rho1 / q.m0, cfl_adv * q.dy / 1.0)
        return q

    def centres(self):
        """xc, yc of DNS.jl:46-47 / code1.m:22 (see `grid`)."""
        if self.grid == "linrange":
            return (linrange(-0.5 * self.dx, self.Lx + 0.5 * self.dx, self.nx + 2),
                    linrange(-0.5 * self.dy, self.Ly + 0.5 * self.dy, self.ny + 2))
        if self.grid != "code1":
            raise ValueError(f"unknown grid {self.grid!r}")
        return cell_centres(self.nx, self.dx), cell_centres(self.ny, self.dy)

    @property
    def dtau(self):
        return self.cfl_tau * self.dy

    @property
    def dx(self):
        return self.Lx / self.nx

    @property
    def dy(self):
        return self.Ly / self.ny


def _ctx(device):
    if not torch.cuda.is_available():
        raise _lib.NskError("no GPU visible: the bubble drivers have no CPU path")
    dev = torch.device(device)
    torch.cuda.set_device(dev)
    return dev, _lib.Context(dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)


def linrange(a: float, b: float, n: int) -> np.ndarray:
    """Julia's LinRange(a, b, n) element by element: (1-t)*a + t*b with t = (i-1)/(n-1)."""
    return np.array([(1 - i / (n - 1)) * a + (i / (n - 1)) * b for i in range(n)]) if n > 1 else np.array([a])


def cell_centres(n: int, d: float) -> np.ndarray:
    """x(i) = d*(i-1.5), i = 1..n+2 (code1.m:22, code3_fun.jl:328-329)."""
    return np.array([d * (i - 1.5) for i in range(1, n + 3)])


class DNS:
    """State and time step of DNS.jl's drop problem."""

    def __init__(self, q: DNSParams = DNSParams(), device: str = "cuda:0"):
        self.q = q
        self.dev, self.ctx = _ctx(device)
        nx, ny, d = q.nx, q.ny, self.dev
        A = lambda a, b=1, fill=0.0: DArr(a, b, d, fill)
        # DNS.jl:52-62
        self.rho, self.rhot, self.rho_o = A(nx + 2, ny + 2, q.rho1), A(nx + 2, ny + 2), A(nx + 2, ny + 2)
        self.u, self.v, self.p = A(nx + 1, ny + 2), A(nx + 2, ny + 1), A(nx + 2, ny + 2)
        self.ut, self.vt = A(nx + 1, ny + 2), A(nx + 2, ny + 1)
        self.uu, self.vv = A(nx + 1, ny + 1), A(nx + 1, ny + 1)
        self.tmp1, self.tmp2 = A(nx + 2, ny + 2), A(nx + 2, ny + 2)
        self.dPdtau = A(nx + 2, ny + 2)  # code1_perf_gpu.jl:183 (used by pressure_solver="pt")
        self.xc, self.yc = A(nx + 2), A(ny + 2)
        xc, yc = q.centres()
        self.xc.set(xc)
        self.yc.set(yc)
        # DNS.jl:64-67
        self.ctx.bub_set_density(self.rho.ptr, None, q.xcenter, q.ycenter, self.xc.ptr, self.yc.ptr,
                                 q.rad, q.rho2, 0.0, nx, ny)
        self.time = 0.0
        self.sweeps = 0
        # True: one C call per time step (nsk_bub_dns_step: the same kernels, one wait per step);
        # False: one C call per kernel, as the reference's driver is written
        self.fused = True

    def _c_args(self):
        q = self.q
        st = _lib.CDnsState(u=self.u.ptr, v=self.v.ptr, p=self.p.ptr, ut=self.ut.ptr, vt=self.vt.ptr,
                            tmp1=self.tmp1.ptr, tmp2=self.tmp2.ptr, r=self.rho.ptr, ro=self.rho_o.ptr,
                            rt=self.rhot.ptr, uu=self.uu.ptr, vv=self.vv.ptr)
        pr = _lib.CBubParams(nx=q.nx, ny=q.ny, maxit=q.maxit, Lx=q.Lx, Ly=q.Ly, gx=q.gx, gy=q.gy, rho1=q.rho1,
                             rho2=q.rho2, m1=q.m0, m2=0.0, sigma=0.0, rro=0.0, usouth=q.usouth, unorth=q.unorth,
                             vwest=q.vwest, veast=q.veast, dt=q.dt, dx=q.dx, dy=q.dy, maxError=q.maxerr, beta=q.beta)
        return st, pr

    def run(self, nsteps: int):
        """nsteps time steps (the loop DNS.jl:70-102) in one C call with one wait at its end; the sweeps of
        every step.  Other pressure solvers than the lexicographic SOR: a loop over step()."""
        if not (self.fused and self.q.pressure_solver == "sor"):
            return [self.step() for _ in range(nsteps)]
        st, pr = self._c_args()
        sw = (C.c_int * max(nsteps, 1))()
        self.ctx.bub_dns_run(C.byref(st), C.byref(pr), nsteps, sw)
        self.time += nsteps * self.q.dt
        out = list(sw)[:nsteps]
        if out:
            self.sweeps = out[-1]
        return out

    def step(self) -> int:
        """DNS.jl:71-101."""
        q, k, nx, ny = self.q, self.ctx, self.q.nx, self.q.ny
        self.time += q.dt
        if self.fused and q.pressure_solver == "sor":
            st, pr = self._c_args()
            sw = C.c_int(0)
            k.bub_dns_step(C.byref(st), C.byref(pr), C.byref(sw))
            self.sweeps = sw.value
            return sw.value
        k.bub_set_tangential(self.u.ptr, self.v.ptr, q.usouth, q.unorth, q.vwest, q.veast, nx, ny)
        k.bub_temp_vel(self.ut.ptr, self.vt.ptr, self.u.ptr, self.v.ptr, self.rho.ptr, q.dt, q.dx, q.dy,
                       q.gx, q.gy, q.m0, nx, ny)
        k.bub_set_boundary(self.rhot.ptr, self.rho.ptr, 1000.0, nx, ny)
        k.bub_compute_sources(self.tmp1.ptr, self.tmp2.ptr, self.rhot.ptr, self.ut.ptr, self.vt.ptr,
                              q.dt, q.dx, q.dy, nx, ny)
        sw = C.c_int(0)
        if q.pressure_solver == "sor":
            k.bub_solve_pressure(self.p.ptr, self.tmp1.ptr, self.tmp2.ptr, self.rhot.ptr, q.beta, q.dx, q.dy,
                                 nx, ny, q.maxit, q.maxerr, C.byref(sw), None)
        elif q.pressure_solver == "redblack":
            k.bub_solve_pressure_rb(self.p.ptr, self.tmp1.ptr, self.tmp2.ptr, self.rhot.ptr, q.beta, q.dx, q.dy,
                                    nx, ny, q.maxit, q.maxerr, C.byref(sw), None)
        elif q.pressure_solver == "pt":
            k.bub_pt_pressure(self.p.ptr, self.dPdtau.ptr, self.tmp1.ptr, self.rhot.ptr, q.dtau, q.damp,
                              q.dx, q.dy, nx, ny, q.maxit, q.maxerr, C.byref(sw), None)
        else:
            raise ValueError(f"unknown pressure_solver {q.pressure_solver!r}")
        k.bub_correct_vel(self.u.ptr, self.v.ptr, self.ut.ptr, self.vt.ptr, self.p.ptr, self.rho.ptr,
                          q.dt, q.dx, q.dy, nx, ny)
        self.rho_o.copy_from(self.rho)
        k.bub_advect_density(self.rho.ptr, self.rho_o.ptr, self.u.ptr, self.v.ptr, q.dt, q.dx, q.dy, q.m0, nx, ny)
        k.bub_velocity(self.uu.ptr, self.vv.ptr, self.u.ptr, self.v.ptr, nx, ny)
        self.sweeps = sw.value
        return sw.value


def dns_main(do_vis: bool = True, do_save: bool = False, *, q: DNSParams = DNSParams(),
             device: str = "cuda:0", out_dir: str = "out_dns", verbose: bool = True) -> DNS:
    """`main(; do_vis, do_save)` of DNS.jl; saves Vx, Vy, C, dx, dy like DNS.jl:63,105."""
    from scipy.io import savemat
    sim = DNS(q, device)
    nstep = 1
    it = 0
    while it < q.nsteps:
        chunk = min(q.nvis - it % q.nvis, q.nsteps - it)  # up to the next output
        sim.run(chunk)
        it += chunk
        if it % q.nvis == 0:
            if do_save:
                os.makedirs(out_dir, exist_ok=True)
                savemat(os.path.join(out_dir, f"step_{nstep}.mat"),
                        {"Vx": sim.uu.get(), "Vy": sim.vv.get(), "C": sim.rho.get(), "dx": q.dx, "dy": q.dy})
                nstep += 1
            if verbose:
                print(f"time = {sim.time}")
    return sim


@dataclass
class Code3Params:
    """code3_fun.jl:296-309,337,340."""
    Lx: float = 1.0
    Ly: float = 1.0
    gx: float = 0.0
    gy: float = -100.0
    rho1: float = 1.0
    rho2: float = 2.0
    m1: float = 0.01
    m2: float = 0.05
    sigma: float = 10.0
    unorth: float = 0.0
    usouth: float = 0.0
    veast: float = 0.0
    vwest: float = 0.0
    rad: float = 0.15
    xc: float = 0.5
    yc: float = 0.7
    nx: int = 32
    ny: int = 32
    dt0: float = 0.00125
    nstep: int = 300
    maxit: int = 200
    maxError: float = 0.001
    beta: float = 1.2
    Nf: int = 100
    cap: int = 1024         # first capacity of the front vectors; they grow on demand as the reference's do

    @property
    def rro(self):
        return self.rho1

    @property
    def dx(self):
        return self.Lx / self.nx

    @property
    def dy(self):
        return self.Ly / self.ny

    @property
    def dt(self):
        # code3_fun.jl:337 with m == m1 everywhere at that point
        return min(self.dt0, 0.25 * min(self.dx, self.dy) ** 2 * min(self.rho1, self.rho2) / self.m1)


def front_circle(q: Code3Params):
    """set_front!, code3_fun.jl:20-26."""
    l = np.arange(q.Nf + 2)
    xf = np.array([q.xc - q.rad * math.sin(2.0 * math.pi * i / q.Nf) for i in l])
    yf = np.array([q.yc + q.rad * math.cos(2.0 * math.pi * i / q.Nf) for i in l])
    return xf, yf


class Code3:
    """State and time step of code3_fun.jl."""

    GRID = ("u", "v", "p", "ut", "vt", "tmp1", "tmp2", "fx", "fy", "un", "vn", "r", "m", "rn", "mn", "rt")
    FRONT = ("xf", "yf", "xfn", "yfn", "uf", "vf", "tx", "ty", "xs", "ys")

    def __init__(self, q: Code3Params = Code3Params(), device: str = "cuda:0"):
        self.q = q
        self.dev, self.ctx = _ctx(device)
        nx, ny, d = q.nx, q.ny, self.dev
        shp = dict(u=(nx + 1, ny + 2), v=(nx + 2, ny + 1), ut=(nx + 1, ny + 2), vt=(nx + 2, ny + 1),
                   un=(nx + 1, ny + 2), vn=(nx + 2, ny + 1))
        for k in self.GRID:
            setattr(self, k, DArr(*shp.get(k, (nx + 2, ny + 2)), d))
        self.r.t.fill_(q.rho1)       # code3_fun.jl:332-333
        self.m.t.fill_(q.m1)
        self.cap = q.cap
        for k in self.FRONT:
            setattr(self, k, DArr(self.cap, 1, d))
        xf, yf = front_circle(q)
        pad = np.zeros(self.cap)
        pad[: q.Nf + 2] = xf
        self.xf.set(pad)
        pad = np.zeros(self.cap)
        pad[: q.Nf + 2] = yf
        self.yf.set(pad)
        self.Nf = q.Nf
        self.grown = 0               # times the front vectors were enlarged
        self.time = 0.0
        self.sweeps = [0, 0, 0, 0]
        self.fused = True            # one C call per time step (nsk_bub_code3_step) / one per kernel

    def _c_args(self):
        q = self.q
        st = _lib.CCode3State(**{n: getattr(self, n).ptr for n in self.GRID + self.FRONT}, Nf=self.Nf, cap=self.cap)
        pr = _lib.CBubParams(nx=q.nx, ny=q.ny, maxit=q.maxit, Lx=q.Lx, Ly=q.Ly, gx=q.gx, gy=q.gy, rho1=q.rho1,
                             rho2=q.rho2, m1=q.m1, m2=q.m2, sigma=q.sigma, rro=q.rro, usouth=q.usouth,
                             unorth=q.unorth, vwest=q.vwest, veast=q.veast, dt=q.dt, dx=q.dx, dy=q.dy,
                             maxError=q.maxError, beta=q.beta)
        return st, pr

    def _grow_and_regrid(self) -> None:
        """The place where the reference resizes xf, yf while it regrids (code3_fun.jl:462-465): the step
        that reported NSK_ERR_FRONT_CAPACITY is complete except for its regridding and left its averaged
        front (Nf + 2 points) in xs, ys.  Every front vector gets twice the room, the averaged front
        goes back into xf, yf and the regridding runs again (until it fits)."""
        n = self.Nf + 2
        x_old, y_old = self.xs.t[0, :n].clone(), self.ys.t[0, :n].clone()
        while True:
            self.cap *= 2
            self.grown += 1
            for k in self.FRONT:
                setattr(self, k, DArr(self.cap, 1, self.dev))
            self.xf.t[0, :n] = x_old
            self.yf.t[0, :n] = y_old
            nf = C.c_int(0)
            try:
                self.ctx.bub_regrid_front(self.xf.ptr, self.yf.ptr, self.xs.ptr, self.ys.ptr, self.q.dx, self.q.dy,
                                          self.Nf, self.cap, C.byref(nf))
            except _lib.NskError as e:
                if e.code != _lib.ERR_FRONT_CAPACITY:
                    raise
                continue
            self.Nf = nf.value
            return

    def _steps_done(self) -> int:
        n = C.c_int(0)
        self.ctx.bub_code3_run_status(C.byref(n))
        return n.value

    def run(self, nsteps: int):
        """nsteps time steps (the loop code3_fun.jl:354-494) in one C call with one wait at its end; the four
        sweep counts of every step.  A front that outgrows its vectors is given larger ones and the run
        goes on from the step that needed them."""
        if not self.fused:
            return [self.step() for _ in range(nsteps)]
        out = []
        while len(out) < nsteps:
            left = nsteps - len(out)
            st, pr = self._c_args()
            sw = (C.c_int * max(4 * left, 1))()
            try:
                self.ctx.bub_code3_run(C.byref(st), C.byref(pr), left, sw)
                done = left
                self.Nf = st.Nf
            except _lib.NskError as e:
                if e.code != _lib.ERR_FRONT_CAPACITY:
                    raise
                done = self._steps_done() + 1      # the step that needed the room counts: only its regridding is left
                self.Nf = st.Nf
                self._grow_and_regrid()
            out += [list(sw[4 * k: 4 * k + 4]) for k in range(done)]
            self.time += done * self.q.dt
        if out:
            self.sweeps = out[-1]
        return out

    def step(self):
        """code3_fun.jl:356-494."""
        q, k, nx, ny = self.q, self.ctx, self.q.nx, self.q.ny
        dt, dx, dy = q.dt, q.dx, q.dy
        if self.fused:
            st, pr = self._c_args()
            sw = (C.c_int * 4)()
            try:
                k.bub_code3_step(C.byref(st), C.byref(pr), sw)
                self.Nf = st.Nf
            except _lib.NskError as e:
                if e.code != _lib.ERR_FRONT_CAPACITY:
                    raise
                self._grow_and_regrid()
            self.time += dt
            self.sweeps = list(sw)
            return self.sweeps
        for a, b in (("un", "u"), ("vn", "v"), ("rn", "r"), ("mn", "m"), ("xfn", "xf"), ("yfn", "yf")):
            getattr(self, a).copy_from(getattr(self, b))
        sw = [0, 0, 0, 0]
        for sub in range(2):
            k.bub_surface_tension(self.fx.ptr, self.fy.ptr, self.tx.ptr, self.ty.ptr, self.xf.ptr, self.yf.ptr,
                                  q.sigma, dx, dy, nx, ny, self.Nf)
            k.bub_set_tangential(self.u.ptr, self.v.ptr, q.usouth, q.unorth, q.vwest, q.veast, nx, ny)
            k.bub_temp_vel_adv(self.ut.ptr, self.vt.ptr, self.u.ptr, self.v.ptr, self.r.ptr, dt, dx, dy,
                               self.fx.ptr, self.fy.ptr, q.gx, q.gy, q.rro, nx, ny)
            k.bub_temp_vel_diff(self.ut.ptr, self.vt.ptr, self.u.ptr, self.v.ptr, self.r.ptr, dt, dx, dy,
                                self.m.ptr, nx, ny)
            k.bub_set_boundary(self.rt.ptr, self.r.ptr, 1000.0, nx, ny)
            k.bub_compute_sources(self.tmp1.ptr, self.tmp2.ptr, self.rt.ptr, self.ut.ptr, self.vt.ptr, dt, dx, dy, nx, ny)
            a = C.c_int(0)
            k.bub_solve_pressure(self.p.ptr, self.tmp1.ptr, self.tmp2.ptr, self.rt.ptr, q.beta, dx, dy, nx, ny,
                                 q.maxit, q.maxError, C.byref(a), None)
            k.bub_correct_vel(self.u.ptr, self.v.ptr, self.ut.ptr, self.vt.ptr, self.p.ptr, self.r.ptr, dt, dx, dy, nx, ny)
            out = C.c_int(0)
            k.bub_advect_move_front(self.uf.ptr, self.vf.ptr, self.xf.ptr, self.yf.ptr, self.u.ptr, self.v.ptr,
                                    dx, dy, dt, q.Lx, q.Ly, nx, ny, self.Nf, C.byref(out))
            # where the reference prints "Front out of bounds" (code3_fun.jl:186-188) its next access by
            # that point throws; a NaN position fails the printed test but throws all the same
            if out.value or not bool(torch.isfinite(self.xf.t[0, : self.Nf + 2]).all()
                                     and torch.isfinite(self.yf.t[0, : self.Nf + 2]).all()):
                raise _lib.NskError(f"a front point left the box [0,{q.Lx}] x [0,{q.Ly}]; the state is not to be used",
                                    _lib.ERR_FRONT_OUTSIDE)
            k.bub_distribute(self.fx.ptr, self.fy.ptr, self.xf.ptr, self.yf.ptr, q.rho1, q.rho2, dx, dy, nx, ny, self.Nf)
            b = C.c_int(0)
            k.bub_construct_den(self.r.ptr, self.fx.ptr, self.fy.ptr, q.beta, q.maxit, q.maxError, dx, dy, nx, ny, C.byref(b))
            k.bub_update_visco(self.r.ptr, self.m.ptr, q.m1, q.m2, q.rho1, q.rho2, nx, ny)
            sw[2 * sub], sw[2 * sub + 1] = a.value, b.value
        # second-order averaging, code3_fun.jl:445-450
        for a, b in (("u", "un"), ("v", "vn"), ("r", "rn"), ("m", "mn")):
            A = getattr(self, a)
            k.bub_average(A.ptr, getattr(self, b).ptr, A.n1 * A.n2)
        k.bub_average(self.xf.ptr, self.xfn.ptr, self.Nf + 2)
        k.bub_average(self.yf.ptr, self.yfn.ptr, self.Nf + 2)
        nf = C.c_int(0)
        try:
            k.bub_regrid_front(self.xf.ptr, self.yf.ptr, self.xs.ptr, self.ys.ptr, dx, dy, self.Nf, self.cap, C.byref(nf))
            self.Nf = nf.value
        except _lib.NskError as e:
            if e.code != _lib.ERR_FRONT_CAPACITY:
                raise
            self._grow_and_regrid()
        self.time += dt
        self.sweeps = sw
        return sw

    def front(self):
        return self.xf.get()[: self.Nf + 2], self.yf.get()[: self.Nf + 2]
