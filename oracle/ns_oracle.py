"""ctypes wrapper of oracle/ns_oracle.c -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may import this.
Arrays are numpy Float64, Fortran order, shape (size(A,1), size(A,2)) as in the Julia source.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_D, _I, _P = C.c_double, C.c_int, C.c_void_p


class OParams(C.Structure):
    """struct ns_params of ns_oracle.c."""
    _fields_ = [("nx", _I), ("ny", _I), ("niter", _I), ("nchk", _I), ("full_rows", _I),
                ("lx", _D), ("ly", _D), ("rho", _D), ("vin", _D), ("psc", _D), ("mu", _D), ("g", _D),
                ("a2", _D), ("b2", _D), ("ox", _D), ("oy", _D), ("sinb", _D), ("cosb", _D),
                ("eps_it", _D), ("dx", _D), ("dy", _D), ("dt", _D), ("damp", _D), ("dtau", _D)]


class OState(C.Structure):
    _fields_ = [(n, _P) for n in ("Pr", "dPrdtau", "C", "C_o", "txx", "tyy", "txy", "Vx", "Vy",
                                  "Vx_o", "Vy_o", "divV", "Rp", "Vprof")]


def build(force: bool = False) -> None:
    if force or not (os.path.exists(os.path.join(_HERE, "libnsoracle.so"))
                     and os.path.exists(os.path.join(_HERE, "libnsoracle_omp.so"))):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


def _ptr(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.f_contiguous, (a.dtype, a.flags)
    return a.ctypes.data_as(_P)


def farray(n1: int, n2: int) -> np.ndarray:
    return np.zeros((n1, n2), order="F")


class Oracle:
    def __init__(self, omp: bool = False):
        build()
        self.lib = C.CDLL(os.path.join(_HERE, "libnsoracle_omp.so" if omp else "libnsoracle.so"))
        L = self.lib
        L.ns_maxabs.restype = _D
        L.ns_maxabs.argtypes = [_P, C.c_size_t]
        L.ns_poisson.restype = _I
        L.ns_step.restype = _I

    # scalar-argument kernels ---------------------------------------------------------
    def update_tau(self, txx, tyy, txy, Vx, Vy, mu, dx, dy):
        nx, ny = txx.shape
        self.lib.ns_update_tau(_ptr(txx), _ptr(tyy), _ptr(txy), _ptr(Vx), _ptr(Vy), _D(mu), _D(dx), _D(dy), nx, ny)

    def predict_V(self, Vx, Vy, txx, tyy, txy, rho, g, dt, dx, dy):
        nx, ny = txx.shape
        self.lib.ns_predict_V(_ptr(Vx), _ptr(Vy), _ptr(txx), _ptr(tyy), _ptr(txy), _D(rho), _D(g), _D(dt), _D(dx), _D(dy), nx, ny)

    def update_divV(self, divV, Vx, Vy, dx, dy):
        nx, ny = divV.shape
        self.lib.ns_update_divV(_ptr(divV), _ptr(Vx), _ptr(Vy), _D(dx), _D(dy), nx, ny)

    def update_dPrdtau(self, Pr, dP, divV, rho, dt, dtau, damp, dx, dy):
        nx, ny = Pr.shape
        self.lib.ns_update_dPrdtau(_ptr(Pr), _ptr(dP), _ptr(divV), _D(rho), _D(dt), _D(dtau), _D(damp), _D(dx), _D(dy), nx, ny)

    def update_Pr(self, Pr, dP, dtau):
        nx, ny = Pr.shape
        self.lib.ns_update_Pr(_ptr(Pr), _ptr(dP), _D(dtau), nx, ny)

    def compute_res(self, Rp, Pr, divV, rho, dt, dx, dy):
        nx, ny = Pr.shape
        self.lib.ns_compute_res(_ptr(Rp), _ptr(Pr), _ptr(divV), _D(rho), _D(dt), _D(dx), _D(dy), nx, ny)

    def maxabs(self, a) -> float:
        return float(self.lib.ns_maxabs(_ptr(a), a.size))

    def correct_V(self, Vx, Vy, Pr, dt, rho, dx, dy):
        nx, ny = Pr.shape
        self.lib.ns_correct_V(_ptr(Vx), _ptr(Vy), _ptr(Pr), _D(dt), _D(rho), _D(dx), _D(dy), nx, ny)

    def set_bc_Pr(self, Pr, val, full_rows=False):
        nx, ny = Pr.shape
        self.lib.ns_set_bc_Pr(_ptr(Pr), _D(val), nx, ny, int(full_rows))

    def set_bc_Vel(self, Vx, Vy, Vprof, full_rows=False):
        nx, ny = Vy.shape[0], Vx.shape[1]
        self.lib.ns_set_bc_Vel(_ptr(Vx), _ptr(Vy), Vprof.ctypes.data_as(_P), nx, ny, int(full_rows))

    def advect(self, Vx, Vx_o, Vy, Vy_o, Cc, C_o, dt, dx, dy):
        nx, ny = Cc.shape
        self.lib.ns_advect(_ptr(Vx), _ptr(Vx_o), _ptr(Vy), _ptr(Vy_o), _ptr(Cc), _ptr(C_o), _D(dt), _D(dx), _D(dy), nx, ny)

    def set_sphere(self, Cc, Vx, Vy, a2, b2, ox, oy, sinb, cosb, lx, ly, dx, dy, iy0=0):
        nx, ny = Cc.shape
        self.lib.ns_set_sphere(_ptr(Cc), _ptr(Vx), _ptr(Vy), _D(a2), _D(b2), _D(ox), _D(oy), _D(sinb), _D(cosb),
                               _D(lx), _D(ly), _D(dx), _D(dy), nx, ny, int(iy0))

    def init_vprof(self, vin, lx, dx, nx) -> np.ndarray:
        v = np.zeros(nx)
        self.lib.ns_init_vprof(v.ctypes.data_as(_P), _D(vin), _D(lx), _D(dx), nx)
        return v

    # parameter-block entry points -------------------------------------------------------
    @staticmethod
    def params(p) -> OParams:
        return OParams(nx=p.nx, ny=p.ny, niter=p.niter, nchk=p.nchk, full_rows=int(p.full_rows),
                       lx=p.lx, ly=p.ly, rho=p.rho, vin=p.vin, psc=p.psc, mu=p.mu, g=p.g, a2=p.a2,
                       b2=p.b2, ox=p.ox, oy=p.oy, sinb=p.sinb, cosb=p.cosb, eps_it=p.eps_it,
                       dx=p.dx, dy=p.dy, dt=p.dt, damp=p.damp, dtau=p.dtau)

    def poisson(self, Pr, dP, Rp, divV, p, niter=None, nchk=None, cap=1024):
        op = self.params(p)
        eh = (C.c_double * cap)()
        ih = (C.c_int * cap)()
        nh = C.c_int(0)
        it = self.lib.ns_poisson(_ptr(Pr), _ptr(dP), _ptr(Rp), _ptr(divV), C.byref(op),
                                 p.niter if niter is None else niter, p.nchk if nchk is None else nchk,
                                 eh, ih, cap, C.byref(nh))
        n = min(nh.value, cap)
        return it, list(ih[:n]), list(eh[:n])


class OracleSim:
    """Host-side state of runme() (NavierStokes.jl:58-74) stepped by the oracle."""

    NAMES = ("Pr", "dPrdtau", "C", "C_o", "txx", "tyy", "txy", "Vx", "Vy", "Vx_o", "Vy_o", "divV", "Rp")

    def __init__(self, p, omp: bool = False):
        self.p, self.o = p, Oracle(omp)
        nx, ny = p.nx, p.ny
        shp = dict(Pr=(nx, ny), dPrdtau=(nx - 2, ny - 2), C=(nx, ny), C_o=(nx, ny), txx=(nx, ny),
                   tyy=(nx, ny), txy=(nx - 1, ny - 1), Vx=(nx + 1, ny), Vy=(nx, ny + 1),
                   Vx_o=(nx + 1, ny), Vy_o=(nx, ny + 1), divV=(nx, ny), Rp=(nx - 2, ny - 2))
        for k, s in shp.items():
            setattr(self, k, farray(*s))
        self.Vprof = self.o.init_vprof(p.vin, p.lx, p.dx, nx)
        self.Vy[:, 0] = self.Vprof
        self.err_evo, self.iter_evo = [], []

    def step(self, cap=1024) -> int:
        st = OState(**{k: getattr(self, k).ctypes.data for k in self.NAMES}, Vprof=self.Vprof.ctypes.data)
        op = Oracle.params(self.p)
        eh = (C.c_double * cap)()
        ih = (C.c_int * cap)()
        nh = C.c_int(0)
        it = self.o.lib.ns_step(C.byref(st), C.byref(op), eh, ih, cap, C.byref(nh))
        n = min(nh.value, cap)
        self.err_evo, self.iter_evo = list(eh[:n]), list(ih[:n])
        return it
