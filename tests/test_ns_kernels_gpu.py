"""Kernel-granular parity: every C-ABI entry point that replaces one @parallel call site of
scripts/NavierStokes.jl:79-100 against the CPU oracle on the same seeded inputs, bit for bit.
The reference holds no test or golden vector for these kernels (parity unpinned upstream); the
grids include the reference's default 153x255 and small/ragged shapes."""
import ctypes as C

import numpy as np
import pytest

from navierstokes_amd import ns_params
from util import Dev, rng_field, same_bits

pytestmark = pytest.mark.gpu

GRIDS = [(3, 3), (4, 7), (7, 5), (65, 9), (130, 67), (153, 255)]


def fields(rng, nx, ny):
    return dict(Vx=rng_field(rng, nx + 1, ny), Vy=rng_field(rng, nx, ny + 1), Pr=rng_field(rng, nx, ny),
                dP=rng_field(rng, nx - 2, ny - 2), divV=rng_field(rng, nx, ny), txx=rng_field(rng, nx, ny),
                tyy=rng_field(rng, nx, ny), txy=rng_field(rng, nx - 1, ny - 1), C=rng_field(rng, nx, ny))


@pytest.mark.parametrize("nx,ny", GRIDS)
def test_update_tau_predict_div(gpu, oracle, nx, ny):
    rng = np.random.default_rng(nx * 1000 + ny)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    f = fields(rng, nx, ny)
    d = {k: Dev(v) for k, v in f.items()}
    gpu.update_tau(d["txx"].ptr, d["tyy"].ptr, d["txy"].ptr, d["Vx"].ptr, d["Vy"].ptr, p.mu, p.dx, p.dy, nx, ny)
    oracle.update_tau(f["txx"], f["tyy"], f["txy"], f["Vx"], f["Vy"], p.mu, p.dx, p.dy)
    for k in ("txx", "tyy", "txy"):
        assert same_bits(d[k].get(), f[k]), k
    g = 0.37  # exercise the gravity term too (the reference runs g = 0)
    gpu.predict_V(d["Vx"].ptr, d["Vy"].ptr, d["txx"].ptr, d["tyy"].ptr, d["txy"].ptr, p.rho, g, p.dt, p.dx, p.dy, nx, ny)
    oracle.predict_V(f["Vx"], f["Vy"], f["txx"], f["tyy"], f["txy"], p.rho, g, p.dt, p.dx, p.dy)
    assert same_bits(d["Vx"].get(), f["Vx"]) and same_bits(d["Vy"].get(), f["Vy"])
    gpu.update_divV(d["divV"].ptr, d["Vx"].ptr, d["Vy"].ptr, p.dx, p.dy, nx, ny)
    oracle.update_divV(f["divV"], f["Vx"], f["Vy"], p.dx, p.dy)
    assert same_bits(d["divV"].get(), f["divV"])


@pytest.mark.parametrize("nx,ny", GRIDS)
def test_poisson_granular(gpu, oracle, nx, ny):
    rng = np.random.default_rng(7 + nx * 1000 + ny)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    f = fields(rng, nx, ny)
    d = {k: Dev(v) for k, v in f.items()}
    Rp = np.zeros((nx - 2, ny - 2), order="F")
    dR = Dev(Rp)
    for full in (0, 1):
        if full == 0 and nx > ny:
            continue  # bc_x! would index out of bounds in the reference
        for _ in range(3):
            gpu.update_dPrdtau(d["Pr"].ptr, d["dP"].ptr, d["divV"].ptr, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, nx, ny)
            gpu.update_Pr(d["Pr"].ptr, d["dP"].ptr, p.dtau, nx, ny)
            gpu.set_bc_Pr(d["Pr"].ptr, 0.0, nx, ny, full, 0, ny)
            oracle.update_dPrdtau(f["Pr"], f["dP"], f["divV"], p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
            oracle.update_Pr(f["Pr"], f["dP"], p.dtau)
            oracle.set_bc_Pr(f["Pr"], 0.0, bool(full))
        assert same_bits(d["Pr"].get(), f["Pr"]) and same_bits(d["dP"].get(), f["dP"])
    gpu.compute_res(dR.ptr, d["Pr"].ptr, d["divV"].ptr, p.rho, p.dt, p.dx, p.dy, nx, ny)
    oracle.compute_res(Rp, f["Pr"], f["divV"], p.rho, p.dt, p.dx, p.dy)
    assert same_bits(dR.get(), Rp)
    m = C.c_double(-1.0)
    gpu.maxabs(dR.ptr, Rp.size, C.byref(m))
    assert m.value == oracle.maxabs(Rp)


def test_maxabs_nan_and_empty(gpu, oracle):
    a = np.asfortranarray(np.random.default_rng(1).standard_normal((300, 17)))
    a[123, 5] = np.nan
    a[7, 7] = np.inf
    m = C.c_double(0.0)
    gpu.maxabs(Dev(a).ptr, a.size, C.byref(m))
    assert np.isnan(m.value) and np.isnan(oracle.maxabs(a))
    a[123, 5] = -3.0
    gpu.maxabs(Dev(a).ptr, a.size, C.byref(m))
    assert m.value == np.inf == oracle.maxabs(a)
    gpu.maxabs(Dev(a).ptr, 0, C.byref(m))
    assert m.value == 0.0


@pytest.mark.parametrize("nx,ny", GRIDS)
def test_correct_bc_sphere_advect(gpu, oracle, nx, ny):
    rng = np.random.default_rng(99 + nx * 1000 + ny)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    f = fields(rng, nx, ny)
    # velocities of a few cells per step so that the back-trace crosses cells and hits the clamps
    f["Vx"] *= 3.0
    f["Vy"] *= 3.0
    f["Vx"][2 % (nx + 1), 1 % ny] = 0.0          # exact zero displacement
    f["Vy"][1 % nx, 2 % (ny + 1)] = p.dy / p.dt  # integer displacement (t = 1 branch of the weight rule)
    d = {k: Dev(v) for k, v in f.items()}
    vprof = oracle.init_vprof(p.vin, p.lx, p.dx, nx)
    dvp = Dev(np.asfortranarray(vprof.reshape(nx, 1)))
    gpu.correct_V(d["Vx"].ptr, d["Vy"].ptr, d["Pr"].ptr, p.dt, p.rho, p.dx, p.dy, nx, ny)
    oracle.correct_V(f["Vx"], f["Vy"], f["Pr"], p.dt, p.rho, p.dx, p.dy)
    assert same_bits(d["Vx"].get(), f["Vx"]) and same_bits(d["Vy"].get(), f["Vy"])
    gpu.set_sphere(d["C"].ptr, d["Vx"].ptr, d["Vy"].ptr, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.dx, p.dy, nx, ny, 0)
    oracle.set_sphere(f["C"], f["Vx"], f["Vy"], p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.dx, p.dy)
    for k in ("C", "Vx", "Vy"):
        assert same_bits(d[k].get(), f[k]), k
    for full in (0, 1):
        if full == 0 and nx > ny + 1:
            continue
        gpu.set_bc_Vel(d["Vx"].ptr, d["Vy"].ptr, dvp.ptr, nx, ny, full, 0, ny)
        oracle.set_bc_Vel(f["Vx"], f["Vy"], vprof, bool(full))
        assert same_bits(d["Vx"].get(), f["Vx"]) and same_bits(d["Vy"].get(), f["Vy"])
    Vx_o, Vy_o, C_o = f["Vx"].copy(order="F"), f["Vy"].copy(order="F"), f["C"].copy(order="F")
    dxo, dyo, dco = Dev(Vx_o), Dev(Vy_o), Dev(C_o)
    gpu.advect(d["Vx"].ptr, dxo.ptr, d["Vy"].ptr, dyo.ptr, d["C"].ptr, dco.ptr, p.dt, p.dx, p.dy, nx, ny, 0, ny)
    oracle.advect(f["Vx"], Vx_o, f["Vy"], Vy_o, f["C"], C_o, p.dt, p.dx, p.dy)
    for k in ("C", "Vx", "Vy"):
        assert same_bits(d[k].get(), f[k]), k
