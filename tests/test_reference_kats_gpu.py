"""The known answers of the reference's own kernel tests, asserted directly on the HIP entry points
(no oracle in between).  Input recipes restated from
  tests/test_set_density.jl:41-76, tests/test_tangential_velocity.jl:66-91,
  tests/test_temp_uvel_gpu.jl:45-71 (and scripts/test_temp_vvel_gpu.jl for the v component),
  scripts/tests/test_move_front_gpu.jl:72-146, scripts/tests/test_advect_front_gpu.jl:78-147
in Float64 (the reference's tests use Float32 and compare its CPU loop with its CUDA kernel at
rtol 1e-5; here the closed form itself is the expected value)."""
import ctypes as C

import numpy as np
import pytest

from util import Dev

pytestmark = pytest.mark.gpu


def F(n1, n2, fill=0.0):
    return np.full((n1, n2), fill, order="F")


def vec(a):
    return Dev(np.asfortranarray(np.asarray(a, dtype=np.float64).reshape(-1, 1)))


def test_set_density_mask(gpu):
    """tests/test_set_density.jl:43-72: 256 x 256 array, x = y = range(0,1,256), centre (.5,.5), rad .2,
    rho2 = 2, m2 = .1 on a background of r = 1, m = .01: interior points inside the circle are set."""
    n = 256
    nx = ny = n - 2
    x = y = np.linspace(0.0, 1.0, n)
    r, m = Dev(F(n, n, 1.0)), Dev(F(n, n, 0.01))
    dx_, dy_ = vec(x), vec(y)
    gpu.bub_set_density(r.ptr, m.ptr, 0.5, 0.5, dx_.ptr, dy_.ptr, 0.2, 2.0, 0.1, nx, ny)
    mask = (x[:, None] - 0.5) ** 2 + (y[None, :] - 0.5) ** 2 < 0.2 ** 2
    mask[0, :] = mask[-1, :] = mask[:, 0] = mask[:, -1] = False
    assert np.array_equal(r.get(), np.where(mask, 2.0, 1.0))
    assert np.array_equal(m.get(), np.where(mask, 0.1, 0.01))
    assert abs(mask.sum() / n ** 2 - np.pi * 0.04) < 2e-3      # the disc's area


def test_tangential_velocity_ghosts(gpu):
    """tests/test_tangential_velocity.jl:66-91: 128 x 128, random interior, usouth/unorth = +-.1,
    vwest/veast = +-.2: every ghost value is 2*wall - inner, everything else is untouched."""
    nx = ny = 128
    rng = np.random.default_rng(0)
    u, v = np.asfortranarray(rng.random((nx + 1, ny + 2))), np.asfortranarray(rng.random((nx + 2, ny + 1)))
    du, dv = Dev(u), Dev(v)
    gpu.bub_set_tangential(du.ptr, dv.ptr, 0.1, -0.1, 0.2, -0.2, nx, ny)
    ur, vr = u.copy(), v.copy()
    ur[:, 0] = 2 * 0.1 - ur[:, 1]
    ur[:, -1] = 2 * -0.1 - ur[:, -2]
    vr[0, :] = 2 * 0.2 - vr[1, :]
    vr[-1, :] = 2 * -0.2 - vr[-2, :]
    assert np.array_equal(du.get(), ur) and np.array_equal(dv.get(), vr)


def test_temp_vel_adv_uniform_flow(gpu):
    """tests/test_temp_uvel_gpu.jl:47-65: nx = ny = 32, dt .01, dx = dy = .1, g = 9.81, rro = 1000, u = v = 1,
    r = 1000 (1100 and f = .1 inside (i-16)^2 + (j-16)^2 < 64).  In a uniform flow the advective terms
    cancel exactly, so ut = 1 + dt*(fx/rbar - (1 - rro/rbar)*gx) on i = 2..nx, j = 2..ny+1 and vt the same
    with fy, gy on i = 2..nx+1, j = 2..ny; nothing else is written."""
    nx = ny = 32
    dt, dx, dy, g, rro = 0.01, 0.1, 0.1, 9.81, 1000.0
    u, v = F(nx + 1, ny + 2, 1.0), F(nx + 2, ny + 1, 1.0)
    r, f = F(nx + 2, ny + 2, rro), F(nx + 2, ny + 2)
    ii, jj = np.meshgrid(np.arange(1, nx + 3), np.arange(1, ny + 3), indexing="ij")
    inside = (ii - 16) ** 2 + (jj - 16) ** 2 < 64
    r[inside] = 1100.0
    f[inside] = 0.1
    ut, vt = Dev(F(nx + 1, ny + 2)), Dev(F(nx + 2, ny + 1))
    du, dv, dr, df = Dev(u), Dev(v), Dev(r), Dev(f)
    gpu.bub_temp_vel_adv(ut.ptr, vt.ptr, du.ptr, dv.ptr, dr.ptr, dt, dx, dy, df.ptr, df.ptr, g, g, rro, nx, ny)
    got_u, got_v = ut.get(), vt.get()
    rbx = 0.5 * (r[2:nx + 1, 1:ny + 1] + r[1:nx, 1:ny + 1])
    want_u = 1.0 + dt * (-0.25 * 0.0 + f[1:nx, 1:ny + 1] / rbx - (1.0 - rro / rbx) * g)
    assert np.array_equal(got_u[1:nx, 1:ny + 1], want_u)
    rby = 0.5 * (r[1:nx + 1, 2:ny + 1] + r[1:nx + 1, 1:ny])
    want_v = 1.0 + dt * (-0.25 * 0.0 + f[1:nx + 1, 1:ny] / rby - (1.0 - rro / rby) * g)
    assert np.array_equal(got_v[1:nx + 1, 1:ny], want_v)
    for a, inner in ((got_u, (slice(1, nx), slice(1, ny + 1))), (got_v, (slice(1, nx + 1), slice(1, ny)))):
        rest = a.copy()
        rest[inner] = 0.0
        assert not rest.any()
    # outside the drop nothing acts on the fluid: the velocity stays 1 there
    assert (want_u[~inside[1:nx, 1:ny + 1] & ~inside[2:nx + 1, 1:ny + 1]] == 1.0).all()


def _circle_front(Nf, cx, cy, rad):
    ang = 2 * np.pi * np.arange(Nf) / (Nf - 1)
    xf, yf = np.zeros(Nf + 2), np.zeros(Nf + 2)
    xf[1:Nf + 1], yf[1:Nf + 1] = cx + rad * np.cos(ang), cy + rad * np.sin(ang)
    xf[0], yf[0], xf[Nf + 1], yf[Nf + 1] = xf[Nf], yf[Nf], xf[1], yf[1]
    return xf, yf


def test_move_front_rigid_rotation(gpu):
    """scripts/tests/test_move_front_gpu.jl:74-127: Nf = 1000 points on a circle of radius 3 about the
    centre of a 10 x 10 box, uf = -0.5 (y - L/2), vf = 0.5 (x - L/2), dt = .01: one step is x + dt*uf,
    y + dt*vf, the ghost points wrap and nobody leaves the box.  The entry point interpolates uf, vf from
    the grid first (advect_front! + move_front! in one call), so the rotation is given as grid velocities
    that are linear in x and y -- bilinear sampling returns them up to rounding."""
    Nf, dt, L = 1000, 0.01, 10.0
    nx = ny = 100
    dx = dy = L / nx
    xf, yf = _circle_front(Nf, L / 2, L / 2, 3.0)
    # u lives at (i dx, (j - .5) dy), v at ((i - .5) dx, j dy), i, j 0-based
    iu, ju = np.meshgrid(np.arange(nx + 1), np.arange(ny + 2), indexing="ij")
    iv, jv = np.meshgrid(np.arange(nx + 2), np.arange(ny + 1), indexing="ij")
    u = np.asfortranarray(-0.5 * ((ju - 0.5) * dy - L / 2))
    v = np.asfortranarray(0.5 * ((iv - 0.5) * dx - L / 2))
    dxf, dyf, duf, dvf = vec(xf), vec(yf), vec(np.zeros(Nf + 2)), vec(np.zeros(Nf + 2))
    du, dv = Dev(u), Dev(v)
    out = C.c_int(-1)
    gpu.bub_advect_move_front(duf.ptr, dvf.ptr, dxf.ptr, dyf.ptr, du.ptr, dv.ptr, dx, dy, dt, L, L, nx, ny, Nf, C.byref(out))
    assert out.value == 0
    gx, gy, guf, gvf = (a.get()[:, 0] for a in (dxf, dyf, duf, dvf))
    x0, y0 = xf[1:Nf + 1], yf[1:Nf + 1]
    assert np.allclose(guf[1:Nf + 1], -0.5 * (y0 - L / 2), rtol=0, atol=1e-14)
    assert np.allclose(gvf[1:Nf + 1], 0.5 * (x0 - L / 2), rtol=0, atol=1e-14)
    assert np.array_equal(gx[1:Nf + 1], x0 + dt * guf[1:Nf + 1]) and np.array_equal(gy[1:Nf + 1], y0 + dt * gvf[1:Nf + 1])
    assert (gx[0], gy[0], gx[Nf + 1], gy[Nf + 1]) == (gx[Nf], gy[Nf], gx[1], gy[1])
    rad = np.hypot(gx[1:Nf + 1] - L / 2, gy[1:Nf + 1] - L / 2)
    assert np.allclose(rad, 3.0 * np.sqrt(1 + (0.5 * dt) ** 2), rtol=1e-13)      # rigid rotation, explicit Euler


def test_advect_front_vortex(gpu):
    """scripts/tests/test_advect_front_gpu.jl:80-123: 128 x 128, dx = dy = .1, a vortex exp(-r/2) (0 inside
    r <= .1) about the grid's centre, Nf = 1000 front points on a circle of radius 3: uf, vf are the
    bilinear samples of u, v at the front (the four weighted corners summed left to right,
    scripts/code3_fun.jl:157-161,172-176).  dt = 0: the front itself stays where it is."""
    nx = ny = 128
    dx = dy = 0.1
    Nf = 1000
    cx, cy = (nx // 2) * dx, (ny // 2) * dy

    def vortex(n1, n2):
        ii, jj = np.meshgrid(np.arange(1, n1 + 1), np.arange(1, n2 + 1), indexing="ij")
        ex, ey = (ii - nx // 2) * dx, (jj - ny // 2) * dy
        r = np.hypot(ex, ey)
        th = np.arctan2(ey, ex)
        s = np.where(r > 0.1, np.exp(-r / 2.0), 0.0)
        return np.asfortranarray(-s * np.sin(th)), np.asfortranarray(s * np.cos(th))

    u, _ = vortex(nx + 1, ny + 2)
    _, v = vortex(nx + 2, ny + 1)
    xf, yf = _circle_front(Nf, cx, cy, 3.0)
    dxf, dyf, duf, dvf = vec(xf), vec(yf), vec(np.zeros(Nf + 2)), vec(np.zeros(Nf + 2))
    du, dv = Dev(u), Dev(v)
    gpu.bub_advect_move_front(duf.ptr, dvf.ptr, dxf.ptr, dyf.ptr, du.ptr, dv.ptr, dx, dy, 0.0, nx * dx, ny * dy, nx, ny, Nf, None)

    def sample(a, x, y):      # x, y already shifted to the field's own grid; 0-based floor
        ip, jp = np.floor(x / dx).astype(int), np.floor(y / dy).astype(int)
        ax, ay = x / dx - ip, y / dy - jp
        return ((1.0 - ax) * (1.0 - ay) * a[ip, jp] + ax * (1.0 - ay) * a[ip + 1, jp]
                + (1.0 - ax) * ay * a[ip, jp + 1] + ax * ay * a[ip + 1, jp + 1])

    x, y = xf[1:Nf + 1], yf[1:Nf + 1]
    guf, gvf = duf.get()[:, 0], dvf.get()[:, 0]
    assert np.array_equal(guf[1:Nf + 1], sample(u, x, y + 0.5 * dy))
    assert np.array_equal(gvf[1:Nf + 1], sample(v, x + 0.5 * dx, y))
    assert guf[0] == 0 and guf[Nf + 1] == 0
    assert np.array_equal(dxf.get()[:, 0], xf) and np.array_equal(dyf.get()[:, 0], yf)
    # the samples of a vortex are tangential to the circular front up to the half-cell offsets of the
    # staggered positions (the recipe defines both components on cell indices): |v . n| << |v| = exp(-1.5)
    radial = (guf[1:Nf + 1] * (x - cx) + gvf[1:Nf + 1] * (y - cy)) / 3.0
    speed = np.hypot(guf[1:Nf + 1], gvf[1:Nf + 1])
    assert np.abs(radial).max() < 0.02 and np.allclose(speed, np.exp(-1.5), rtol=0.1)
