"""DNS.jl (= code1.m) and code3_fun.jl kernels through the C ABI against the CPU oracle, bit for
bit: every kernel on seeded random fields (ragged grids included), the single-workgroup
anti-diagonal SOR against the lexicographic loop, the gather-form spreading against the
sequential scatter, and whole time steps at the reference sizes."""
import ctypes as C

import numpy as np
import pytest

from navierstokes_amd.bubble import Code3, Code3Params, DNS, DNSParams, cell_centres, front_circle
from oracle.bubble_oracle import BubbleOracle, OracleCode3, OracleDNS
from util import Dev, same_bits, strict_bits

pytestmark = pytest.mark.gpu
GRIDS = [(2, 2), (5, 3), (32, 32), (70, 45), (129, 200)]


@pytest.fixture(scope="module")
def bo():
    return BubbleOracle()


def rf(rng, n1, n2, lo=None):
    a = rng.standard_normal((n1, n2))
    if lo is not None:
        a = lo + rng.random((n1, n2))
    return np.asfortranarray(a)


def vec(a):
    return Dev(np.asfortranarray(a.reshape(-1, 1)))


@pytest.mark.parametrize("nx,ny", GRIDS)
def test_grid_kernels(gpu, bo, nx, ny):
    rng = np.random.default_rng(nx * 131 + ny)
    dt, dx, dy = 0.00125, 1.0 / nx, 1.3 / ny
    u, v = rf(rng, nx + 1, ny + 2), rf(rng, nx + 2, ny + 1)
    r, m = rf(rng, nx + 2, ny + 2, 1.0), rf(rng, nx + 2, ny + 2, 0.01)
    fx, fy, p = rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2)
    ut, vt = rf(rng, nx + 1, ny + 2), rf(rng, nx + 2, ny + 1)
    d = {k: Dev(a) for k, a in dict(u=u, v=v, r=r, m=m, fx=fx, fy=fy, p=p, ut=ut, vt=vt).items()}
    # set_tangential
    gpu.bub_set_tangential(d["u"].ptr, d["v"].ptr, 0.1, -0.1, 0.2, -0.2, nx, ny)
    bo.set_tangential(u, v, 0.1, -0.1, 0.2, -0.2, nx, ny)
    assert same_bits(d["u"].get(), u) and same_bits(d["v"].get(), v)
    # code1 predictor
    gpu.bub_temp_vel(d["ut"].ptr, d["vt"].ptr, d["u"].ptr, d["v"].ptr, d["r"].ptr, dt, dx, dy, 0.3, -100.0, 0.01, nx, ny)
    bo.temp_uvel(ut, u, v, r, dt, dx, dy, 0.3, 0.01, nx, ny)
    bo.temp_vvel(vt, u, v, r, dt, dx, dy, -100.0, 0.01, nx, ny)
    assert same_bits(d["ut"].get(), ut) and same_bits(d["vt"].get(), vt)
    # code3 predictor: advection then diffusion
    gpu.bub_temp_vel_adv(d["ut"].ptr, d["vt"].ptr, d["u"].ptr, d["v"].ptr, d["r"].ptr, dt, dx, dy, d["fx"].ptr, d["fy"].ptr, 0.3, -100.0, 1.0, nx, ny)
    bo.temp_uvel_adv(ut, u, v, r, dt, dx, dy, fx, 0.3, 1.0, nx, ny)
    bo.temp_vvel_adv(vt, u, v, r, dt, dx, dy, fy, -100.0, 1.0, nx, ny)
    assert same_bits(d["ut"].get(), ut) and same_bits(d["vt"].get(), vt)
    gpu.bub_temp_vel_diff(d["ut"].ptr, d["vt"].ptr, d["u"].ptr, d["v"].ptr, d["r"].ptr, dt, dx, dy, d["m"].ptr, nx, ny)
    bo.temp_uvel_diff(ut, u, v, r, dt, dx, dy, m, nx, ny)
    bo.temp_vvel_diff(vt, u, v, r, dt, dx, dy, m, nx, ny)
    assert same_bits(d["ut"].get(), ut) and same_bits(d["vt"].get(), vt)
    # rt, sources
    rt, t1, t2 = np.zeros_like(r), np.zeros_like(r), np.zeros_like(r)
    drt, dt1, dt2 = Dev(rt), Dev(t1), Dev(t2)
    gpu.bub_set_boundary(drt.ptr, d["r"].ptr, 1000.0, nx, ny)
    bo.set_boundary(rt, r, 1000.0, nx, ny)
    assert same_bits(drt.get(), rt)
    gpu.bub_compute_sources(dt1.ptr, dt2.ptr, drt.ptr, d["ut"].ptr, d["vt"].ptr, dt, dx, dy, nx, ny)
    bo.compute_sources(t1, t2, rt, ut, vt, dt, dx, dy, nx, ny)
    assert same_bits(dt1.get(), t1) and same_bits(dt2.get(), t2)
    # SOR: a few sweeps without convergence, then to convergence
    for maxit, tol in ((3, 0.0), (200, 1e-3)):
        sw, ch = C.c_int(0), C.c_double(0.0)
        gpu.bub_solve_pressure(d["p"].ptr, dt1.ptr, dt2.ptr, drt.ptr, 1.2, dx, dy, nx, ny, maxit, tol, C.byref(sw), C.byref(ch))
        chr_ = C.c_double(0.0)
        it = bo.lib.bub_solve_pressure(p.ctypes.data_as(C.c_void_p), t1.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p),
                                       rt.ctypes.data_as(C.c_void_p), C.c_double(1.2), C.c_double(dx), C.c_double(dy), nx, ny, maxit,
                                       C.c_double(tol), C.byref(chr_))
        assert sw.value == it and ch.value == chr_.value
        assert same_bits(d["p"].get(), p)
    # corrector, density advection, corner velocities, viscosity
    gpu.bub_correct_vel(d["u"].ptr, d["v"].ptr, d["ut"].ptr, d["vt"].ptr, d["p"].ptr, d["r"].ptr, dt, dx, dy, nx, ny)
    bo.correct_uvel(u, ut, p, r, dt, dx, nx, ny)
    bo.correct_vvel(v, vt, p, r, dt, dy, nx, ny)
    assert same_bits(d["u"].get(), u) and same_bits(d["v"].get(), v)
    ro = r.copy(order="F")
    dro = Dev(ro)
    gpu.bub_advect_density(d["r"].ptr, dro.ptr, d["u"].ptr, d["v"].ptr, dt, dx, dy, 0.01, nx, ny)
    bo.advect_density(r, ro, u, v, dt, dx, dy, 0.01, nx, ny)
    assert same_bits(d["r"].get(), r)
    uu, vv = np.zeros((nx + 1, ny + 1), order="F"), np.zeros((nx + 1, ny + 1), order="F")
    duu, dvv = Dev(uu), Dev(vv)
    gpu.bub_velocity(duu.ptr, dvv.ptr, d["u"].ptr, d["v"].ptr, nx, ny)
    bo.velocity(uu, vv, u, v, nx, ny)
    assert same_bits(duu.get(), uu) and same_bits(dvv.get(), vv)
    gpu.bub_update_visco(d["r"].ptr, d["m"].ptr, 0.01, 0.05, 1.0, 2.0, nx, ny)
    bo.update_visco(r, m, 0.01, 0.05, 1.0, 2.0, nx, ny)
    assert same_bits(d["m"].get(), m)
    # density SOR
    sw = C.c_int(0)
    gpu.bub_construct_den(d["r"].ptr, d["fx"].ptr, d["fy"].ptr, 1.2, 50, 1e-3, dx, dy, nx, ny, C.byref(sw))
    it = bo.construct_den(r, fx, fy, 1.2, 50, 1e-3, dx, dy, nx, ny)
    assert sw.value == it and same_bits(d["r"].get(), r)
    gpu.bub_average(d["r"].ptr, dro.ptr, r.size)
    assert same_bits(d["r"].get(), 0.5 * (r + ro))


@pytest.mark.parametrize("nx,ny,rows,copies", [(70, 45, 0, 0), (129, 200, 0, 0), (129, 200, 8, 0), (200, 129, 16, 3),
                                                (256, 256, 0, 0), (256, 256, 4, 5), (64, 300, 64, 4), (65, 33, 32, 0)])
def test_sor_on_the_whole_chip(gpu, bo, nx, ny, rows, copies):
    """Lexicographic SOR on grids beyond one workgroup's LDS: tiles of 64 x `rows` cells, a wavefront per
    (tile, sweep), several sweeps in flight across the chip.  Pressure and density solves: sweep count,
    last change and values (strict bit patterns) equal the serial loop of the oracle for stopping points
    from the first sweep to the sweep limit, with more and with fewer copies of the unknown than sweeps
    fit in flight, and equal the single-workgroup kernel."""
    rng = np.random.default_rng(nx * 7 + ny + rows)
    dx, dy = 1.0 / nx, 1.3 / ny
    rt = rf(rng, nx + 2, ny + 2, 1.0)
    t1 = rf(rng, nx + 2, ny + 2) * 30.0
    ut, vt = rf(rng, nx + 1, ny + 2), rf(rng, nx + 2, ny + 1)
    t1d, t2 = np.zeros_like(rt), np.zeros_like(rt)
    bo.compute_sources(t1d, t2, rt, ut, vt, 0.00125, dx, dy, nx, ny)
    p0 = rf(rng, nx + 2, ny + 2)
    fx, fy, r0 = rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2, 1.0)
    gpu.set_option("sor_grid_rows", rows)
    gpu.set_option("sor_grid_copies", copies)
    drt, dt1, dt2, dfx, dfy = Dev(rt), Dev(t1), Dev(t2), Dev(fx), Dev(fy)
    try:
        for maxit, tol in ((1, 0.0), (2, 0.0), (7, 0.0), (40, 0.0), (400, 5e-2), (400, 1e-3)):
            p = p0.copy(order="F")
            chr_ = C.c_double(0.0)
            it = bo.lib.bub_solve_pressure(p.ctypes.data_as(C.c_void_p), t1.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p),
                                           rt.ctypes.data_as(C.c_void_p), C.c_double(1.2), C.c_double(dx), C.c_double(dy), nx, ny, maxit,
                                           C.c_double(tol), C.byref(chr_))
            for grid in (1, 0):
                if grid == 0 and (maxit > 7 or nx * ny > 30000):
                    continue                      # the single-workgroup kernel takes milliseconds per sweep
                gpu.set_option("sor_grid", grid)
                dp = Dev(p0)
                sw, ch = C.c_int(0), C.c_double(0.0)
                gpu.bub_solve_pressure(dp.ptr, dt1.ptr, dt2.ptr, drt.ptr, 1.2, dx, dy, nx, ny, maxit, tol, C.byref(sw), C.byref(ch))
                assert (sw.value, ch.value) == (it, chr_.value), (grid, maxit, tol)
                assert strict_bits(dp.get(), p), (grid, maxit, tol)
            gpu.set_option("sor_grid", 1)
            r = r0.copy(order="F")
            it = bo.construct_den(r, fx, fy, 1.2, maxit, tol, dx, dy, nx, ny)
            dr = Dev(r0)
            sw = C.c_int(0)
            gpu.bub_construct_den(dr.ptr, dfx.ptr, dfy.ptr, 1.2, maxit, tol, dx, dy, nx, ny, C.byref(sw))
            assert sw.value == it, (maxit, tol)
            assert strict_bits(dr.get(), r), (maxit, tol)
    finally:
        gpu.set_option("sor_grid", 1)
        gpu.set_option("sor_grid_rows", 0)
        gpu.set_option("sor_grid_copies", 0)


def test_set_density(gpu, bo):
    nx, ny = 61, 40
    x, y = cell_centres(nx, 1.0 / nx), cell_centres(ny, 1.0 / ny)
    r, m = np.full((nx + 2, ny + 2), 1.0, order="F"), np.full((nx + 2, ny + 2), 0.01, order="F")
    dr, dm, dx_, dy_ = Dev(r), Dev(m), vec(x), vec(y)     # keep every device array referenced
    gpu.bub_set_density(dr.ptr, dm.ptr, 0.5, 0.7, dx_.ptr, dy_.ptr, 0.15, 2.0, 0.05, nx, ny)
    bo.set_density(r, m, 0.5, 0.7, x, y, 0.15, 2.0, 0.05, nx, ny)
    assert same_bits(dr.get(), r) and same_bits(dm.get(), m) and (r == 2.0).sum() > 10


@pytest.mark.parametrize("nx,ny,Nf", [(32, 32, 100), (48, 20, 37), (16, 64, 400), (64, 64, 2600)])
def test_front_kernels(gpu, bo, nx, ny, Nf):
    q = Code3Params(nx=nx, ny=ny, Nf=Nf)
    dx, dy = q.dx, q.dy
    rng = np.random.default_rng(Nf)
    xf, yf = front_circle(q)
    xf[1:Nf + 1] += rng.standard_normal(Nf) * 0.004
    yf[1:Nf + 1] += rng.standard_normal(Nf) * 0.004
    xf[0], yf[0], xf[Nf + 1], yf[Nf + 1] = xf[Nf], yf[Nf], xf[1], yf[1]
    cap = Nf * 3
    pad = lambda a: np.concatenate([a, np.zeros(cap - a.size)])
    dxf, dyf = vec(pad(xf)), vec(pad(yf))
    tx, ty = np.zeros(cap), np.zeros(cap)
    dtx, dty = vec(tx), vec(ty)
    fx, fy = np.zeros((nx + 2, ny + 2), order="F"), np.zeros((nx + 2, ny + 2), order="F")
    dfx, dfy = Dev(fx + 7.0), Dev(fy - 7.0)       # must be overwritten, not accumulated into
    gpu.bub_surface_tension(dfx.ptr, dfy.ptr, dtx.ptr, dty.ptr, dxf.ptr, dyf.ptr, 10.0, dx, dy, nx, ny, Nf)
    bo.front_tangents(tx, ty, xf, yf, Nf)
    bo.surface_tension(fx, fy, xf, yf, tx, ty, 10.0, dx, dy, nx, ny, Nf)
    assert same_bits(dtx.get()[: Nf + 2, 0], tx[: Nf + 2]) and same_bits(dty.get()[: Nf + 2, 0], ty[: Nf + 2])
    assert same_bits(dfx.get(), fx) and same_bits(dfy.get(), fy) and np.abs(fx).max() > 0
    gpu.bub_distribute(dfx.ptr, dfy.ptr, dxf.ptr, dyf.ptr, 1.0, 2.0, dx, dy, nx, ny, Nf)
    bo.distribute(fx, fy, xf, yf, 1.0, 2.0, dx, dy, nx, ny, Nf)
    assert same_bits(dfx.get(), fx) and same_bits(dfy.get(), fy)
    # advect + move
    u, v = rf(rng, nx + 1, ny + 2), rf(rng, nx + 2, ny + 1)
    uf, vf = np.zeros(cap), np.zeros(cap)
    duf, dvf = vec(uf), vec(vf)
    cnt = C.c_int(-1)
    du, dv = Dev(u), Dev(v)
    gpu.bub_advect_move_front(duf.ptr, dvf.ptr, dxf.ptr, dyf.ptr, du.ptr, dv.ptr, dx, dy, 0.00125, 1.0, 1.0, nx, ny, Nf, C.byref(cnt))
    bo.advect_front(uf, vf, xf, yf, u, v, dx, dy, nx, Nf)
    out = bo.move_front(xf, yf, uf, vf, Nf, 0.00125, 1.0, 1.0)
    assert cnt.value == out
    assert same_bits(duf.get()[: Nf + 2, 0], uf[: Nf + 2]) and same_bits(dvf.get()[: Nf + 2, 0], vf[: Nf + 2])
    assert same_bits(dxf.get()[: Nf + 2, 0], xf) and same_bits(dyf.get()[: Nf + 2, 0], yf)
    # regrid: stretch one half of the front so that points are added and removed
    xf[1:Nf // 2] = 0.5 + (xf[1:Nf // 2] - 0.5) * 1.6
    xf[0], xf[Nf + 1] = xf[Nf], xf[1]
    dxf, dyf = vec(pad(xf)), vec(pad(yf))
    nf = C.c_int(0)
    sx, sy = vec(np.zeros(cap)), vec(np.zeros(cap))
    gpu.bub_regrid_front(dxf.ptr, dyf.ptr, sx.ptr, sy.ptr, dx, dy, Nf, cap, C.byref(nf))
    xo, yo = pad(xf), pad(yf)
    n2 = bo.regrid_front(xo, yo, dx, dy, Nf, cap)
    assert nf.value == n2 and n2 != Nf
    assert same_bits(dxf.get()[: n2 + 2, 0], xo[: n2 + 2]) and same_bits(dyf.get()[: n2 + 2, 0], yo[: n2 + 2])


STEP_FORMS = ["merged", "step", "kernels"]


def step_form(g, form):
    """merged: one C call per time step, element-wise kernels as phases of the solver launches;
    step: one C call per time step, a launch per kernel; kernels: one C call per kernel."""
    g.fused = form != "kernels"
    g.ctx.set_option("bub_merge_launches", int(form == "merged"))


@pytest.mark.parametrize("form", STEP_FORMS)
def test_dns_steps(form):
    """scripts/DNS.jl at its sizes (32 x 32, dt .00125, beta 1.2, maxit 400): 25 steps, as one launch
    per time step, one C call per time step (nsk_bub_dns_step) and one call per kernel."""
    q = DNSParams()
    g, o = DNS(q), OracleDNS(q)
    step_form(g, form)
    try:
        for _ in range(25):
            assert g.step() == o.step()
    finally:
        g.ctx.set_option("bub_merge_launches", 1)
    for k in ("rho", "u", "v", "p", "uu", "vv"):
        assert same_bits(getattr(g, k).get(), getattr(o, k)), k


@pytest.mark.parametrize("nx,ny", GRIDS + [(300, 257)])
def test_scalable_pressure_solvers(gpu, bo, nx, ny):
    """Red/black SOR (code3_perf_gpu.jl:129-183) and the pseudo-transient iteration
    (code1_perf_gpu.jl:98-118): a fixed number of iterations, then to convergence; iteration
    count, last change and fields equal the oracle's bit for bit."""
    rng = np.random.default_rng(nx * 17 + ny)
    dx, dy = 1.0 / nx, 1.3 / ny
    p, t1 = rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2) * 10.0
    rt = rf(rng, nx + 2, ny + 2, 1.0)
    t2 = np.zeros_like(p)
    bo.set_boundary(rt, rt.copy(order="F"), 1000.0, nx, ny)
    bo.compute_sources(np.zeros_like(p), t2, rt, rf(rng, nx + 1, ny + 2), rf(rng, nx + 2, ny + 1), 1e-3, dx, dy, nx, ny)
    dt1, dt2, drt = Dev(t1), Dev(t2), Dev(rt)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    D = C.c_double
    # red/black: 3 and 11 sweeps (odd: crosses a batch boundary), then to a tolerance
    prb = p.copy(order="F")
    dprb = Dev(prb)
    for maxit, tol in ((3, 0.0), (11, 0.0), (400, 1e-4 * np.abs(t1).max() * dx * dx)):
        sw, ch, chr_ = C.c_int(0), D(0.0), D(0.0)
        gpu.bub_solve_pressure_rb(dprb.ptr, dt1.ptr, dt2.ptr, drt.ptr, 1.2, dx, dy, nx, ny, maxit, tol, C.byref(sw), C.byref(ch))
        it = bo.lib.bub_solve_pressure_rb(vp(prb), vp(t1), vp(t2), vp(rt), D(1.2), D(dx), D(dy), nx, ny, maxit, D(tol), C.byref(chr_))
        assert sw.value == it and ch.value == chr_.value, (maxit, sw.value, it, ch.value, chr_.value)
        assert same_bits(dprb.get(), prb)
    # pseudo-transient
    ppt, dP = p.copy(order="F"), np.zeros_like(p)
    dppt, ddP = Dev(ppt), Dev(dP)
    dtau = 0.9 / np.sqrt(2.0) * min(dx, dy)
    for maxit, tol in ((5, 0.0), (9, 0.0), (300, 1e-3 * np.abs(t1).max() * dtau * dtau)):
        sw, ch, chr_ = C.c_int(0), D(0.0), D(0.0)
        gpu.bub_pt_pressure(dppt.ptr, ddP.ptr, dt1.ptr, drt.ptr, dtau, 0.72, dx, dy, nx, ny, maxit, tol, C.byref(sw), C.byref(ch))
        it = bo.lib.bub_pt_pressure(vp(ppt), vp(dP), vp(t1), vp(rt), D(dtau), D(0.72), D(dx), D(dy), nx, ny, maxit, D(tol), C.byref(chr_))
        assert sw.value == it and ch.value == chr_.value, (maxit, sw.value, it, ch.value, chr_.value)
        assert same_bits(dppt.get(), ppt) and same_bits(ddP.get(), dP)


@pytest.mark.parametrize("nx,ny", [(2, 2), (3, 7), (31, 33), (32, 32), (33, 20), (64, 64), (66, 60), (70, 45)])
def test_sor_sweeps_in_flight(gpu, bo, nx, ny):
    """The lexicographic SOR with several sweeps in flight (grids that fit LDS; with the quotients
    through reciprocals and with IEEE divisions) and with one sweep at a time give the oracle's
    values, sweep count and last change: every stopping point from the first sweep to one past a
    full round of groups, tolerance met early, late and never, and the operands the reciprocal
    form hands back to the IEEE division: NaN, infinities, zeros of either sign, subnormal and
    huge pressures, density sums outside [2^-100, 2^100] and of the wrong sign."""
    rng = np.random.default_rng(nx * 977 + ny)
    dx, dy = 1.0 / nx, 1.3 / ny
    rt, t2 = rf(rng, nx + 2, ny + 2, 1.0), rf(rng, nx + 2, ny + 2, 0.01)
    t1, fx, fy = rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2), rf(rng, nx + 2, ny + 2)
    t2 *= dx * dx
    p0 = rf(rng, nx + 2, ny + 2)

    def variant(fill, count=3):
        a = p0.copy(order="F")
        for v in np.atleast_1d(fill):
            for _ in range(count):
                a[rng.integers(0, nx + 2), rng.integers(0, ny + 2)] = v
        return a

    pnan = variant(np.nan, 1)
    pzero = np.asfortranarray(np.where(rng.random(p0.shape) < 0.5, 0.0, -0.0))
    pzero[nx // 2 + 1, ny // 2 + 1] = 1.0
    specials = [pnan, variant(np.inf, 1), pzero, variant([0.0, -0.0]), variant([3e-310, -5e-324]), variant([1e290, -1e305]),
                variant([1e-280, -1e-300])]
    rt_tiny, rt_huge, rt_neg = rt * 1e-45, rt * 1e40, rt.copy(order="F")
    rt_neg[(nx + 2) // 2, (ny + 2) // 2] = -3.0
    drt, dt1, dt2, dfx, dfy = Dev(rt), Dev(t1), Dev(t2), Dev(fx), Dev(fy)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    cases = [(m, 0.0, p0, rt) for m in (1, 2, 3, 4, 15, 16, 17, 31, 32, 33, 34, 65, 97)]
    cases += [(400, 1e-2, p0, rt), (400, 1e-5, p0, rt), (40, 1e-300, p0, rt)]
    cases += [(37, 1e-3, start, rt) for start in specials]
    cases += [(23, 1e-3, p0, other) for other in (rt_tiny, rt_huge, rt_neg)]
    try:
        for pipe, fast in ((1, 1), (1, 0), (0, 1)):
            gpu.set_option("sor_pipeline", pipe)
            gpu.set_option("sor_fast_division", fast)
            for n, (maxit, tol, start, dens) in enumerate(cases):
                what = (pipe, fast, n, maxit, tol)
                ddens = drt if dens is rt else Dev(dens)
                p, dp_ = start.copy(order="F"), Dev(start)
                sw, ch, chr_ = C.c_int(0), C.c_double(0.0), C.c_double(0.0)
                gpu.bub_solve_pressure(dp_.ptr, dt1.ptr, dt2.ptr, ddens.ptr, 1.2, dx, dy, nx, ny, maxit, tol, C.byref(sw), C.byref(ch))
                it = bo.lib.bub_solve_pressure(vp(p), vp(t1), vp(t2), vp(dens), C.c_double(1.2), C.c_double(dx), C.c_double(dy), nx, ny, maxit,
                                               C.c_double(tol), C.byref(chr_))
                assert sw.value == it, (what, sw.value, it)
                assert strict_bits(np.array([ch.value]), np.array([chr_.value])), what
                assert strict_bits(dp_.get(), p), what
                if dens is not rt or fast == 0:
                    continue
                r, dr = start.copy(order="F"), Dev(start)
                gpu.bub_construct_den(dr.ptr, dfx.ptr, dfy.ptr, 1.2, maxit, tol, dx, dy, nx, ny, C.byref(sw))
                it = bo.construct_den(r, fx, fy, 1.2, maxit, tol, dx, dy, nx, ny)
                assert sw.value == it and same_bits(dr.get(), r), (what, sw.value, it)
    finally:
        gpu.set_option("sor_pipeline", 1)
        gpu.set_option("sor_fast_division", 1)


def test_solver_nan_runs_to_maxit(gpu, bo):
    """A NaN change never satisfies `< maxError`: both solvers run all maxit iterations."""
    nx = ny = 6
    p = np.asfortranarray(np.zeros((nx + 2, ny + 2)))
    p[3, 3] = np.nan
    one = np.asfortranarray(np.ones((nx + 2, ny + 2)))
    dp_, d1 = Dev(p), Dev(one)
    sw, ch = C.c_int(0), C.c_double(0.0)
    gpu.bub_solve_pressure_rb(dp_.ptr, d1.ptr, d1.ptr, d1.ptr, 1.2, 0.1, 0.1, nx, ny, 19, 1e-3, C.byref(sw), C.byref(ch))
    assert sw.value == 19 and np.isnan(ch.value)
    ddP = Dev(np.asfortranarray(np.zeros((nx + 2, ny + 2))))
    gpu.bub_pt_pressure(dp_.ptr, ddP.ptr, d1.ptr, d1.ptr, 0.01, 0.72, 0.1, 0.1, nx, ny, 10, 1e-3, C.byref(sw), C.byref(ch))
    assert sw.value == 10 and np.isnan(ch.value)


@pytest.mark.parametrize("solver", ["redblack", "pt"])
def test_dns_steps_other_solvers(solver):
    """The drop problem with the scalable pressure solvers: 6 steps, bit-identical to the oracle
    run with the same solver."""
    q = DNSParams(pressure_solver=solver, maxit=200 if solver == "redblack" else 3000)
    g, o = DNS(q), OracleDNS(q)
    for _ in range(6):
        assert g.step() == o.step()
    for k in ("rho", "u", "v", "p", "uu", "vv"):
        assert same_bits(getattr(g, k).get(), getattr(o, k)), k


@pytest.mark.parametrize("preset", ["dns_jl", "dns_loop_jl"])
def test_dns_presets_of_the_two_scripts(preset):
    """scripts/DNS.jl as written (maxit 400, cell centres from LinRange(-0.5dx, Lx+0.5dx, nx+2), DNS.jl:46-47) and
    scripts/DNS_loop.jl (maxit 200, dt from its two CFL numbers, DNS_loop.jl:43-52): 12 steps against the oracle
    run with the same parameter block."""
    q = getattr(DNSParams, preset)()
    assert q.grid == "linrange" and q.maxit == (400 if preset == "dns_jl" else 200)
    if preset == "dns_loop_jl":
        assert q.dt == min((1 / 4.1) * (q.dy * q.dy) * q.rho1 / q.m0, 1.0 * q.dy / 1.0) and q.dt != 0.00125
    g, o = DNS(q), OracleDNS(q)
    xc, _ = q.centres()
    assert xc[0] == -0.5 * q.dx and xc[-1] == q.Lx + 0.5 * q.dx and same_bits(g.xc.get(), xc)
    assert same_bits(g.rho.get(), o.rho)
    for _ in range(12):
        assert g.step() == o.step()
    for k in ("rho", "u", "v", "p", "uu", "vv"):
        assert same_bits(getattr(g, k).get(), getattr(o, k)), k


@pytest.mark.parametrize("form", STEP_FORMS)
@pytest.mark.parametrize("nx,ny,Nf", [(32, 32, 100), (21, 40, 57)])
def test_code3_steps(form, nx, ny, Nf):
    """scripts/code3_fun.jl at its sizes (32 x 32, Nf = 100) and on an uneven grid: 12 steps incl. front
    regridding, with the element-wise kernels merged into the solver launches, as one C call per time
    step (nsk_bub_code3_step) and as one call per kernel."""
    q = Code3Params(nx=nx, ny=ny, Nf=Nf)
    g, o = Code3(q), OracleCode3(q)
    step_form(g, form)
    try:
        for _ in range(12):
            assert g.step() == o.step()
            assert g.Nf == o.Nf
    finally:
        g.ctx.set_option("bub_merge_launches", 1)
    for k in ("r", "m", "u", "v", "p"):
        assert same_bits(getattr(g, k).get(), getattr(o, k)), k
    gx, gy = g.front()
    ox, oy = o.front()
    assert same_bits(gx, ox) and same_bits(gy, oy)


@pytest.mark.parametrize("merged", [1, 0])
def test_runs_of_steps(merged):
    """The time loops in one C call each (nsk_bub_dns_run, nsk_bub_code3_run): the sweep counts of every
    step, the front sizes and the final fields are those of the oracle's step-by-step loop; a run
    continues where single steps left off and the other way round."""
    q = DNSParams()
    g, o = DNS(q), OracleDNS(q)
    g.ctx.set_option("bub_merge_launches", merged)
    try:
        assert g.run(0) == []
        assert g.run(13) == [o.step() for _ in range(13)]
        assert g.step() == o.step()
        assert g.run(6) == [o.step() for _ in range(6)]
        for k in ("rho", "u", "v", "p", "uu", "vv"):
            assert same_bits(getattr(g, k).get(), getattr(o, k)), k
        assert abs(g.time - 20 * q.dt) < 1e-12
        q3 = Code3Params()
        g3, o3 = Code3(q3), OracleCode3(q3)
        assert g3.run(7) == [o3.step() for _ in range(7)] and g3.Nf == o3.Nf
        assert g3.step() == o3.step() and g3.Nf == o3.Nf
        assert g3.run(5) == [o3.step() for _ in range(5)] and g3.Nf == o3.Nf
        for k in ("r", "m", "u", "v", "p"):
            assert same_bits(getattr(g3, k).get(), getattr(o3, k)), k
        (gx, gy), (ox, oy) = g3.front(), o3.front()
        assert same_bits(gx, ox) and same_bits(gy, oy)
        # a front that outgrows its vectors in the middle of a run gets larger ones, as the reference's
        # resize! does (code3_fun.jl:462-465), and the run goes on from the step that needed them
        from dataclasses import replace
        q4 = Code3Params(nx=48, ny=48, Nf=40, cap=48)
        g4, o4 = Code3(q4), OracleCode3(replace(q4, cap=4096))
        assert g4.run(6) == [o4.step() for _ in range(6)] and g4.Nf == o4.Nf
        assert g4.grown >= 1 and g4.cap > 48 and g4.Nf + 2 > 48
        for k in ("r", "m", "u", "v", "p"):
            assert same_bits(getattr(g4, k).get(), getattr(o4, k)), k
        (gx, gy), (ox, oy) = g4.front(), o4.front()
        assert same_bits(gx, ox) and same_bits(gy, oy)
    finally:
        g.ctx.set_option("bub_merge_launches", 1)


def test_fused_steps_report_their_errors(gpu):
    """Null arrays, a front that does not fit its vectors and a front that outgrows them are errors of
    the whole-step entry points, not faults."""
    from navierstokes_amd import _lib
    with pytest.raises(_lib.NskError, match="null"):
        gpu.bub_dns_step(C.byref(_lib.CDnsState()), C.byref(_lib.CBubParams(nx=8, ny=8, maxit=3)), None)
    q = Code3Params(nx=16, ny=16, Nf=60, cap=64)
    g = Code3(q)
    g.Nf = 63
    with pytest.raises(_lib.NskError, match="front of 63 points in vectors of 64"):
        g.step()
    # coarse front on a fine grid: regridding doubles it.  The C entry points report it with a code of
    # their own and leave the step complete but for the regridding ...
    g = Code3(Code3Params(nx=48, ny=48, Nf=40, cap=48))
    st, pr = g._c_args()
    with pytest.raises(_lib.NskError, match="outgrew its capacity") as e:
        gpu_steps = 0
        for _ in range(3):
            g.ctx.bub_code3_step(C.byref(st), C.byref(pr), None)
            gpu_steps += 1
    assert e.value.code == _lib.ERR_FRONT_CAPACITY and g._steps_done() == 0 and st.Nf + 2 <= 48
    st, pr = Code3(Code3Params(nx=48, ny=48, Nf=40, cap=48))._c_args()
    with pytest.raises(_lib.NskError, match="outgrew its capacity") as e:
        g.ctx.bub_code3_run(C.byref(st), C.byref(pr), 4, None)
    assert e.value.code == _lib.ERR_FRONT_CAPACITY


@pytest.mark.parametrize("form", STEP_FORMS)
def test_code3_front_grows_like_the_reference(form):
    """code3_fun.jl:462-465 resizes xf, yf when the regridding runs out of room.  The host driver does the
    same for every form of the step: the fields and the front equal the oracle's, which is given room
    from the start."""
    from dataclasses import replace
    q = Code3Params(nx=48, ny=48, Nf=40, cap=48)
    g, o = Code3(q), OracleCode3(replace(q, cap=4096))
    step_form(g, form)
    try:
        for _ in range(5):
            assert g.step() == o.step()
            assert g.Nf == o.Nf
    finally:
        g.ctx.set_option("bub_merge_launches", 1)
    assert g.grown >= 1
    for k in ("r", "m", "u", "v", "p"):
        assert same_bits(getattr(g, k).get(), getattr(o, k)), k
    (gx, gy), (ox, oy) = g.front(), o.front()
    assert same_bits(gx, ox) and same_bits(gy, oy)


@pytest.mark.parametrize("form", STEP_FORMS)
def test_code3_front_leaving_the_box_is_an_error(form):
    """A front point that leaves [0,Lx] x [0,Ly]: the reference prints a message (code3_fun.jl:186-188) and
    its next array access by that point throws; the oracle stops with FrontLeftTheBox; the step and the
    run entry points return NSK_ERR_FRONT_OUTSIDE in the same time step."""
    from navierstokes_amd import _lib
    from oracle.bubble_oracle import FrontLeftTheBox
    # one pressure sweep per solve leaves the velocity far from divergence-free: the drop is torn out of
    # the cavity in the third time step
    q = Code3Params(nx=16, ny=24, gy=-30.0, gx=5.0, rho2=3.82, veast=0.3, vwest=-0.3, rad=0.169, maxit=1, beta=1.34,
                    xc=0.555, yc=0.562, Nf=57, sigma=0.0, dt0=0.0022222222222222222)
    o, n_ok = OracleCode3(q), 0
    with pytest.raises(FrontLeftTheBox):
        for _ in range(8):
            o.step()
            n_ok += 1
    assert n_ok == 2
    g = Code3(q)
    step_form(g, form)
    try:
        for _ in range(n_ok):
            g.step()
        with pytest.raises(_lib.NskError, match="left the box") as e:
            g.step()
        assert e.value.code == _lib.ERR_FRONT_OUTSIDE
        if g.fused:
            g2 = Code3(q)
            step_form(g2, form)
            with pytest.raises(_lib.NskError, match=f"left the box.* in step {n_ok + 1} of the run|left the box") as e:
                g2.run(n_ok + 3)
            assert e.value.code == _lib.ERR_FRONT_OUTSIDE and g2._steps_done() == n_ok
    finally:
        g.ctx.set_option("bub_merge_launches", 1)
