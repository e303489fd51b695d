"""The C ABI the way a foreign-language driver (julia/NSK.jl) uses it: the library owns device
memory (nsk_zeros / nsk_upload / nsk_download / nsk_copy / nsk_free), runs on its own stream, and
reports every misuse through a nonzero return code and nsk_last_error() -- never by computing
something else.  Results against the oracle, bit for bit."""
import ctypes as C

import numpy as np
import pytest

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import c_params, init_vprof
from oracle.ns_oracle import OracleSim
from util import same_bits

pytestmark = pytest.mark.gpu


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_library_owned_memory_whole_steps():
    p = ns_params("NavierStokes", nx=51, ny=83)
    nx, ny = p.nx, p.ny
    ctx = _lib.Context(0)                 # no stream adopted: the context's own stream
    ctx.use_own_stream()
    shapes = dict(Pr=(nx, ny), dPrdtau=(nx - 2, ny - 2), C=(nx, ny), C_o=(nx, ny), txx=(nx, ny), tyy=(nx, ny),
                  txy=(nx - 1, ny - 1), Vx=(nx + 1, ny), Vy=(nx, ny + 1), Vx_o=(nx + 1, ny), Vy_o=(nx, ny + 1),
                  divV=(nx, ny), Rp=(nx - 2, ny - 2), Vprof=(nx, 1))
    h = {}
    try:
        for k, (a, b) in shapes.items():
            ptr = C.c_void_p()
            ctx.zeros(a, b, C.byref(ptr))
            assert ptr.value
            h[k] = ptr
        # @zeros really zeroes
        z = np.ones(shapes["Pr"], order="F")
        ctx.download(_vp(z), h["Pr"], z.size)
        assert not z.any()
        # initial condition, NavierStokes.jl:72-74
        vprof = init_vprof(p)
        ctx.upload(h["Vprof"], _vp(vprof), nx)
        vy = np.zeros((nx, ny + 1), order="F")
        vy[:, 0] = vprof
        ctx.upload(h["Vy"], _vp(vy), vy.size)
        back = np.empty_like(vy)
        ctx.download(_vp(back), h["Vy"], vy.size)
        assert same_bits(back, vy)
        ctx.copy(h["Vy_o"], h["Vy"], vy.size)
        ctx.download(_vp(back), h["Vy_o"], vy.size)
        assert same_bits(back, vy)
        st = _lib.CNSState(**{k: v.value for k, v in h.items()})
        cp = c_params(p)
        ref = OracleSim(p)
        he, hi = (C.c_double * 64)(), (C.c_int * 64)()
        for _ in range(2):
            iters, nh = C.c_int(0), C.c_int(0)
            ctx.ns_step(C.byref(st), C.byref(cp), C.byref(iters), he, hi, 64, C.byref(nh))
            assert iters.value == ref.step()
            assert list(hi[: nh.value]) == ref.iter_evo and list(he[: nh.value]) == ref.err_evo
        for k in ("Pr", "Vx", "Vy", "C"):
            out = np.empty(shapes[k], order="F")
            ctx.download(_vp(out), h[k], out.size)
            assert same_bits(out, getattr(ref, k)), k
    finally:
        for ptr in h.values():
            ctx.free(ptr)
        ctx.close()


def test_misuse_is_reported_not_computed(gpu):
    lib = _lib.load()
    assert lib.nsk_version().decode()
    # null context
    assert lib.nsk_sync(None) != 0 and b"null context" in lib.nsk_last_error()
    # device out of range
    hnd = C.c_void_p()
    assert lib.nsk_create(4096, C.byref(hnd)) != 0 and not hnd.value
    assert b"out of range" in lib.nsk_last_error()
    # unknown option / bad value
    with pytest.raises(_lib.NskError):
        gpu.set_option("no_such_option", 1)
    with pytest.raises(_lib.NskError):
        gpu.set_option("poisson_k", 9)
    with pytest.raises(_lib.NskError):
        gpu.set_option("poisson_div", 1)
    # grids the kernels cannot take, null arrays, bad pitch
    p = ns_params("NavierStokes", nx=33, ny=41)
    buf = C.c_void_p()
    gpu.zeros(64, 64, C.byref(buf))
    try:
        for bad in (dict(nx=2), dict(ny=2), dict(ld_pr=7), dict(ld_dp=3), dict(iy0=-1),
                    dict(iy0=10, nyg=20)):
            cp = c_params(p)
            for k, v in bad.items():
                setattr(cp, k, v)
            with pytest.raises(_lib.NskError):
                gpu.ns_poisson_iters(buf, buf, buf, C.byref(cp), 3)
        cp = c_params(p)
        with pytest.raises(_lib.NskError):
            gpu.ns_poisson_iters(None, buf, buf, C.byref(cp), 3)
        with pytest.raises(_lib.NskError):
            gpu.ns_poisson_iters(buf, buf, buf, C.byref(cp), -1)
        with pytest.raises(_lib.NskError):
            gpu.update_Pr(buf, buf, 0.1, 2, 2)
        with pytest.raises(_lib.NskError, match="null array"):
            gpu.update_Pr(buf, None, 0.1, 8, 8)
        with pytest.raises(_lib.NskError, match="null array"):
            gpu.advect(buf, buf, buf, None, buf, buf, 0.1, 0.1, 0.1, 7, 7, 0, 7)
        with pytest.raises(_lib.NskError, match="null array"):
            gpu.bub_distribute(buf, buf, None, buf, 1.0, 2.0, 0.1, 0.1, 6, 6, 4)
        with pytest.raises(_lib.NskError, match="null array"):
            gpu.copy(buf, None, 5)
        with pytest.raises(_lib.NskError):
            gpu.bub_solve_pressure(buf, buf, buf, buf, 1.2, 0.1, 0.1, 8, 8, 0, 1e-3, None, None)
        with pytest.raises(_lib.NskError):
            gpu.bub_solve_pressure_rb(buf, buf, buf, buf, 1.2, 0.1, 0.1, 1, 8, 10, 1e-3, None, None)
        with pytest.raises(_lib.NskError):
            gpu.zeros(-1, 5, C.byref(C.c_void_p()))
        gpu.ns_poisson_iters(buf, buf, buf, C.byref(cp), 0)      # zero passes: a no-op, not an error
        gpu.sync()
    finally:
        gpu.free(buf)
