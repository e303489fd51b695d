"""nsk_ns_poisson_iters_static: the variant of nsk_ns_poisson_iters that allocates nothing (its
ping-pong partners are a static device array shared by all calls on a GPU).  Checked through the
C ABI against nsk_ns_poisson_iters, bit for bit, on calls in a row that reuse the array with
different layouts, the later ones smaller than the first."""
import ctypes as C

import numpy as np
import pytest

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import c_params, _pitch
from util import Dev, rng_field, same_bits

pytestmark = pytest.mark.gpu

STATIC_ELEMS = 4 << 20      # values per field the static array holds (include/nsk.h)


def run(ctx, entry, p, Pr, dP, divV, count, pitched):
    nx, ny = Pr.shape
    if pitched:
        ld = _pitch(nx)
        dPr, dDv = Dev(Pr, ld, pad_value=np.nan), Dev(divV, ld, pad_value=np.nan)
        emb = np.full((nx, ny), np.nan, order="F")
        emb[1:-1, 1:-1] = dP
        dDp = Dev(emb, ld, pad_value=np.nan)
        cp = c_params(p, ld_pr=ld, dp_embedded=1)
    else:
        dPr, dDv, dDp = Dev(Pr), Dev(divV), Dev(dP)
        cp = c_params(p)
    getattr(ctx, entry)(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count)
    out_dp = dDp.get()
    return dPr.get(), (np.asfortranarray(out_dp[1:-1, 1:-1]) if pitched else out_dp)


def test_static_entry_point_equals_allocating_one(gpu, oracle):
    rng = np.random.default_rng(20)
    # first a large window (1279 x 2047 pitched = 2 620 160 values of the 4 194 304 the array holds),
    # then smaller ones in both storage forms, odd and even launch counts
    cases = [(1279, 2047, True, 13), (153, 255, True, 12), (153, 255, False, 7), (300, 97, True, 40),
             (45, 75, False, 45), (9, 33, True, 1)]
    for nx, ny, pitched, count in cases:
        p = ns_params("NavierStokes", nx=nx, ny=ny)
        Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
        want_Pr, want_dP = run(gpu, "ns_poisson_iters", p, Pr0, dP0, dv, count, pitched)
        got_Pr, got_dP = run(gpu, "ns_poisson_iters_static", p, Pr0, dP0, dv, count, pitched)
        assert same_bits(got_Pr, want_Pr), (nx, ny, pitched, count)
        assert same_bits(got_dP, want_dP), (nx, ny, pitched, count)
        if nx * ny < 50000:     # and both equal the oracle's pass-by-pass loop
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            for _ in range(count):
                oracle.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
                oracle.update_Pr(Pr, dP, p.dtau)
                oracle.set_bc_Pr(Pr, 0.0, False)
            assert same_bits(got_Pr, Pr) and same_bits(got_dP, dP), (nx, ny, pitched, count)


def test_static_entry_point_same_pitch_narrower_grid(gpu):
    """Two grids of different width can share a row pitch (140 and 130 columns both take 144): the columns
    between the narrower grid's last one and the pitch then hold what the wider grid left there -- huge and
    non-finite values here -- unless the call clears them, as the allocating entry point's partners are."""
    rng = np.random.default_rng(22)
    wide = ns_params("NavierStokes", nx=140, ny=67)
    assert _pitch(140) == _pitch(130) == 144
    Pw = rng_field(rng, 140, 67) * 1e300
    Pw[131:, :] = np.inf
    Pw[135, 5] = np.nan
    run(gpu, "ns_poisson_iters_static", wide, Pw, rng_field(rng, 138, 65) * 1e300, rng_field(rng, 140, 67), 7, True)
    p = ns_params("NavierStokes", nx=130, ny=67)
    Pr0, dP0, dv = rng_field(rng, 130, 67), rng_field(rng, 128, 65), rng_field(rng, 130, 67, 10.0)
    for count in (9, 4):
        want = run(gpu, "ns_poisson_iters", p, Pr0, dP0, dv, count, True)
        got = run(gpu, "ns_poisson_iters_static", p, Pr0, dP0, dv, count, True)
        assert same_bits(got[0], want[0]) and same_bits(got[1], want[1]), count


def test_static_entry_point_allocates_nothing():
    """What the entry point is for: on a context whose ping-pong partners and staging copies do not exist
    yet, calls of the static entry point leave the device's free memory where it was -- in both storage
    forms, the dense one included, which the allocating entry point would stage through copies of its own
    on a long run."""
    import torch
    torch.cuda.synchronize()
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)        # fresh: nothing allocated on demand yet
    try:
        rng = np.random.default_rng(23)
        nx, ny = 301, 199
        p = ns_params("NavierStokes", nx=nx, ny=ny)
        Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
        ld = _pitch(nx)
        emb = np.zeros((nx, ny), order="F")
        emb[1:-1, 1:-1] = dP0
        arrays = {True: (Dev(Pr0, ld), Dev(emb, ld), Dev(dv, ld), c_params(p, ld_pr=ld, dp_embedded=1)),
                  False: (Dev(Pr0), Dev(dP0), Dev(dv), c_params(p))}
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        for pitched in (True, False):
            a, b, c, cp = arrays[pitched]
            for count in (64, 5):                   # 64 passes: a run the allocating entry point would stage when dense
                ctx.ns_poisson_iters_static(a.ptr, b.ptr, c.ptr, C.byref(cp), count)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] == free0
        # ... while the allocating entry point, on the same context, does take memory for its partners
        a, b, c, cp = arrays[True]
        big = ns_params("NavierStokes", nx=1279, ny=2047)
        t = torch.zeros((3, 2047, _pitch(1279)), dtype=torch.float64, device="cuda:0")
        torch.cuda.synchronize()
        free1 = torch.cuda.mem_get_info(0)[0]
        cpb = c_params(big, ld_pr=_pitch(1279), dp_embedded=1)
        ctx.ns_poisson_iters(C.c_void_p(t[0].data_ptr()), C.c_void_p(t[1].data_ptr()), C.c_void_p(t[2].data_ptr()), C.byref(cpb), 4)
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info(0)[0] < free1
    finally:
        ctx.close()


def test_static_entry_point_from_a_second_context(gpu):
    """The array belongs to the GPU, not to a context: a second context uses the same one."""
    import torch
    ctx2 = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        rng = np.random.default_rng(21)
        p = ns_params("NavierStokes", nx=130, ny=67)
        Pr0, dP0, dv = rng_field(rng, 130, 67), rng_field(rng, 128, 65), rng_field(rng, 130, 67, 10.0)
        want = run(gpu, "ns_poisson_iters", p, Pr0, dP0, dv, 9, True)
        a = run(gpu, "ns_poisson_iters_static", p, Pr0, dP0, dv, 9, True)
        b = run(ctx2, "ns_poisson_iters_static", p, Pr0, dP0, dv, 9, True)
        for got in (a, b):
            assert same_bits(got[0], want[0]) and same_bits(got[1], want[1])
    finally:
        ctx2.close()


def test_static_entry_point_rejects_windows_beyond_its_array(gpu):
    import torch
    nx, ny = 2047, 2049                      # pitched 2048 x 2049 = 4 196 352 > 4 194 304
    ld = _pitch(nx)
    assert ld * ny > STATIC_ELEMS
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    t = torch.zeros((3, ny, ld), dtype=torch.float64, device="cuda:0")
    cp = c_params(p, ld_pr=ld, dp_embedded=1)
    ptr = lambda i: C.c_void_p(t[i].data_ptr())
    with pytest.raises(_lib.NskError, match="exceeds the static scratch"):
        gpu.ns_poisson_iters_static(ptr(0), ptr(1), ptr(2), C.byref(cp), 4)
    gpu.ns_poisson_iters(ptr(0), ptr(1), ptr(2), C.byref(cp), 4)     # the allocating one takes it
    # one row fewer fits exactly
    p2 = ns_params("NavierStokes", nx=nx, ny=2048)
    cp2 = c_params(p2, ld_pr=ld, dp_embedded=1)
    gpu.ns_poisson_iters_static(ptr(0), ptr(1), ptr(2), C.byref(cp2), 4)
    torch.cuda.synchronize()
    assert float(t[:, :, :nx].abs().max()) == 0.0      # (the pad column holds don't-care values)
