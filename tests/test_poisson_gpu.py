"""The hot path: K passes of update_dPrdτ! + update_Pr! + set_bc_Pr! per launch
(scripts/NavierStokes.jl:85-87) against the oracle's pass-by-pass loop, bit for bit, for every
pass count per launch, both storage forms, whole domains and slab windows."""
import ctypes as C

import numpy as np
import pytest

from navierstokes_amd import ns_params
from navierstokes_amd.ns import c_params, _pitch
from util import Dev, rng_field, same_bits

pytestmark = pytest.mark.gpu


def oracle_iters(oracle, p, Pr, dP, divV, count, full):
    for _ in range(count):
        oracle.update_dPrdtau(Pr, dP, divV, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
        oracle.update_Pr(Pr, dP, p.dtau)
        oracle.set_bc_Pr(Pr, 0.0, bool(full))


def run_gpu(gpu, p, Pr, dP, divV, count, full, pitched, kmax=None, iy0=0, nyg=None):
    nx, ny = Pr.shape
    if pitched:
        ld = _pitch(nx)
        dPr, dDv = Dev(Pr, ld, pad_value=np.nan), Dev(divV, ld, pad_value=np.nan)
        emb = np.full((nx, ny), np.nan, order="F")
        emb[1:-1, 1:-1] = dP
        dDp = Dev(emb, ld, pad_value=np.nan)
        cp = c_params(p.with_(full_rows=bool(full)), ld_pr=ld, dp_embedded=1, iy0=iy0, nyg=nyg, ny=ny)
    else:
        dPr, dDv, dDp = Dev(Pr), Dev(divV), Dev(dP)
        cp = c_params(p.with_(full_rows=bool(full)), iy0=iy0, nyg=nyg, ny=ny)
    gpu.ns_poisson_iters(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count)
    out_dp = dDp.get()
    return dPr.get(), (np.asfortranarray(out_dp[1:-1, 1:-1]) if pitched else out_dp)


@pytest.mark.parametrize("pitched", [False, True])
@pytest.mark.parametrize("nx,ny", [(3, 3), (4, 7), (5, 4), (9, 33), (121, 40), (130, 67), (153, 255), (255, 153), (300, 97)])
def test_poisson_iters_whole_domain(gpu, oracle, nx, ny, pitched, monkeypatch):
    rng = np.random.default_rng(nx * 7919 + ny)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
    for full in (0, 1):
        if full == 0 and nx > ny:
            continue
        for count in (1, 2, 3, 4, 5, 8, 11):
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            oracle_iters(oracle, p, Pr, dP, dv, count, full)
            gPr, gdP = run_gpu(gpu, p, Pr0, dP0, dv, count, full, pitched)
            assert same_bits(gPr, Pr), (count, full, np.argwhere(gPr != Pr)[:5])
            assert same_bits(gdP, dP), (count, full, np.argwhere(gdP != dP)[:5])


@pytest.mark.parametrize("nx,ny", [(3, 3000), (3000, 3), (4, 2500), (2600, 5)])
def test_poisson_extreme_aspect_ratios(gpu, oracle, nx, ny):
    """One interior column / one interior row: many chunks of a single strip, many strips of a
    single chunk."""
    rng = np.random.default_rng(nx + ny)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
    for full in (0, 1):
        Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
        oracle_iters(oracle, p, Pr, dP, dv, 13, full)
        for pitched in (False, True):
            gPr, gdP = run_gpu(gpu, p, Pr0, dP0, dv, 13, full, pitched)
            assert same_bits(gPr, Pr) and same_bits(gdP, dP), (full, pitched)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8])
def test_poisson_every_pass_count(oracle, k, monkeypatch):
    """Each templated pass count per launch (library option NSK_POISSON_K) on the default grid."""
    import torch
    from navierstokes_amd import _lib
    monkeypatch.setenv("NSK_POISSON_K", str(k))
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        nx, ny = 153, 255
        rng = np.random.default_rng(k)
        p = ns_params("NavierStokes")
        Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
        for pitched in (False, True):
            for count in (k, 2 * k, 2 * k + 1):
                Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
                oracle_iters(oracle, p, Pr, dP, dv, count, 0)
                gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, count, 0, pitched)
                assert same_bits(gPr, Pr) and same_bits(gdP, dP), (pitched, count)
    finally:
        ctx.close()


def test_poisson_loop_with_checks(gpu, oracle):
    """nsk_ns_poisson = NavierStokes.jl:84-95 incl. the residual checks and the early exit."""
    p = ns_params("NavierStokes", nx=63, ny=105)
    nx, ny = p.nx, p.ny
    rng = np.random.default_rng(5)
    Pr0, dP0 = np.zeros((nx, ny), order="F"), np.zeros((nx - 2, ny - 2), order="F")
    dv = rng_field(rng, nx, ny, 0.05)
    Pr, dP, Rp = Pr0.copy(order="F"), dP0.copy(order="F"), np.zeros((nx - 2, ny - 2), order="F")
    it, ih, eh = oracle.poisson(Pr, dP, Rp, dv, p)
    assert 0 < it < p.niter, "test wants an early exit"
    dPr, dDp, dDv, dRp = Dev(Pr0), Dev(dP0), Dev(dv), Dev(Rp * 0)
    cp = c_params(p)
    iters, nh = C.c_int(0), C.c_int(0)
    he, hi = (C.c_double * 256)(), (C.c_int * 256)()
    gpu.ns_poisson(dPr.ptr, dDp.ptr, dRp.ptr, dDv.ptr, C.byref(cp), C.byref(iters), he, hi, 256, C.byref(nh))
    assert iters.value == it and nh.value == len(ih)
    assert list(hi[: nh.value]) == ih and list(he[: nh.value]) == eh
    assert same_bits(dPr.get(), Pr) and same_bits(dDp.get(), dP) and same_bits(dRp.get(), Rp)


def test_poisson_loop_err_scaling(gpu, oracle):
    """err = max|Rp|*ly^2/psc with ly^2 evaluated first (NavierStokes.jl:90): ly = 3 separates the
    association orders; the fused loop must report the oracle's (= the reference's) bits."""
    p = ns_params("NavierStokes", nx=9, ny=15, ly=3.0, psc=7.0, niter=8, nchk=4)
    nx, ny = p.nx, p.ny
    for seed in range(12):
        rng = np.random.default_rng(seed)
        Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny)
        Pr, dP, Rp = Pr0.copy(order="F"), dP0.copy(order="F"), np.zeros((nx - 2, ny - 2), order="F")
        it, ih, eh = oracle.poisson(Pr, dP, Rp, dv, p)
        dPr, dDp, dDv, dRp = Dev(Pr0), Dev(dP0), Dev(dv), Dev(Rp * 0)
        cp = c_params(p)
        iters, nh = C.c_int(0), C.c_int(0)
        he, hi = (C.c_double * 16)(), (C.c_int * 16)()
        gpu.ns_poisson(dPr.ptr, dDp.ptr, dRp.ptr, dDv.ptr, C.byref(cp), C.byref(iters), he, hi, 16, C.byref(nh))
        assert iters.value == it and list(hi[: nh.value]) == ih and list(he[: nh.value]) == eh, seed


@pytest.mark.parametrize("count", [1, 3, 4, 5, 12])
def test_poisson_slab_windows(gpu, oracle, count):
    """A window of a taller global grid: its first/last rows are halo rows that must stay untouched,
    and after `count` passes all rows further than `count` from a halo edge equal the global result."""
    nx, nyg = 70, 96
    p = ns_params("NavierStokes", nx=nx, ny=nyg)
    rng = np.random.default_rng(11)
    Pr0, dP0, dv = rng_field(rng, nx, nyg), rng_field(rng, nx - 2, nyg - 2), rng_field(rng, nx, nyg, 10.0)
    Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
    oracle_iters(oracle, p, Pr, dP, dv, count, 0)
    for lo, hi in [(0, 40), (30, 75), (18, 72), (60, 96), (48, 96)]:   # global 0-based row ranges [lo, hi)
        wPr = np.asfortranarray(Pr0[:, lo:hi])
        wdv = np.asfortranarray(dv[:, lo:hi])
        wdP = np.asfortranarray(dP0[:, lo:hi - 2])     # dP row j <-> Pr row j+1
        for pitched in (False, True):
            gPr, gdP = run_gpu(gpu, p, wPr, wdP, wdv, count, 0, pitched, iy0=lo, nyg=nyg)
            a = 0 if lo == 0 else count
            b = (hi - lo) if hi == nyg else (hi - lo) - count
            assert same_bits(gPr[:, a:b], Pr[:, lo + a:lo + b]), (lo, hi, pitched)
            nyw = hi - lo
            pa, pb = max(a, 1), min(b, nyw - 1)            # window Pr rows whose dPrdτ is valid
            assert same_bits(gdP[:, pa - 1:pb - 1], dP[:, lo + pa - 1:lo + pb - 1]), (lo, hi, pitched)
            if lo > 0:
                assert same_bits(gPr[:, 0], wPr[:, 0])
            if hi < nyg:
                assert same_bits(gPr[:, -1], wPr[:, -1])


@pytest.mark.parametrize("div", [0, 2, 3, 4])
def test_poisson_division_forms(oracle, div):
    """x/dx/dx is evaluated either by IEEE division or by an FMA sequence that yields the same
    correctly rounded quotient.  Forced forms on: ordinary data, data near the underflow range
    and subnormal data (where the FMA form must hand over to IEEE division), exact zeros, huge
    values, and fields that contain Inf / NaN."""
    import torch
    from navierstokes_amd import _lib
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    ctx.set_option("poisson_div", div)
    try:
        nx, ny = 153, 255
        p = ns_params("NavierStokes")
        rng = np.random.default_rng(100 + div)
        for scale, zero_frac in [(1.0, 0.0), (1e-250, 0.0), (1e-262, 0.2), (1e-290, 0.0), (1e-305, 0.3), (1e-318, 0.5),
                                 (1.0, 0.7), (1e290, 0.0), (1e300, 0.0), (-1.0, 0.0)]:
            special = scale < 0
            scale = abs(scale)
            Pr0 = rng_field(rng, nx, ny, scale)
            dP0 = rng_field(rng, nx - 2, ny - 2, scale)
            dv = rng_field(rng, nx, ny, scale)
            if special:     # Inf and NaN somewhere in the field, a subnormal island elsewhere
                Pr0[40, 60], Pr0[100, 200], Pr0[120, 30] = np.inf, np.nan, -np.inf
                Pr0[60:70, 120:130] *= 1e-315
            if zero_frac:
                Pr0[rng.random((nx, ny)) < zero_frac] = 0.0
                Pr0[:, ny // 2:] = 0.0
                dP0[:, ny // 2:] = 0.0
                dv[:, ny // 2:] = 0.0
            for pitched in (False, True):
                Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
                with np.errstate(all="ignore"):
                    oracle_iters(oracle, p, Pr, dP, dv, 9, 0)
                gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, 9, 0, pitched)
                assert ctx.get_option("poisson_div_used") == div      # dx = dy = 1/255 qualifies for every form
                if scale < 1e200 and not special:
                    assert np.isfinite(Pr).all()
                # Inf and NaN must sit in the same cells (same_bits lets any NaN equal any NaN)
                assert np.array_equal(np.isnan(gPr), np.isnan(Pr)) and np.array_equal(np.isinf(gPr), np.isinf(Pr))
                assert same_bits(gPr, Pr), (scale, zero_frac, pitched)
                assert same_bits(gdP, dP), (scale, zero_frac, pitched)
    finally:
        ctx.close()


def test_poisson_full_size_grid(oracle):
    """The benchmark grid (NavierStokes_Re6.jl at n = 16: 2559 x 4095, 23 strips x ~80 chunks, all
    task shapes and the XCD-aware task numbering in play): 11 passes against the oracle, and the
    size-independent property that the result does not depend on how the passes are grouped into
    launches or how the grid is cut into tasks."""
    import torch
    from navierstokes_amd import _lib
    p = ns_params("NavierStokes_Re6", n=16)
    nx, ny = p.nx, p.ny
    assert (nx, ny) == (2559, 4095)
    rng = np.random.default_rng(2559)
    Pr0, dP0, dv = rng_field(rng, nx, ny, 1e-2), rng_field(rng, nx - 2, ny - 2, 1e-2), rng_field(rng, nx, ny, 1e-2)
    Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
    oracle_iters(oracle, p, Pr, dP, dv, 11, 0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        # (passes per launch, rows per task, edge-strip height %, XCD numbering, non-temporal row requests)
        for k, rows, edge, xcd, stream in [(0, 0, 0, 1, -1), (0, 0, 0, 0, 0), (0, 0, 0, 1, 1), (3, 37, 50, 1, 1),
                                           (8, 200, 100, 1, -1), (1, 0, 0, 1, 0)]:
            ctx.set_option("poisson_k", k)
            ctx.set_option("poisson_rows", rows)
            ctx.set_option("poisson_edge_percent", edge)
            ctx.set_option("poisson_xcd", xcd)
            ctx.set_option("poisson_stream", stream)
            assert ctx.get_option("poisson_stream") == stream
            gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, 11, 0, True)
            assert same_bits(gPr, Pr), (k, rows, edge, xcd, stream, np.argwhere(gPr != Pr)[:5])
            assert same_bits(gdP, dP), (k, rows, edge, xcd, stream, np.argwhere(gdP != dP)[:5])
        ctx.set_option("poisson_k", 0)
        ctx.set_option("poisson_rows", 0)
        ctx.set_option("poisson_stream", -1)
        gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, 11, 0, False)      # dense, odd leading dimension
        assert same_bits(gPr, Pr) and same_bits(gdP, dP)
    finally:
        ctx.close()


def test_poisson_slab_windows_full_width(gpu, oracle):
    """Slab windows of the benchmark grid (what each rank of a 2-GPU run computes between two
    halo exchanges): 12 passes on the lower and the upper window equal the single-domain result
    on all rows further than 12 from the windows' halo edges."""
    p = ns_params("NavierStokes_Re6", n=16)
    nx, nyg, count = p.nx, p.ny, 12
    rng = np.random.default_rng(4095)
    Pr0, dP0, dv = rng_field(rng, nx, nyg, 1e-2), rng_field(rng, nx - 2, nyg - 2, 1e-2), rng_field(rng, nx, nyg, 1e-2)
    Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
    oracle_iters(oracle, p, Pr, dP, dv, count, 0)
    for lo, hi in [(0, 2112), (1984, nyg)]:
        wPr, wdv = np.asfortranarray(Pr0[:, lo:hi]), np.asfortranarray(dv[:, lo:hi])
        wdP = np.asfortranarray(dP0[:, lo:hi - 2])
        gPr, gdP = run_gpu(gpu, p, wPr, wdP, wdv, count, 0, True, iy0=lo, nyg=nyg)
        a = 0 if lo == 0 else count
        b = (hi - lo) if hi == nyg else (hi - lo) - count
        assert same_bits(gPr[:, a:b], Pr[:, lo + a:lo + b]), (lo, hi)
        pa, pb = max(a, 1), min(b, hi - lo - 1)
        assert same_bits(gdP[:, pa - 1:pb - 1], dP[:, lo + pa - 1:lo + pb - 1]), (lo, hi)


def test_planner_picks_passes_per_launch_by_grid(gpu, oracle):
    """Left to itself the library chooses the passes per launch and the task height per grid: few
    passes and short tasks on a small grid (where a task's halo rows dominate), five or six passes on
    the benchmark grid; never more passes than were asked for.  The choice never changes results
    (every choice is covered by the parity tests above); here only the choice is checked."""
    for opt in ("poisson_k", "poisson_rows"):
        gpu.set_option(opt, 0)
    gpu.set_option("poisson_small", 0)      # (the small grid would otherwise run its 40 passes in one launch)
    try:
        rng = np.random.default_rng(77)
        for n, want in ((1, (1, 2, 3, 4)), (16, (5, 6))):
            p = ns_params("NavierStokes_Re6", n=n)
            Pr0, dP0, dv = rng_field(rng, p.nx, p.ny, 1e-2), rng_field(rng, p.nx - 2, p.ny - 2, 1e-2), rng_field(rng, p.nx, p.ny, 1e-2)
            run_gpu(gpu, p, Pr0, dP0, dv, 40, 0, True)
            assert gpu.get_option("poisson_k") in want, (n, gpu.get_option("poisson_k"))
            assert gpu.get_option("poisson_rows_used") >= 2
            # tasks in pairs on the benchmark grid (two wavefronts per SIMD either way), plain tasks on the
            # small one (a strip of one or two chunks has nothing to pair)
            assert gpu.get_option("poisson_pairs_used") == (1 if n == 16 else 0), n
        gpu.set_option("poisson_pairs", 0)
        run_gpu(gpu, p, Pr0, dP0, dv, 12, 0, True)
        assert gpu.get_option("poisson_pairs_used") == 0
        gpu.set_option("poisson_pairs", 1)
        run_gpu(gpu, p, Pr0, dP0, dv, 3, 0, True)
        assert gpu.get_option("poisson_k") <= 3
    finally:
        gpu.set_option("poisson_small", 1)
        gpu.set_option("poisson_pairs", 1)


@pytest.mark.parametrize("nx,ny", [(45, 75), (130, 67), (257, 300)])
def test_poisson_dense_arrays_staged(gpu, oracle, nx, ny):
    """Dense arrays (the reference's own layout: odd leading dimension, dPrdτ as its own
    (nx-2) x (ny-2) array) are iterated on pitched staging copies when the run of passes is long
    enough; with staging switched off the register-input kernel does the same passes in place."""
    rng = np.random.default_rng(nx * 3 + ny)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
    Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
    oracle_iters(oracle, p, Pr, dP, dv, 45, 0)
    try:
        for stage in (1, 0):
            gpu.set_option("poisson_stage", stage)
            gPr, gdP = run_gpu(gpu, p, Pr0, dP0, dv, 45, 0, False)
            assert same_bits(gPr, Pr) and same_bits(gdP, dP), stage
    finally:
        gpu.set_option("poisson_stage", 1)


@pytest.mark.parametrize("pitched", [True, False])
def test_edges_first_entry_point(gpu, oracle, pitched):
    """nsk_ns_poisson_iters_edges_first: the last launch produces the edge rows first (on a second
    stream), the rows between them on the context's stream and the skipped rows not at all.  All
    rows but the skipped ones equal nsk_ns_poisson_iters bit for bit, for every way the call can go:
    cut launch, a single launch (nothing to cut: plain sequence), no interior rows, no edge rows,
    dense arrays on staging copies; work queued on either stream afterwards sees finished rows."""
    import torch
    nx, nyg = 150, 140
    p = ns_params("NavierStokes", nx=nx, ny=nyg)
    rng = np.random.default_rng(31)
    Pr0, dP0, dv = rng_field(rng, nx, nyg), rng_field(rng, nx - 2, nyg - 2), rng_field(rng, nx, nyg, 10.0)
    lo, hi = 20, 120                                  # a window with halo rows on both sides
    wPr, wdv, wdP = (np.asfortranarray(Pr0[:, lo:hi]), np.asfortranarray(dv[:, lo:hi]),
                     np.asfortranarray(dP0[:, lo:hi - 2]))
    nyw = hi - lo
    edge = torch.cuda.Stream()
    for count, skip_lo, e_lo, e_hi, skip_hi in [(12, 8, 8, 8, 8), (40, 8, 8, 8, 8), (1, 8, 8, 8, 8), (13, 0, 0, 10, 6),
                                                (12, 5, 20, 0, 0), (12, 10, 40, 40, 10), (12, 0, 0, 0, 0), (7, 3, 1, 1, 3)]:
        want_Pr, want_dP = run_gpu(gpu, p, wPr, wdP, wdv, count, 0, pitched, iy0=lo, nyg=nyg)
        if pitched:
            ld = _pitch(nx)
            dPr, dDv = Dev(wPr, ld, pad_value=np.nan), Dev(wdv, ld, pad_value=np.nan)
            emb = np.full((nx, nyw), np.nan, order="F")
            emb[1:-1, 1:-1] = wdP
            dDp = Dev(emb, ld, pad_value=np.nan)
            cp = c_params(p, ld_pr=ld, dp_embedded=1, iy0=lo, nyg=nyg, ny=nyw)
        else:
            dPr, dDv, dDp = Dev(wPr), Dev(wdv), Dev(wdP)
            cp = c_params(p, iy0=lo, nyg=nyg, ny=nyw)
        gpu.ns_poisson_iters_edges_first(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count, skip_lo, e_lo, e_hi, skip_hi,
                                         C.c_void_p(edge.cuda_stream))
        with torch.cuda.stream(edge):                 # ordered behind the edge rows only
            e_rows = torch.cat([dPr.t[skip_lo:skip_lo + e_lo], dPr.t[nyw - skip_hi - e_hi:nyw - skip_hi]]).clone()
        edge.synchronize()
        got_Pr = dPr.get()
        got_dP = dDp.get()
        if pitched:
            got_dP = np.asfortranarray(got_dP[1:-1, 1:-1])
        a, b = skip_lo, nyw - skip_hi
        assert same_bits(got_Pr[:, a:b], want_Pr[:, a:b]), (count, skip_lo, e_lo, e_hi, skip_hi)
        pa, pb = max(a, 1), min(b, nyw - 1)
        assert same_bits(got_dP[:, pa - 1:pb - 1], want_dP[:, pa - 1:pb - 1]), (count, skip_lo, e_lo, e_hi, skip_hi)
        want_e = np.concatenate([want_Pr[:, a:a + e_lo], want_Pr[:, b - e_hi:b]], axis=1)
        assert same_bits(e_rows.cpu().numpy().T[:nx], want_e), "edge rows as seen from the edge stream"
    with pytest.raises(Exception, match="do not fit"):
        gpu.ns_poisson_iters_edges_first(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), 6, 60, 10, 10, 60, C.c_void_p(edge.cuda_stream))
    with pytest.raises(Exception, match="edge_stream is null"):
        gpu.ns_poisson_iters_edges_first(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), 6, 1, 1, 1, 1, None)


def test_tasks_run_again_wait_for_all_their_rows(gpu, oracle):
    """Every task run twice: subnormal pressure values put each task outside the range of the fast division, so
    it is produced again with IEEE division from the rows its first run has just fetched.  A tall narrow grid
    in one-pass launches makes ~1500 tasks of two rows that all do so at once.  The second run requests its
    first rows into the LDS ring slots the first run used last; the first run's last LDS write (the scaled
    grad V row) could land after the freshly requested row, which was then scaled twice: the first output row
    of a few tasks per launch was wrong in one call of five.  Repeated, because the failure was a race."""
    import torch
    nx, ny = 127, 1574
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    rng = np.random.default_rng(903)
    def field(n, m, scale):
        f = np.asfortranarray(rng.standard_normal((n, m)) * scale)
        f[rng.random((n, m)) < 0.3] = 0.0
        return f
    Pr0, dP0, dv = field(nx, ny, 1e-310), field(nx - 2, ny - 2, 1e-310), field(nx, ny, 1e-309)
    edge = torch.cuda.Stream()
    try:
        for k, count in ((1, 2), (1, 3), (2, 4)):
            gpu.set_option("poisson_k", k)
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            with np.errstate(all="ignore"):
                oracle_iters(oracle, p, Pr, dP, dv, count, 1)
            for rep in range(4):
                gPr, gdP = run_gpu(gpu, p, Pr0, dP0, dv, count, 1, True)
                assert same_bits(gPr, Pr), (k, count, rep, np.argwhere(gPr != Pr)[:5])
                assert same_bits(gdP, dP), (k, count, rep, np.argwhere(gdP != dP)[:5])
                # the last launch cut in three (edge rows on a second stream)
                ld = _pitch(nx)
                dPr, dDv = Dev(Pr0, ld, pad_value=np.nan), Dev(dv, ld, pad_value=np.nan)
                emb = np.full((nx, ny), np.nan, order="F")
                emb[1:-1, 1:-1] = dP0
                dDp = Dev(emb, ld, pad_value=np.nan)
                cp = c_params(p.with_(full_rows=True), ld_pr=ld, dp_embedded=1, ny=ny)
                gpu.ns_poisson_iters_edges_first(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count, 0, 369, 732, 0,
                                                 C.c_void_p(edge.cuda_stream))
                edge.synchronize()
                gPr, gdP = dPr.get(), np.asfortranarray(dDp.get()[1:-1, 1:-1])
                assert same_bits(gPr, Pr), ("edges first", k, count, rep, np.argwhere(gPr != Pr)[:5])
                assert same_bits(gdP, dP), ("edges first", k, count, rep, np.argwhere(gdP != dP)[:5])
    finally:
        gpu.set_option("poisson_k", 0)
