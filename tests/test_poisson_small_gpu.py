"""Runs of Poisson passes (scripts/NavierStokes.jl:85-87) on grids whose fields fit on chip: one launch
for the whole run, the rows dealt to workgroups that hand their edge rows to each other every few
passes.  Against the oracle's pass-by-pass loop and against the launch-per-K-passes kernel, bit for
bit: both storage forms, both row ranges of bc_x!, pass counts that are and are not multiples of the
exchange interval, every tile shape the options reach, data that forces the second launch to run
the passes again with IEEE division."""
import numpy as np
import pytest

from navierstokes_amd import ns_params
from test_poisson_gpu import oracle_iters, run_gpu
from util import rng_field, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    import torch
    from navierstokes_amd import _lib
    c = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def fields(nx, ny, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return rng_field(rng, nx, ny, scale), rng_field(rng, nx - 2, ny - 2, scale), rng_field(rng, nx, ny, 10.0 * scale)


@pytest.mark.parametrize("pitched", [False, True])
@pytest.mark.parametrize("nx,ny", [(153, 255), (3, 3), (5, 9), (16, 24), (100, 37), (3, 3000), (341, 100), (170, 2047), (64, 8),
                                   (37, 17)])
def test_one_launch_equals_the_pass_by_pass_loop(ctx, oracle, nx, ny, pitched):
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    Pr0, dP0, dv = fields(nx, ny, nx * 31 + ny)
    for full in (0, 1):
        if full == 0 and nx > ny:
            continue
        for count in (12, 13, 16, 17, 40):
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            oracle_iters(oracle, p, Pr, dP, dv, count, full)
            gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, count, full, pitched)
            assert ctx.get_option("poisson_k") == count, "the whole run was to be one launch"
            assert same_bits(gPr, Pr), (count, full, np.argwhere(gPr != Pr)[:5])
            assert same_bits(gdP, dP), (count, full, np.argwhere(gdP != dP)[:5])


@pytest.mark.parametrize("halo,cols,rows", [(8, 8, 16), (4, 10, 8), (8, 16, 24), (3, 10, 7), (1, 2, 2), (6, 60, 12), (8, 300, 300),
                                            (8, 16, 100), (2, 60, 3), (5, 9, 9)])
def test_tile_shapes(ctx, oracle, halo, cols, rows):
    """Tiles and passes between exchanges as the options set them (2 x 2: an exchange after every pass;
    60 columns: one column of tiles; 300 x 300: one workgroup, no exchange -- on a grid small enough for
    its cell slots) -- none of it changes a bit."""
    nx, ny = (60, 255) if cols < 300 else (60, 50)
    if cols * rows <= 4:
        nx, ny = 16, 16         # (a workgroup per 2 x 2 cells: no more of them than compute units)
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    Pr0, dP0, dv = fields(nx, ny, halo * 100 + rows)
    ctx.set_option("poisson_small_halo", halo)
    ctx.set_option("poisson_small_rows", rows)
    ctx.set_option("poisson_small_cols", cols)
    for count in (12, 31):
        for full in (0, 1):
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            oracle_iters(oracle, p, Pr, dP, dv, count, full)
            gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, count, full, True)
            assert ctx.get_option("poisson_k") == count
            assert ctx.get_option("poisson_rows_used") == min(rows, ny)
            assert same_bits(gPr, Pr) and same_bits(gdP, dP), (count, np.argwhere(gPr != Pr)[:5])


@pytest.mark.parametrize("nx,ny", [(33, 49), (17, 25), (153, 241)])
def test_ragged_last_tiles_of_one_column_or_row_are_avoided(ctx, oracle, nx, ny):
    """The cells on the rim of the grid are written by their inner neighbours, which therefore have to belong
    to the same tile: tile sizes that would leave a single column or row for the last tile are passed over."""
    p = ns_params("NavierStokes", nx=nx, ny=ny)
    Pr0, dP0, dv = fields(nx, ny, nx + ny)
    for opts in ({}, {"poisson_small_cols": 16, "poisson_small_rows": 8}, {"poisson_small_cols": 8, "poisson_small_rows": 16}):
        for name in ("poisson_small_cols", "poisson_small_rows"):
            ctx.set_option(name, opts.get(name, 0))
        Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
        oracle_iters(oracle, p, Pr, dP, dv, 20, 1)
        gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, 20, 1, True)
        assert same_bits(gPr, Pr) and same_bits(gdP, dP), opts


def test_switched_off_is_the_launch_per_k_passes_kernel(ctx, oracle):
    nx, ny = 153, 255
    p = ns_params("NavierStokes")
    Pr0, dP0, dv = fields(nx, ny, 5)
    got = {}
    for on in (1, 0):
        ctx.set_option("poisson_small", on)
        got[on] = run_gpu(ctx, p, Pr0, dP0, dv, 100, 0, True)
        assert (ctx.get_option("poisson_k") == 100) == bool(on)
    assert same_bits(got[0][0], got[1][0]) and same_bits(got[0][1], got[1][1])


@pytest.mark.parametrize("div", [-1, 0, 2, 3])
def test_division_forms_in_one_launch(ctx, oracle, div):
    """The exact two-operation division holds for finite pressure values that are 0 or >= 2^-848: the first
    launch notices values outside that range and the second launch runs the passes again with IEEE
    division, in place.  Ordinary data, data near and inside the subnormal range, zeros, huge values, Inf
    and NaN; forms 3 and 4 do not exist in this kernel and mean IEEE division."""
    ctx.set_option("poisson_div", div)
    nx, ny = 153, 255
    p = ns_params("NavierStokes")
    rng = np.random.default_rng(300 + div)
    for scale, zero_frac in [(1.0, 0.0), (1e-250, 0.0), (1e-262, 0.2), (1e-305, 0.3), (1e-318, 0.5), (1.0, 0.7), (1e300, 0.0),
                             (-1.0, 0.0)]:
        special = scale < 0
        scale = abs(scale)
        Pr0, dP0, dv = rng_field(rng, nx, ny, scale), rng_field(rng, nx - 2, ny - 2, scale), rng_field(rng, nx, ny, scale)
        if special:
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
                oracle_iters(oracle, p, Pr, dP, dv, 21, 0)
            gPr, gdP = run_gpu(ctx, p, Pr0, dP0, dv, 21, 0, pitched)
            assert ctx.get_option("poisson_k") == 21
            assert ctx.get_option("poisson_div_used") == (2 if div in (-1, 2) else 0)
            assert np.array_equal(np.isnan(gPr), np.isnan(Pr)) and np.array_equal(np.isinf(gPr), np.isinf(Pr))
            assert same_bits(gPr, Pr), (scale, zero_frac, pitched)
            assert same_bits(gdP, dP), (scale, zero_frac, pitched)


def test_whole_time_steps_take_the_one_launch_path(oracle):
    """NavierStokes.jl at its default size: three time steps with the one-launch passes equal three with
    the launch-per-K-passes kernel (and both equal the oracle's in test_ns_step_gpu)."""
    from navierstokes_amd.ns import NavierStokes
    p = ns_params("NavierStokes")
    out = {}
    for on in (1, 0):
        sim = NavierStokes(p)
        sim.ctx.set_option("poisson_small", on)
        its = [sim.step() for _ in range(3)]
        out[on] = (its, list(sim.host_fields().values()) + [sim.dPrdtau_host()])
    assert out[0][0] == out[1][0]
    for a, b in zip(out[0][1], out[1][1]):
        assert same_bits(a, b)


def test_a_workgroup_that_never_arrives_ends_in_an_error(ctx):
    """Every wait between workgroups is bounded: with one workgroup gone (a test option makes it leave at
    once) its neighbours give up after the set number of polls, everybody else follows, the launch ends,
    and the next synchronising call reports it; the context works again afterwards."""
    from navierstokes_amd import _lib
    nx, ny = 153, 255
    p = ns_params("NavierStokes")
    Pr0, dP0, dv = fields(nx, ny, 9)
    ctx.set_option("poisson_small_spin_limit", 2000)
    ctx.set_option("poisson_small_test_absent", 1 + 57)
    with pytest.raises(_lib.NskError, match="gave up waiting") as e:
        run_gpu(ctx, p, Pr0, dP0, dv, 40, 0, True)
        ctx.sync()
    assert e.value.code == _lib.ERR_TIMEOUT
    ctx.set_option("poisson_small_test_absent", 0)
    ctx.set_option("poisson_small_spin_limit", 0)
    want = run_gpu(ctx, p, Pr0, dP0, dv, 40, 0, True)
    ctx.set_option("poisson_small", 0)
    got = run_gpu(ctx, p, Pr0, dP0, dv, 40, 0, True)
    assert same_bits(want[0], got[0]) and same_bits(want[1], got[1])


def test_two_contexts_on_two_streams_side_by_side(oracle):
    """The workgroups of a one-launch run wait for each other, so two such launches side by side could starve
    each other of compute units; within a process the library chains them on the device.  Two contexts on
    two streams, launches queued alternately with no host wait in between: both give the oracle's bits."""
    import ctypes as C
    import torch
    from navierstokes_amd import _lib
    from navierstokes_amd.ns import _pitch, c_params
    from util import Dev
    nx, ny, count = 153, 255, 60
    p = ns_params("NavierStokes")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    ctxs = [_lib.Context(0, st.cuda_stream) for st in streams]
    try:
        cases = []
        for i, k in enumerate(ctxs):
            Pr0, dP0, dv = fields(nx, ny, 70 + i)
            ld = _pitch(nx)
            emb = np.full((nx, ny), np.nan, order="F")
            emb[1:-1, 1:-1] = dP0
            dev = (Dev(Pr0, ld, pad_value=np.nan), Dev(emb, ld, pad_value=np.nan), Dev(dv, ld, pad_value=np.nan))
            Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
            oracle_iters(oracle, p, Pr, dP, dv, 3 * count, 0)
            cases.append((k, dev, Pr, dP))
        torch.cuda.synchronize()
        cp = c_params(p, ld_pr=_pitch(nx), dp_embedded=1)
        for _ in range(3):
            for k, dev, _, _ in cases:
                k.ns_poisson_iters(dev[0].ptr, dev[1].ptr, dev[2].ptr, C.byref(cp), count)
        for k, dev, Pr, dP in cases:
            k.sync()
            assert k.get_option("poisson_k") == count
            assert same_bits(dev[0].get(), Pr)
            assert same_bits(np.asfortranarray(dev[1].get()[1:-1, 1:-1]), dP)
    finally:
        for k in ctxs:
            k.close()
