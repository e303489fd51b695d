"""The slab decomposition behind the C ABI (nsk_slabs_*, include/nsk.h): several slabs driven by one
host thread, halo rows moved by device copies or by RCCL.  Only one GPU is available to the tests, so
all slabs of a group live on cuda:0: the copy transport runs 2-4 slabs there against the
single-domain oracle (windows, halo bookkeeping, stream/event ordering, scatter/gather of the
reference's global arrays, the residual and the whole loop NavierStokes.jl:84-95); the RCCL
transport runs with one slab per communicator -- set-up through a unique id, the all-reduce, and a
send/receive of row blocks to itself on the communication stream."""
import ctypes as C

import numpy as np
import pytest

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import c_params
from util import rng_field, same_bits

pytestmark = pytest.mark.gpu


def fields(rng, nx, nyg, scale=1.0):
    return rng_field(rng, nx, nyg, scale), rng_field(rng, nx - 2, nyg - 2, scale), rng_field(rng, nx, nyg, 10.0 * scale)


def oracle_passes(oracle, p, Pr, dP, dv, count):
    for _ in range(count):
        oracle.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
        oracle.update_Pr(Pr, dP, p.dtau)
        oracle.set_bc_Pr(Pr, 0.0, bool(p.full_rows))


def make_group(p, world, ny_loc, halo, transport=2, devices=None):
    return _lib.Slabs(world, 0, devices or [0] * world, c_params(p), ny_loc, halo, transport=transport)


def load(g, Pr, dP, dv):
    g.upload("Pr", Pr)
    g.upload("dPrdtau", dP)
    g.upload("divV", dv)


def gathered(g, nx, nyg):
    Pr, dP = np.full((nx, nyg), np.nan, order="F"), np.full((nx - 2, nyg - 2), np.nan, order="F")
    g.download("Pr", Pr)
    g.download("dPrdtau", dP)
    return Pr, dP


@pytest.mark.parametrize("edges_first", [1, 0])
@pytest.mark.parametrize("world,ny_loc,halo,count", [(2, 40, 8, 8), (2, 40, 8, 27), (3, 30, 5, 23), (4, 16, 16, 40),
                                                     (3, 40, 12, 60), (5, 30, 14, 31)])
def test_copy_transport_matches_single_domain(gpu, oracle, world, ny_loc, halo, count, edges_first):
    """edges_first (default): the last launch of each block of passes produces the rows to be sent
    first, on a second stream, and skips the halo rows; 0: whole window, then the exchange."""
    nx, nyg = 70, world * ny_loc
    p = ns_params("NavierStokes", nx=nx, ny=nyg)
    Pr0, dP0, dv = fields(np.random.default_rng(world * 100 + count), nx, nyg)
    g = make_group(p, world, ny_loc, halo)
    try:
        g.set_option("edges_first", edges_first)
        with pytest.raises(_lib.NskError, match="unknown option"):
            g.set_option("no_such_switch", 1)
        assert g.info() == {"world": world, "nlocal": world, "transport": 2, "exchanges": 0}
        for i in range(world):
            w = g.window(i)
            lo, hi = (halo if i > 0 else 0), (halo if i < world - 1 else 0)
            assert (w["rank"], w["iy0"], w["ny_win"], w["own_lo"], w["own_hi"]) == \
                   (i, i * ny_loc - lo, ny_loc + lo + hi, lo, lo + ny_loc)
            assert w["ld"] % 2 == 0 and w["ld"] > nx and all(g.fields(i).values())
        load(g, Pr0, dP0, dv)
        back = gathered(g, nx, nyg)                       # scatter + gather is the identity
        assert same_bits(back[0], Pr0) and same_bits(back[1], dP0)
        g.poisson_iters(count)
        g.sync()
        assert g.info()["exchanges"] == -(-count // halo)
        Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
        oracle_passes(oracle, p, Pr, dP, dv, count)
        got = gathered(g, nx, nyg)
        assert same_bits(got[0], Pr), np.argwhere(got[0] != Pr)[:4]
        assert same_bits(got[1], dP), np.argwhere(got[1] != dP)[:4]
        Rp = np.zeros((nx - 2, nyg - 2), order="F")
        oracle.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
        assert g.residual() == oracle.maxabs(Rp)
    finally:
        g.close()


def test_halo_rows_are_refreshed_by_the_exchange(gpu):
    """Stale halo rows (as after a kernel that does not update them) are repaired by nsk_slabs_exchange."""
    world, ny_loc, halo, nx = 3, 20, 6, 33
    p = ns_params("NavierStokes", nx=nx, ny=world * ny_loc)
    Pr0, dP0, dv = fields(np.random.default_rng(3), nx, world * ny_loc)
    g = make_group(p, world, ny_loc, halo)
    try:
        load(g, Pr0, dP0, dv)
        wins = [g.window(i) for i in range(world)]
        for i, w in enumerate(wins):                      # poison the halo rows of Pr directly in device memory
            n = w["ny_win"] * w["ld"]
            host = np.empty(n)
            _lib.load().nsk_download(gpu.handle, host.ctypes.data_as(C.c_void_p), C.c_void_p(g.fields(i)["Pr"]), n)
            rows = host.reshape(w["ny_win"], w["ld"])
            rows[: w["own_lo"]] = 777.0
            rows[w["own_hi"]:] = -777.0
            _lib.load().nsk_upload(gpu.handle, C.c_void_p(g.fields(i)["Pr"]), host.ctypes.data_as(C.c_void_p), n)
        g.exchange(1)
        g.sync()
        for i, w in enumerate(wins):
            n = w["ny_win"] * w["ld"]
            host = np.empty(n)
            _lib.load().nsk_download(gpu.handle, host.ctypes.data_as(C.c_void_p), C.c_void_p(g.fields(i)["Pr"]), n)
            rows = host.reshape(w["ny_win"], w["ld"])[:, :nx]
            assert same_bits(rows.T, Pr0[:, w["iy0"]:w["iy0"] + w["ny_win"]]), i
    finally:
        g.close()


def test_whole_loop_with_checks_on_slabs(gpu, oracle):
    """nsk_slabs_poisson = NavierStokes.jl:84-95 on three slabs: iteration count, (iter, err) history, fields."""
    world, ny_loc, halo = 3, 35, 7
    p = ns_params("NavierStokes", nx=63, ny=world * ny_loc, ly=3.0)
    nx, nyg = p.nx, p.ny
    rng = np.random.default_rng(5)
    Pr0, dP0 = np.zeros((nx, nyg), order="F"), np.zeros((nx - 2, nyg - 2), order="F")
    dv = rng_field(rng, nx, nyg, 0.05)
    Pr, dP, Rp = Pr0.copy(order="F"), dP0.copy(order="F"), np.zeros((nx - 2, nyg - 2), order="F")
    it, ih, eh = oracle.poisson(Pr, dP, Rp, dv, p)
    assert 0 < it < p.niter, "test wants an early exit"
    g = make_group(p, world, ny_loc, halo)
    try:
        load(g, Pr0, dP0, dv)
        assert g.poisson() == (it, ih, eh)
        got = gathered(g, nx, nyg)
        assert same_bits(got[0], Pr) and same_bits(got[1], dP)
    finally:
        g.close()


def test_slabs_of_the_benchmark_width(gpu, oracle):
    """2559 cells wide (23 column strips, the vectorised kernel), three slabs of 400 rows, 30 passes, 3 exchanges."""
    world, ny_loc, halo, count = 3, 400, 12, 30
    p1 = ns_params("NavierStokes_Re6", n=16)
    p = p1.with_(ny=world * ny_loc)
    nx, nyg = p.nx, p.ny
    Pr0, dP0, dv = fields(np.random.default_rng(9), nx, nyg, 1e-2)
    g = make_group(p, world, ny_loc, halo)
    try:
        load(g, Pr0, dP0, dv)
        g.poisson_iters(count)
        Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
        oracle_passes(oracle, p, Pr, dP, dv, count)
        got = gathered(g, nx, nyg)
        assert same_bits(got[0], Pr) and same_bits(got[1], dP)
        assert g.info()["exchanges"] == 3
    finally:
        g.close()


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("world,halo,nx,ny_loc,transport", [(2, 8, 45, 40, 2), (3, 12, 54, 30, 2), (1, 0, 45, 60, 1), (4, 6, 45, 20, 2)])
def test_whole_time_steps_on_slabs(gpu, world, halo, nx, ny_loc, transport, fused):
    """nsk_slabs_step = NavierStokes.jl:79-100 on slabs (windows of every array, exchanges of divV, Pr,
    dPrdtau, Vx, C and the face field Vy, the reach check of advect!): three steps from the reference's
    initial condition against the single-domain oracle; with one slab the RCCL group does the same.  Around
    the Poisson loop either the three fused kernels on windows (the default) or one launch per kernel."""
    from navierstokes_amd.ns import init_vprof
    from oracle.ns_oracle import OracleSim
    nyg = world * ny_loc
    p = ns_params("NavierStokes", nx=nx, ny=nyg, niter=400, nchk=100)
    g = _lib.Slabs(world, 0, [0] * world, c_params(p), ny_loc, halo, transport=transport,
                   unique_id=_lib.rccl_unique_id() if transport == 1 else None)
    ref = OracleSim(p)
    try:
        g.set_option("step_fused", fused)
        assert g.get_option("step_fused") == fused
        g.init_flow(init_vprof(p))
        for _ in range(3):
            it = ref.step()
            assert g.step() == (it, ref.iter_evo, ref.err_evo)
        shapes = {"Pr": (nx, nyg), "Vx": (nx + 1, nyg), "Vy": (nx, nyg + 1), "C": (nx, nyg), "dPrdtau": (nx - 2, nyg - 2)}
        for k, shp in shapes.items():
            out = np.full(shp, np.nan, order="F")
            g.download(k, out)
            assert same_bits(out, getattr(ref, k)), (k, np.argwhere(out != getattr(ref, k))[:4])
    finally:
        g.close()


def test_slab_time_step_rejects_a_halo_the_back_trace_would_leave(gpu):
    from navierstokes_amd.ns import init_vprof
    p = ns_params("NavierStokes", nx=33, ny=60, niter=20, nchk=10)
    g = make_group(p, 2, 30, 4)
    try:
        with pytest.raises(_lib.NskError, match="init_flow first"):
            g.step()
        g.init_flow(init_vprof(p))
        Vy = np.zeros((33, 61), order="F")
        Vy[:, 20:40] = 3.0 * p.dy / p.dt              # three rows per step: reach 5 > halo - 1 = 3
        g.upload("Vy", Vy)
        with pytest.raises(_lib.NskError, match="raise the halo width"):
            g.step()
    finally:
        g.close()
    with pytest.raises(_lib.NskError, match="at least 4 halo rows"):
        g = make_group(p, 2, 30, 3)
        try:
            g.init_flow(init_vprof(p))
        finally:
            g.close()


@pytest.mark.parametrize("edges_first", [1, 0])
def test_a_transfer_that_does_not_complete_ends_in_an_error(gpu, oracle, edges_first):
    """The host-side waits of a group have a deadline (first_contact_ms until the first exchange has been
    waited for successfully, deadline_ms afterwards).  An exchange held back on the device for 3 s (a
    bounded idle kernel where the transfer is queued) against a deadline of 0.3 s: the residual returns
    NSK_ERR_TIMEOUT within the deadline and says which slab waited for what; nothing was aborted on the
    device, so once the idle kernel has ended the same group delivers the right values."""
    import time
    world, ny_loc, halo, nx = 3, 40, 8, 70
    nyg = world * ny_loc
    p = ns_params("NavierStokes", nx=nx, ny=nyg)
    Pr0, dP0, dv = fields(np.random.default_rng(5), nx, nyg)
    g = make_group(p, world, ny_loc, halo)
    try:
        g.set_option("edges_first", edges_first)
        g.set_option("first_contact_ms", 300)
        g.set_option("deadline_ms", 300)
        load(g, Pr0, dP0, dv)
        g.set_option("test_stall_ms", 3000)
        t0 = time.perf_counter()
        g.poisson_iters(halo)                       # queues the passes and the (held back) exchange
        with pytest.raises(_lib.NskError, match="no completion within 300 ms on slab .* of 3 .*1 exchanges queued") as e:
            g.residual()
        waited = time.perf_counter() - t0
        assert e.value.code == _lib.ERR_TIMEOUT and 0.25 < waited < 2.5, waited   # (well before the idle kernel ends)
        assert "device copies" in str(e.value) and ("edges-first" if edges_first else "plain") in str(e.value)
        time.sleep(3.0)                             # the idle kernel ends by itself
        g.set_option("deadline_ms", 20000)
        g.set_option("first_contact_ms", 20000)
        m = g.residual()
        Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
        oracle_passes(oracle, p, Pr, dP, dv, halo)
        gPr, gdP = gathered(g, nx, nyg)
        assert same_bits(gPr, Pr) and same_bits(gdP, dP)
        Rp = np.zeros((nx - 2, nyg - 2), order="F")
        oracle.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
        assert m == oracle.maxabs(Rp)
    finally:
        g.close()


def test_rccl_transport_on_one_slab(gpu, oracle):
    """RCCL path as far as one GPU can take it: library loaded at run time, unique id, communicator of
    one rank, ncclAllReduce(max) inside the residual, and ncclSend/ncclRecv of row blocks to itself on
    the communication stream."""
    uid = _lib.rccl_unique_id()
    assert len(uid) == 128 and any(uid)
    nx, ny = 70, 64
    p = ns_params("NavierStokes", nx=nx, ny=ny, niter=60, nchk=20)
    Pr0, dP0, dv = fields(np.random.default_rng(12), nx, ny)
    g = _lib.Slabs(1, 0, [0], c_params(p), ny, 0, transport=1, unique_id=uid)
    try:
        assert g.info()["transport"] == 1
        load(g, Pr0, dP0, dv)
        Pr, dP, Rp = Pr0.copy(order="F"), dP0.copy(order="F"), np.zeros((nx - 2, ny - 2), order="F")
        it, ih, eh = oracle.poisson(Pr, dP, Rp, dv, p)
        assert g.poisson() == (it, ih, eh)                 # every check went through the all-reduce
        got = gathered(g, nx, ny)
        assert same_bits(got[0], Pr) and same_bits(got[1], dP)
        g.call("nsk_slabs_rccl_self_exchange", 9)          # first 9 owned rows of Pr -> divV, through RCCL
        g.sync()
        out = np.full((nx, ny), np.nan, order="F")
        g.download("divV", out)
        assert same_bits(out[:, :9], Pr[:, :9]) and same_bits(out[:, 9:], dv[:, 9:])
    finally:
        g.close()


def test_slab_group_misuse_is_reported(gpu):
    p = ns_params("NavierStokes", nx=33, ny=60)
    cp = c_params(p)
    with pytest.raises(_lib.NskError, match="slabs of"):
        _lib.Slabs(3, 0, [0, 0, 0], cp, 25, 4)             # 3 x 25 != 60 rows
    with pytest.raises(_lib.NskError, match="halo"):
        _lib.Slabs(3, 0, [0, 0, 0], cp, 20, 0)
    with pytest.raises(_lib.NskError, match="every slab in this process"):
        _lib.Slabs(3, 0, [0], cp, 20, 4, transport=2)
    with pytest.raises(_lib.NskError, match="id of nsk_rccl_unique_id"):
        _lib.Slabs(3, 1, [0], cp, 20, 4, transport=1)
    with pytest.raises(_lib.NskError, match="out of range"):
        _lib.Slabs(3, 0, [0, 0, 4096], cp, 20, 4)
    g = _lib.Slabs(3, 0, [0, 0, 0], cp, 20, 4)
    try:
        with pytest.raises(_lib.NskError, match="local slab index"):
            g.window(3)
        with pytest.raises(_lib.NskError, match="does not use RCCL"):
            g.call("nsk_slabs_rccl_self_exchange", 2)
    finally:
        g.close()
