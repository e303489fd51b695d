"""The two headline configurations at their own sizes.

* scripts/NavierStokes_Re6.jl, 2559 x 4095 on one GPU: one WHOLE time step from the script's
  initial condition -- the Poisson loop runs until the reference's own stopping rule ends it
  (NavierStokes_Re6.jl:84-95: tens of thousands of passes, a residual check every 4(nx-1)) --
  against the oracle's OpenMP build: pass count, (iter, err) history and every field.
* The same problem weak-scaled to 8 slabs (2559 x 32760 cells): eight ranks, here sharing the one
  GPU under gloo, run passes with three halo exchanges and the all-reduced residual check against
  the single-domain oracle on the whole grid.

Both take minutes (the CPU oracle, not the GPU), hence the `slow` mark; they print a line now and
then so that a watching harness sees progress."""
import os
import socket
import threading
import time

import numpy as np
import pytest
import torch

from navierstokes_amd import ns_params
from util import same_bits

pytestmark = [pytest.mark.gpu, pytest.mark.slow, pytest.mark.timeout(1500)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def say(capsys, msg):
    with capsys.disabled():
        print(f"\n    [{time.strftime('%H:%M:%S')}] {msg}", end="", flush=True)
    try:    # a harness that watches files rather than the captured stdout
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "headline_progress.log"), "a") as f:
            f.write(f"{time.strftime('%H:%M:%S')} {msg}\n")
    except OSError:
        pass


def in_thread_with_progress(capsys, what, fn):
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("v", fn()))
    t0 = time.perf_counter()
    th.start()
    while th.is_alive():
        th.join(25.0)
        say(capsys, f"{what}: {time.perf_counter() - t0:.0f} s")
    return res["v"]


def test_whole_time_step_at_2559x4095(capsys):
    from navierstokes_amd.ns import NavierStokes
    from oracle.ns_oracle import OracleSim
    p = ns_params("NavierStokes_Re6", n=16)
    assert (p.nx, p.ny, p.niter, p.nchk) == (2559, 4095, 127950, 10232)
    sim = NavierStokes(p)
    t0 = time.perf_counter()
    it_gpu = sim.step()
    torch.cuda.synchronize()
    say(capsys, f"GPU: {it_gpu} passes, {len(sim.err_evo)} checks in {time.perf_counter() - t0:.2f} s")
    ref = OracleSim(p, omp=True)
    it_ref = in_thread_with_progress(capsys, "oracle time step", ref.step)
    assert it_gpu == it_ref and it_gpu >= 5 * p.nchk            # the loop ran to its stopping rule
    assert sim.iter_evo == ref.iter_evo and sim.err_evo == ref.err_evo
    assert ref.err_evo[-1] < p.eps_it <= ref.err_evo[-2]
    got = sim.host_fields()
    for k in ("Pr", "Vx", "Vy", "C"):
        assert same_bits(got[k], getattr(ref, k)), k
    assert same_bits(sim.dPrdtau_host(), ref.dPrdtau)


# ---- eight slabs of 2559 x 4095 ---------------------------------------------------------------------
NX, NY_LOC, WORLD, HALO, PASSES = 2559, 4095, 8, 12, 30


def block(field: int, b: int) -> np.ndarray:
    """Rows b*NY_LOC .. (b+1)*NY_LOC of global field `field` (0 Pr, 1 dPrdtau on the Pr layout, 2 divV),
    shaped (rows, NX): every rank and the checker can generate any part of the global fields."""
    rng = np.random.default_rng(1000 * field + b)
    return rng.standard_normal((NY_LOC, NX)) * 1e-2


def global_rows(field: int, lo: int, hi: int) -> np.ndarray:
    parts = []
    for b in range(lo // NY_LOC, (hi - 1) // NY_LOC + 1):
        a0, a1 = max(lo, b * NY_LOC), min(hi, (b + 1) * NY_LOC)
        parts.append(block(field, b)[a0 - b * NY_LOC:a1 - b * NY_LOC])
    return np.concatenate(parts)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _slab_worker(rank, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        from navierstokes_amd import _lib
        from navierstokes_amd.dist import HipBackend, SlabPoisson, make_slab
        from navierstokes_amd.ns import _pitch
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        p1 = ns_params("NavierStokes_Re6", n=16)
        p = p1.with_(ny=NY_LOC * WORLD, ly=p1.ly * WORLD)
        slab = make_slab(rank, WORLD, NY_LOC, HALO)
        ld = _pitch(NX)
        lo, hi = slab.iy0, slab.iy0 + slab.ny_win

        def win(field):
            t = torch.zeros((slab.ny_win, ld), dtype=torch.float64)
            t[:, :NX] = torch.from_numpy(global_rows(field, lo, hi))
            return t.to(dev)
        Pr, dP, divV = win(0), win(1), win(2)
        ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
        sp = SlabPoisson(p, slab, HipBackend(ctx, p, ld), Pr, dP, divV)
        sp.iterate(PASSES)
        assert sp.exchanges == 3
        res = sp.residual()
        torch.cuda.synchronize()
        own = slice(slab.own_lo, slab.own_hi)
        np.save(os.path.join(out, f"Pr{rank}.npy"), Pr[own, :NX].cpu().numpy())
        np.save(os.path.join(out, f"dP{rank}.npy"), dP[own, :NX].cpu().numpy())
        np.save(os.path.join(out, f"res{rank}.npy"), np.array([res]))
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_eight_slabs_of_2559x4095(tmp_path, capsys):
    import torch.multiprocessing as mp
    from oracle.ns_oracle import Oracle
    nyg = NY_LOC * WORLD
    assert (NX, nyg) == (2559, 32760)
    say(capsys, "8 ranks on cuda:0 under gloo: 30 passes, 3 exchanges, residual")
    mp.spawn(_slab_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    say(capsys, "ranks done; single-domain oracle on 2559 x 32760")
    o = Oracle(omp=True)
    p1 = ns_params("NavierStokes_Re6", n=16)
    p = p1.with_(ny=nyg, ly=p1.ly * WORLD)
    Pr = np.asfortranarray(global_rows(0, 0, nyg).T)
    dv = np.asfortranarray(global_rows(2, 0, nyg).T)
    dP = np.asfortranarray(global_rows(1, 0, nyg).T[1:-1, 1:-1])

    def passes():
        for _ in range(PASSES):
            o.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
            o.update_Pr(Pr, dP, p.dtau)
            o.set_bc_Pr(Pr, 0.0, False)
        Rp = np.zeros((NX - 2, nyg - 2), order="F")
        o.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
        return o.maxabs(Rp)
    want_res = in_thread_with_progress(capsys, "oracle", passes)
    for r in range(WORLD):
        rows = slice(r * NY_LOC, (r + 1) * NY_LOC)
        got = np.load(os.path.join(tmp_path, f"Pr{r}.npy"))
        assert same_bits(got.T, Pr[:, rows]), f"Pr of rank {r}"
        got = np.load(os.path.join(tmp_path, f"dP{r}.npy")).T          # on the Pr layout: rim unused
        a, b = max(rows.start, 1), min(rows.stop, nyg - 1)
        assert same_bits(got[1:-1, a - rows.start:b - rows.start], dP[:, a - 1:b - 1]), f"dPrdtau of rank {r}"
        assert np.load(os.path.join(tmp_path, f"res{r}.npy"))[0] == want_res, f"residual on rank {r}"


def test_eight_slabs_of_2559x4095_behind_the_c_abi(capsys):
    """The same grid through nsk_slabs_* the way a single-threaded driver (Julia's ccall) holds a node:
    one group with all eight slabs, the reference's global arrays scattered by nsk_slabs_upload, halo
    rows moved by peer copies between the slabs' streams (here all on cuda:0), 30 passes = 3 exchanges
    with the edges-first schedule, the residual, and the gathered arrays against the single-domain
    oracle."""
    from navierstokes_amd import _lib
    from navierstokes_amd.ns import c_params
    from oracle.ns_oracle import Oracle
    nyg = NY_LOC * WORLD
    p1 = ns_params("NavierStokes_Re6", n=16)
    p = p1.with_(ny=nyg, ly=p1.ly * WORLD)
    Pr = np.asfortranarray(global_rows(0, 0, nyg).T)
    dv = np.asfortranarray(global_rows(2, 0, nyg).T)
    dP = np.asfortranarray(global_rows(1, 0, nyg).T[1:-1, 1:-1])
    say(capsys, "8 slabs in one group on cuda:0: upload, 30 passes, residual, download")
    g = _lib.Slabs(WORLD, 0, [0] * WORLD, c_params(p), NY_LOC, HALO, transport=2)
    try:
        for name, a in (("Pr", Pr), ("dPrdtau", dP), ("divV", dv)):
            g.upload(name, a)
        g.poisson_iters(PASSES)
        got_res = g.residual()
        assert g.info()["exchanges"] == 3
        got_Pr, got_dP = np.full((NX, nyg), np.nan, order="F"), np.full((NX - 2, nyg - 2), np.nan, order="F")
        g.download("Pr", got_Pr)
        g.download("dPrdtau", got_dP)
    finally:
        g.close()
    o = Oracle(omp=True)

    def passes():
        for _ in range(PASSES):
            o.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
            o.update_Pr(Pr, dP, p.dtau)
            o.set_bc_Pr(Pr, 0.0, False)
        Rp = np.zeros((NX - 2, nyg - 2), order="F")
        o.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
        return o.maxabs(Rp)
    want_res = in_thread_with_progress(capsys, "oracle", passes)
    assert same_bits(got_Pr, Pr) and same_bits(got_dP, dP)
    assert got_res == want_res


def test_time_steps_on_eight_slabs_behind_the_c_abi(capsys):
    """nsk_slabs_init_flow + nsk_slabs_step (NavierStokes_Re6.jl:72-100 on windows of all thirteen arrays)
    at the size of the weak-scaled configuration: two time steps from the script's initial condition
    with the Poisson loop cut to 120 passes (two residual checks) so that the single-domain oracle
    on 2559 x 32760 finishes in seconds; pass counts, residual history and every field."""
    from navierstokes_amd import _lib
    from navierstokes_amd.ns import c_params, init_vprof
    from oracle.ns_oracle import OracleSim
    nyg = NY_LOC * WORLD
    p1 = ns_params("NavierStokes_Re6", n=16)
    p = p1.with_(ny=nyg, ly=p1.ly * WORLD, niter=120, nchk=60)
    say(capsys, "8 slabs in one group on cuda:0: two time steps of 120 passes")
    g = _lib.Slabs(WORLD, 0, [0] * WORLD, c_params(p), NY_LOC, HALO, transport=2)
    try:
        g.init_flow(init_vprof(p))
        got = [g.step() for _ in range(2)]
        shapes = {"Pr": (NX, nyg), "Vx": (NX + 1, nyg), "Vy": (NX, nyg + 1), "C": (NX, nyg), "dPrdtau": (NX - 2, nyg - 2)}
        fields = {}
        for k, shp in shapes.items():
            fields[k] = np.full(shp, np.nan, order="F")
            g.download(k, fields[k])
    finally:
        g.close()
    ref = OracleSim(p, omp=True)

    def steps():
        out = []
        for _ in range(2):
            it = ref.step()
            out.append((it, list(ref.iter_evo), list(ref.err_evo)))
        return out
    want = in_thread_with_progress(capsys, "oracle time steps", steps)
    assert [(a, list(b), list(c)) for a, b, c in got] == want
    assert want[-1][0] == 120 and len(want[-1][2]) == 2
    for k in ("Pr", "Vx", "Vy", "C"):
        assert same_bits(fields[k], getattr(ref, k)), k
    assert same_bits(fields["dPrdtau"], ref.dPrdtau)
