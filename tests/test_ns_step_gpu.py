"""Whole time steps (scripts/NavierStokes.jl:77-100) through the C ABI against the oracle:
iteration counts, the (iter, err) history the reference prints at :92, and the saved fields.
North-star bar: rel-L2 <= 1e-10 on Vx, Vy, Pr after N steps; the implementation is held to
bit equality (the tolerance is asserted as well so the bar is explicit)."""
import numpy as np
import pytest

from navierstokes_amd import ns_params
from navierstokes_amd.ns import NavierStokes
from oracle.ns_oracle import OracleSim
from util import rel_l2, same_bits

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("mode", ["fused", "granular"])
def test_steps_small_grid(mode):
    p = ns_params("NavierStokes", nx=45, ny=75)
    if mode == "granular":
        p = p.with_(niter=600)    # one launch per kernel and pass: keep it short
    sim, ref = NavierStokes(p, mode=mode), OracleSim(p)
    for _ in range(3):
        assert sim.step() == ref.step()
        assert sim.iter_evo == ref.iter_evo and sim.err_evo == ref.err_evo
    got = sim.host_fields()
    for k in ("Pr", "Vx", "Vy", "C"):
        assert rel_l2(got[k], getattr(ref, k)) <= TOL, k
        assert same_bits(got[k], getattr(ref, k)), k
    assert same_bits(sim.dPrdtau_host(), ref.dPrdtau)


def test_steps_reference_default_grid():
    """scripts/NavierStokes.jl as shipped: 153 x 255, Re = 1e4, first 3 of its 2000 steps."""
    p = ns_params("NavierStokes")
    sim, ref = NavierStokes(p), OracleSim(p)
    for _ in range(3):
        assert sim.step() == ref.step()
        assert sim.err_evo == ref.err_evo
    got = sim.host_fields()
    for k in ("Pr", "Vx", "Vy", "C"):
        assert rel_l2(got[k], getattr(ref, k)) <= TOL, k
        assert same_bits(got[k], getattr(ref, k)), k


def test_full_rows_switch():
    """bc_x! over all rows (the presumably intended launch range) instead of the reference's."""
    p = ns_params("NavierStokes", nx=45, ny=75, full_rows=True)
    sim, ref = NavierStokes(p), OracleSim(p)
    for _ in range(2):
        assert sim.step() == ref.step()
    got = sim.host_fields()
    for k in ("Pr", "Vx", "Vy", "C"):
        assert same_bits(got[k], getattr(ref, k)), k


def test_restart_file_continues_the_same_run(tmp_path):
    """2 + 2 steps through a restart file (which carries dPrdτ, unlike the reference's .mat
    output) equal 4 steps in one go, bit for bit; a file for another grid is refused."""
    from navierstokes_amd.ns import runme
    kw = dict(script="NavierStokes", nx=47, ny=79, verbose=False, out_dir=str(tmp_path))
    whole = runme(nt=4, **kw)
    runme(nt=2, restart_every=2, **kw)
    cont = runme(nt=4, restart_from=str(tmp_path / "restart.npz"), **kw)
    assert cont.it == 4
    a, b = whole.host_fields(), cont.host_fields()
    for k in a:
        assert same_bits(a[k], b[k]), k
    assert same_bits(whole.dPrdtau_host(), cont.dPrdtau_host())
    other = NavierStokes(ns_params("NavierStokes", nx=31, ny=79))
    with pytest.raises(ValueError):
        other.load_restart(str(tmp_path / "restart.npz"))


def test_steps_benchmark_grid():
    """Two whole time steps on the benchmark grid (2559 x 4095, Re = 1e6 parameters) with the
    Poisson loop cut to 60 passes (check every 30) so the oracle finishes in seconds."""
    p = ns_params("NavierStokes_Re6", n=16).with_(niter=60, nchk=30)
    sim, ref = NavierStokes(p), OracleSim(p)
    for _ in range(2):
        assert sim.step() == ref.step()
        assert sim.iter_evo == ref.iter_evo and sim.err_evo == ref.err_evo
    got = sim.host_fields()
    for k in ("Pr", "Vx", "Vy", "C"):
        assert rel_l2(got[k], getattr(ref, k)) <= TOL, k
        assert same_bits(got[k], getattr(ref, k)), k
    assert same_bits(sim.dPrdtau_host(), ref.dPrdtau)
