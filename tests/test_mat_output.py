"""The on-disk contract downstream of the path: `out_vis/step_N.mat` with the keys
Pr, Vx, Vy, C, dx, dy (scripts/NavierStokes.jl:75,108-110), read by vis/vis_navier_stokes.m:6-14
with MATLAB's `load`.  The reference writes through MAT.jl (a v7.3/HDF5 container); this build
writes MAT v5, which `load` reads the same way.  What is pinned here: key names, array
orientation (size(A,1) x size(A,2), not transposed), Float64, scalar dx/dy, the file names and
the steps at which files appear."""
import os

import numpy as np
import pytest
from scipy.io import loadmat, whosmat

from navierstokes_amd import ns_params
from navierstokes_amd.ns import save_mat

KEYS = {"Pr", "Vx", "Vy", "C", "dx", "dy"}


def check_file(path, nx, ny, dx, dy):
    """What vis_navier_stokes.m needs from a file; returns the arrays."""
    info = {name: (shape, cls) for name, shape, cls in whosmat(path)}
    assert set(info) == KEYS
    assert info["Pr"] == ((nx, ny), "double") and info["C"] == ((nx, ny), "double")
    assert info["Vx"] == ((nx + 1, ny), "double") and info["Vy"] == ((nx, ny + 1), "double")
    assert info["dx"] == ((1, 1), "double") and info["dy"] == ((1, 1), "double")
    m = loadmat(path)
    assert m["dx"][0, 0] == dx and m["dy"][0, 0] == dy
    for k in ("Pr", "Vx", "Vy", "C"):
        assert m[k].dtype == np.float64
    # the expressions of vis/vis_navier_stokes.m:10-13 must be well formed on these shapes
    om = np.diff(m["Vy"][:, 1:-1], axis=0) / m["dx"] - np.diff(m["Vx"][1:-1, :], axis=1) / m["dy"]
    assert om.shape == (nx - 1, ny - 1)
    vxc = 0.5 * (m["Vx"][:-1, :] + m["Vx"][1:, :])
    vyc = 0.5 * (m["Vy"][:, :-1] + m["Vy"][:, 1:])
    assert np.sqrt(vxc ** 2 + vyc ** 2).shape == m["Pr"].shape == m["C"].shape
    return m


def test_save_mat_round_trip(tmp_path):
    nx, ny = 7, 11
    rng = np.random.default_rng(0)
    f = {"Pr": np.asfortranarray(rng.standard_normal((nx, ny))),
         "Vx": np.asfortranarray(rng.standard_normal((nx + 1, ny))),
         "Vy": np.asfortranarray(rng.standard_normal((nx, ny + 1))),
         "C": np.asfortranarray(rng.standard_normal((nx, ny)))}
    path = str(tmp_path / "out_vis" / "step_0.mat")      # the directory is created on demand (:75)
    save_mat(path, f, 0.125, 0.0625)
    m = check_file(path, nx, ny, 0.125, 0.0625)
    for k, v in f.items():
        assert np.array_equal(m[k], v), k                 # element [ix, iy] stays element [ix, iy]


@pytest.mark.gpu
def test_runme_writes_the_reference_files(tmp_path, gpu):
    """runme(do_save=true): step_0.mat before the loop, step_N.mat whenever N % nsave == 0, holding
    what the oracle's run holds at those steps."""
    from navierstokes_amd.ns import runme
    from oracle.ns_oracle import OracleSim
    out = str(tmp_path / "out_vis")
    over = dict(nx=21, ny=37, nsave=2)
    sim = runme(do_vis=False, do_save=True, script="NavierStokes", nt=5, out_dir=out, verbose=False, **over)
    assert sorted(os.listdir(out)) == ["step_0.mat", "step_2.mat", "step_4.mat"]
    p = ns_params("NavierStokes", **over)
    assert sim.p == p
    ref = OracleSim(p)
    for it in range(0, 5):
        if it % p.nsave == 0:
            m = check_file(os.path.join(out, f"step_{it}.mat"), p.nx, p.ny, p.dx, p.dy)
            for k in ("Pr", "Vx", "Vy", "C"):
                assert np.array_equal(m[k], getattr(ref, k)), (it, k)
        ref.step()


@pytest.mark.gpu
def test_runme_without_saving_writes_nothing(tmp_path, gpu):
    from navierstokes_amd.ns import runme
    out = str(tmp_path / "out_vis")
    runme(do_vis=False, do_save=False, script="NavierStokes", nt=1, out_dir=out, verbose=False, nx=9, ny=15)
    assert not os.path.exists(out)
