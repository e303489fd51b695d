"""The C ABI from a caller that is neither Python nor Julia: examples/abi_client.cpp (plain C++ on
include/nsk.h, linked against libnsk.so, built by `make -C examples`) reads the reference's dense arrays
and the parameter block from a file, runs Poisson passes (scripts/NavierStokes.jl:85-87) and the residual
(:89-90) and writes the result; compared here with the oracle bit for bit."""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from navierstokes_amd import ns_params
from navierstokes_amd.ns import c_params
from test_poisson_gpu import oracle_iters
from util import rng_field, same_bits

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLIENT = os.path.join(ROOT, "examples", "abi_client")


@pytest.mark.parametrize("nx,ny,count", [(153, 255, 40), (153, 255, 7), (300, 97, 13)])
def test_cxx_caller_of_the_abi(oracle, tmp_path, nx, ny, count):
    if not os.path.exists(CLIENT):       # built by __graft_entry__.build(); a one-second host compile otherwise
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    p = ns_params("NavierStokes", nx=nx, ny=ny, full_rows=nx > ny)
    rng = np.random.default_rng(nx + count)
    Pr0, dP0, dv = rng_field(rng, nx, ny), rng_field(rng, nx - 2, ny - 2), rng_field(rng, nx, ny, 10.0)
    cp = c_params(p)
    problem, result = tmp_path / "problem.bin", tmp_path / "result.bin"
    with open(problem, "wb") as f:
        f.write(struct.pack("<ii", 0x4E534B31, C.sizeof(cp)))
        f.write(bytes(cp))
        for a in (Pr0, dP0, dv):
            f.write(np.asfortranarray(a).tobytes(order="F"))
    out = subprocess.run([CLIENT, str(problem), str(result), str(count), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    raw = np.fromfile(result, dtype=np.float64)
    gPr = raw[: nx * ny].reshape((nx, ny), order="F")
    gdP = raw[nx * ny: nx * ny + (nx - 2) * (ny - 2)].reshape((nx - 2, ny - 2), order="F")
    Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
    oracle_iters(oracle, p, Pr, dP, dv, count, int(p.full_rows))
    Rp = np.zeros((nx - 2, ny - 2), order="F")
    oracle.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
    assert same_bits(gPr, Pr) and same_bits(gdP, dP)
    assert raw[-1] == np.abs(Rp).max() == line["max_abs_residual"]
    assert line["passes"] == count and line["us_per_pass"] > 0
    if count >= 12 and nx * ny < 100000:
        assert line["passes_per_launch"] == count        # the run of passes was one launch (the fields fit on chip)


def test_cxx_caller_reports_errors(tmp_path):
    if not os.path.exists(CLIENT):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples")])
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"\\0" * 64)
    out = subprocess.run([CLIENT, str(bad), str(tmp_path / "r.bin"), "3"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "not a problem file" in out.stderr
