"""The kit that ties this repository to the reference for scripts/NavierStokes.jl:115-250:
julia/make_fixture.jl runs the reference's own kernels and writes fixture directories,
tools/check_reference_fixture.py compares them with the HIP library (or the oracle).  No Julia exists
here, so the reference side cannot run; these tests keep the Python side honest: a fixture of the same
format written from the oracle must read back and compare bit-identical with the oracle (CPU) and with
the HIP library through the C ABI (GPU), and the tool must notice a planted difference."""
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_reference_fixture as kit  # noqa: E402


@pytest.fixture(scope="module")
def fixture_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("fx"))
    kit.write_from_oracle(d)
    return d


def test_fixture_files_round_trip(tmp_path):
    a = {"f": np.asfortranarray(np.arange(12.0).reshape(3, 4)), "i": np.array([[3], [1], [2]], dtype=np.int64),
         "s": np.array([0.1])}
    kit.write_fixture(str(tmp_path), a)
    b = kit.read_fixture(str(tmp_path))
    assert np.array_equal(b["f"], a["f"]) and b["f"].shape == (3, 4)
    assert np.array_equal(b["i"], a["i"]) and b["i"].dtype == np.int64
    assert b["s"][0, 0] == 0.1


def test_pseudo_data_is_exactly_representable():
    p = kit.pseudo(40, 23, 7)
    assert np.array_equal(p * 1024.0, np.round(p * 1024.0)) and p.min() >= -1.0 and p.max() < 1.0
    assert len(np.unique(p)) > 300


def test_checker_passes_on_the_oracle_and_sees_a_planted_difference(fixture_dir, capsys):
    assert kit.main([fixture_dir, "--backend", "oracle"]) == 0
    out = capsys.readouterr().out
    assert "rows 1..size(A,1) -> full_rows = 0" in out and "=> bit-identical with full_rows in [0]" in out
    # one bit of one pressure value of the small time-step recipe
    path = os.path.join(fixture_dir, "steps_21x37_3", "Pr.bin")
    raw = bytearray(open(path, "rb").read())
    raw[8 * 100] ^= 1
    open(path, "wb").write(raw)
    try:
        assert kit.main([fixture_dir, "--backend", "oracle"]) == 1
        assert "NOT bit-identical with either setting" in capsys.readouterr().out
    finally:
        raw[8 * 100] ^= 1
        open(path, "wb").write(raw)


def test_julia_side_names_what_the_checker_reads():
    """Every array the checker's recipes produce is dumped under the same name by julia/make_fixture.jl."""
    src = open(os.path.join(ROOT, "julia", "make_fixture.jl"), encoding="utf-8").read()
    dumped = set(re.findall(r'dump!\(d, "([A-Za-z_0-9]+)"', src))
    K = kit.OracleKernels()
    want = set(kit.kernel_recipe(K, 9, 12, False)) | {"bc_x_rows_"}
    steps = set(kit.steps_recipe(K, 9, 12, 1, False)) - {k for k in kit.steps_recipe(K, 9, 12, 1, False) if k.startswith("scalar_")}
    assert want - {"bc_x_rows_"} <= dumped, want - dumped
    assert steps <= dumped, steps - dumped
    assert 'startswith(l, "macro ∇V")' in src and "@init_parallel_stencil(Threads, Float64, 2)" in src


@pytest.mark.gpu
def test_checker_passes_on_the_hip_library(fixture_dir, capsys):
    assert kit.main([fixture_dir, "--backend", "hip"]) == 0
    out = capsys.readouterr().out
    assert out.count("=> bit-identical with full_rows in [0]") == 2 and "hip backend" in out
