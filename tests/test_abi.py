"""The C-ABI library loads without a GPU, exports every symbol include/nsk.h declares, and the
Python binding's structs match the header.  No compute call is made here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    names = []
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            text = open(os.path.join(ROOT, "include", h)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            names += re.findall(r"\b(nsk_[a-z0-9_A-Z]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported():
    from navierstokes_amd import _lib
    lib = _lib.load()
    names = declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # ... and the binding knows every one of them
    unbound = [n for n in names if n not in _lib.exported_names()]
    assert not unbound, unbound


def test_struct_layout_matches_header():
    from navierstokes_amd import _lib
    text = open(os.path.join(ROOT, "include", "nsk.h")).read()
    body = re.search(r"typedef struct nsk_ns_params \{(.*?)\} nsk_ns_params;", text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            ty, rest = decl.split(None, 1)
            fields += [(n.strip(), ty) for n in rest.split(",")]
    want = [(n, {"int": C.c_int, "double": C.c_double}[t]) for n, t in fields]
    assert [(n, t) for n, t in _lib.CNSParams._fields_] == want
    body = re.search(r"typedef struct nsk_ns_state \{(.*?)\} nsk_ns_state;", text, re.S).group(1)
    names = [n.strip(" *\n") for n in body.replace("double", "").replace(";", "").split(",")]
    assert [n for n, _ in _lib.CNSState._fields_] == names


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("this box has a GPU")
    from navierstokes_amd import _lib
    from navierstokes_amd.ns import NavierStokes
    from navierstokes_amd import ns_params
    with pytest.raises(_lib.NskError):
        _lib.Context(0)
    with pytest.raises(_lib.NskError):
        NavierStokes(ns_params("NavierStokes", nx=9, ny=15))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "navierstokes_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("oracle's", ""), (dirpath, f)


def test_header_is_plain_c(tmp_path):
    """include/nsk.h is the boundary for foreign-language callers: it must compile as C99 and
    as C++ with nothing but the standard headers."""
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "nsk.h")
    src = tmp_path / "h.c"
    src.write_text(f'#include "{hdr}"\nint main(void) {{ nsk_ns_params p; p.nx = 3; return p.nx - 3; }}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", str(src)], check=True)
    cpp = tmp_path / "h.cpp"
    cpp.write_text(f'#include "{hdr}"\nint main() {{ nsk_ns_state s{{}}; return s.Pr != nullptr; }}\n')
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", str(cpp)], check=True)


def test_launch_planner_host_logic():
    """Host side of the hot path that needs no GPU: how a launch of the multi-pass kernel is cut
    into (strip, chunk) tasks.  All tasks of a launch are resident at once (<= 2 per SIMD); the
    benchmark grid gets 5 or 6 passes per launch, small grids fewer passes and short tasks; forced
    values are honoured; never more passes than asked for."""
    from navierstokes_amd import _lib
    big = _lib.poisson_plan(2559, 4095)
    assert big["k"] in (5, 6) and big["tasks"] <= 2048 and big["strips"] == 23
    assert big["rows_edge"] < big["rows"] and 2 <= big["rows"] <= 64
    assert 15000 < big["est_ns_per_pass"] < 30000
    small = _lib.poisson_plan(159, 255)
    assert small["k"] <= 4 and small["rows"] <= 8 and small["strips"] == 2
    assert _lib.poisson_plan(2559, 4095, k_limit=3)["k"] <= 3
    assert _lib.poisson_plan(2559, 4095, staged=False)["k"] <= 5
    forced = _lib.poisson_plan(2559, 4095, k=8, rows=200, edge_percent=50)
    assert (forced["k"], forced["rows"], forced["rows_edge"]) == (8, 200, 100)
    # columns 0 .. nx-2 are covered by strips of 128 - 2*halo output columns
    for k, wout in ((1, 124), (2, 124), (3, 120), (4, 120), (5, 116), (8, 112)):
        for nx in (3, wout + 1, wout + 2, 1000):
            assert _lib.poisson_plan(nx, 100, k=k)["strips"] == -(-(nx - 1) // wout)
    # slab of a taller grid on a smaller part
    half = _lib.poisson_plan(2559, 4215, num_simds=512)
    assert half["tasks"] <= 1024
    with pytest.raises(_lib.NskError):
        _lib.poisson_plan(2, 100)
