"""CPU checks behind the sweeps-in-flight SOR kernel: the model of its schedule against the serial
sweep, and the five-operation division against the IEEE quotient."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_schedule_model_reproduces_the_serial_sweep():
    import sim_sor_schedule as m
    rng = np.random.default_rng(3)
    for (nx, ny), maxit in (((5, 5), 9), ((9, 4), 5), ((4, 9), 40), ((12, 5), 3), ((16, 16), 40)):
        assert m.plan(nx, ny, maxit)[0] >= 2
        p = rng.standard_normal((ny + 2, nx + 2))
        coef = 0.2 + 0.05 * rng.random((ny + 2, nx + 2))
        for tol in (0.0, 1e-2, 10.0):
            ref, its, ch = m.serial(p, coef, 1.2, maxit, tol)
            got, kc, ch2, _ = m.piped(p, coef, 1.2, maxit, tol)   # asserts its own hazards
            assert kc == its and ch == ch2 and np.array_equal(ref, got)


def test_five_operation_division_equals_ieee(tmp_path):
    exe = str(tmp_path / "fma_division_check")
    src = os.path.join(ROOT, "tools", "micro", "fma_division_check.c")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe, "5"], check=True, capture_output=True, text=True).stdout
    assert " 0 mismatches" in out, out
