"""Host side of the exact division by a constant (x/dx/dx, x/dy/dy in scripts/NavierStokes.jl:135,145):
which FMA form the library picks per divisor.  The 2-operation form q = RN(x*rh + RN(x*rl)) is only
allowed after an exhaustive check of the divisor's near-midpoint quotients; here that check is
repeated independently with exact rational arithmetic, and the chosen form is evaluated on the
worst-case dividends and on random ones."""
import math
import random
from fractions import Fraction as F

import numpy as np
import pytest

from navierstokes_amd import _lib, ns_params


def near_midpoint_significands(d, window=128):
    """All 53-bit significands X whose quotient X/D lies within ~2^-100 (relative) of a midpoint
    between two doubles: solutions of X 2^s - O D = N, O odd, |N| <= window, s = 53, 54."""
    m, _ = math.frexp(d)
    D = int(m * 2 ** 53)
    k = (D & -D).bit_length() - 1
    Dp = D >> k
    out = set()
    for s in (53, 54):
        mod = 1 << (s - k)
        inv = pow(Dp, -1, mod)
        for N in range(-window, window + 1):
            if N == 0 or N % (1 << k):
                continue
            O = (-(N >> k) * inv) % mod
            while O < (1 << 54):
                if O > (1 << 53) and O & 1 and (O * D + N) % (1 << s) == 0:
                    X = (O * D + N) >> s
                    if (1 << 52) <= X < (1 << 53):
                        out.add(X)
                O += mod
    return sorted(out)


def form_value(form, x, d, rh, rl):
    """The library's division forms in exact arithmetic with one rounding per operation."""
    RN = float
    if form == 0:
        return x / d
    if form == 2:
        return RN(F(x) * F(rh) + F(RN(F(x) * F(rl))))
    q0 = RN(F(x) * F(rh) + F(RN(F(x) * F(rl)))) if form == 4 else RN(F(x) * F(rh))
    r = RN(F(x) - F(q0) * F(d))
    return RN(F(q0) + F(r) * F(rh))


def test_benchmark_divisors_get_the_two_operation_form():
    for script, n in (("NavierStokes", 1), ("NavierStokes_Re6", 16), ("NavierStokes_Re6", 8), ("NavierStokes_Re6", 1)):
        p = ns_params(script, n=n)
        for d in (p.dx, p.dy):
            form, rh, rl = _lib.division_form(d)
            assert form == 2, (script, n, d, form)
            assert rh == 1.0 / d and F(rl) == F(float(F(1) / F(d) - F(rh)))


def test_form_choice_agrees_with_an_independent_exhaustive_check():
    rnd = random.Random(7)
    divisors = [0.1, 1 / 3, 0.6 / 2559, 1 / 4095, 15.14653634046108, 1.8133425308337055e-06, 7069.092255524926]
    divisors += [rnd.uniform(0.5, 1.0) * 2.0 ** rnd.randint(-25, 25) for _ in range(150)]
    n2 = nfail = 0
    for d in divisors:
        form, rh, rl = _lib.division_form(d)
        assert form in (2, 3, 4)
        cands = near_midpoint_significands(d)
        bad = [X for X in cands if form_value(2, float(X), d, rh, rl) != float(X) / d]
        m, _ = math.frexp(d)
        k = (int(m * 2 ** 53) & -int(m * 2 ** 53)).bit_length() - 1
        if k <= 8:
            assert (form == 2) == (not bad), (d, form, bad[:2])
        n2 += form == 2
        nfail += bool(bad)
        # whatever was chosen is exact on the worst cases and on a random sample, at several scales
        xs = [float(X) for X in cands[:40]] + [rnd.uniform(-1, 1) * 2.0 ** rnd.randint(-40, 40) for _ in range(40)]
        for x in xs:
            for scale in (1.0, 2.0 ** -700, 2.0 ** 600):
                assert form_value(form, x * scale, d, rh, rl) == (x * scale) / d, (d, form, x, scale)
    assert n2 > 100 and nfail >= 3        # the three known failing divisors above are among them


def test_divisors_outside_the_supported_range_use_ieee_division():
    for d in (2.0 ** -31, 2.0 ** 31, 1e-300, 1e300, float("inf"), float("nan"), 0.0, -0.25):
        assert _lib.division_form(d)[0] == 0, d


def test_zero_dividend_is_exact_in_every_form():
    d = 0.6 / 2559
    form, rh, rl = _lib.division_form(d)
    for f in (2, 3, 4):
        assert form_value(f, 0.0, d, rh, rl) == 0.0
