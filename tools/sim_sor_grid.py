#!/usr/bin/env python3
"""CPU model of the task numbering of the tiled lexicographic SOR (csrc/bubble_sor_grid.hip).

Checks, for many tile grids, ring sizes and sweep limits, that
  * every sweep of every tile appears exactly once in the numbering,
  * everything a task waits for (left / lower tile of its sweep, right / upper tile and -- for the last
    tile -- itself in the previous sweep, the judgement of sweep k - R for the first tile) has a smaller
    task number, which is what makes the counter-driven execution free of deadlock,
and replays the numbering with a bounded number of workers on a small grid: tiles are executed in
task order by a Gauss-Seidel tile update on a ring of copies, and the result must equal the serial
lexicographic sweeps bit for bit."""
import itertools
import sys

import numpy as np


def numbering(TI, TJ, R, maxit):
    D = TI + TJ - 2
    P = R if R > D // 2 else D // 2 + 1
    rounds = maxit + ((maxit - 1) // R) * (P - R) + D // 2
    order = [(I, J) for par in (0, 1) for J in range(TJ) for I in range(TI) if (I + J) % 2 == par]
    NT = TI * TJ
    tasks = {}
    for n in range(rounds * NT):
        rho, idx = n // NT + 1, n % NT
        I, J = order[idx]
        x = rho - (I + J) // 2
        if x < 1:
            continue
        b, m = divmod(x - 1, P)
        if m >= R:
            continue
        k = b * R + 1 + m
        if k > maxit:
            continue
        assert (I, J, k) not in tasks
        tasks[(I, J, k)] = n
    return tasks


def check(TI, TJ, R, maxit):
    t = numbering(TI, TJ, R, maxit)
    assert len(t) == TI * TJ * maxit, (TI, TJ, R, maxit, len(t))
    for (I, J, k), n in t.items():
        deps = []
        if I > 0:
            deps.append((I - 1, J, k))
        if J > 0:
            deps.append((I, J - 1, k))
        if k > 1:
            if I + 1 < TI:
                deps.append((I + 1, J, k - 1))
            if J + 1 < TJ:
                deps.append((I, J + 1, k - 1))
            if I + 1 == TI and J + 1 == TJ:
                deps.append((I, J, k - 1))
        if I == 0 and J == 0 and k > R:
            deps.append((TI - 1, TJ - 1, k - R))       # the task that judges sweep k - R
        for d in deps:
            assert t[d] < n, ("waits for a later task", (TI, TJ, R, maxit), (I, J, k), d)
    return t


def replay(nx, ny, bw, bh, R, maxit, seed=0):
    """Tile updates in task order on a ring of copies == serial sweeps."""
    rng = np.random.default_rng(seed)
    p0 = rng.standard_normal((nx + 2, ny + 2))
    f = rng.standard_normal((nx + 2, ny + 2))
    beta = 1.2

    def cell(old, e, w, n, s, ff):
        return (1.0 - beta) * old + beta * (0.25 * (e + w + n + s + ff))

    ref = p0.copy()
    for _ in range(maxit):
        for i in range(1, nx + 1):
            for j in range(1, ny + 1):
                ref[i, j] = cell(ref[i, j], ref[i + 1, j], ref[i - 1, j], ref[i, j + 1], ref[i, j - 1], f[i, j])
    TI, TJ = -(-nx // bw), -(-ny // bh)
    t = check(TI, TJ, R, maxit)
    ring = [np.full_like(p0, np.nan) for _ in range(R)]
    for (I, J, k), _ in sorted(t.items(), key=lambda kv: kv[1]):
        old = p0 if k == 1 else ring[(k - 1) % R]
        new = ring[k % R]
        i0, j0 = 1 + bw * I, 1 + bh * J
        i1, j1 = min(i0 + bw, nx + 1), min(j0 + bh, ny + 1)
        tile = old[i0 - 1:i1 + 1, j0 - 1:j1 + 1].copy()
        tile[0, :] = (p0 if I == 0 else new)[i0 - 1, j0 - 1:j1 + 1]
        tile[:, 0] = (p0 if J == 0 else new)[i0 - 1:i1 + 1, j0 - 1]
        if I + 1 == TI:
            tile[-1, :] = p0[i1, j0 - 1:j1 + 1]
        if J + 1 == TJ:
            tile[:, -1] = p0[i0 - 1:i1 + 1, j1]
        for i in range(1, i1 - i0 + 1):
            for j in range(1, j1 - j0 + 1):
                tile[i, j] = cell(tile[i, j], tile[i + 1, j], tile[i - 1, j], tile[i, j + 1], tile[i, j - 1],
                                  f[i0 + i - 1, j0 + j - 1])
        new[i0:i1, j0:j1] = tile[1:-1, 1:-1]
    got = ring[maxit % R][1:-1, 1:-1]
    assert np.array_equal(got, ref[1:-1, 1:-1]), (nx, ny, bw, bh, R, maxit)


def main():
    n = 0
    for TI, TJ in itertools.product((1, 2, 3, 4, 7, 16), (1, 2, 3, 5, 8, 32)):
        D = TI + TJ - 2
        for R in sorted({3, 4, D // 2 + 1, D // 2 + 2, D // 2 + 3, 2 * D + 5}):
            if R < 3:
                continue
            for maxit in (1, 2, 3, R - 1, R, R + 1, 2 * R + 1, 37):
                if maxit >= 1:
                    check(TI, TJ, min(R, maxit + 1) if min(R, maxit + 1) >= 3 else 3, maxit)
                    n += 1
    for args in ((9, 7, 4, 3, 3, 5), (9, 7, 4, 3, 4, 9), (13, 5, 4, 2, 3, 7), (6, 6, 6, 6, 3, 4), (10, 11, 3, 2, 5, 12),
                 (10, 11, 3, 2, 3, 12)):
        replay(*args)
        n += 1
    print(f"{n} cases: every wait is for a smaller task number; replays equal the serial sweeps")
    return 0


if __name__ == "__main__":
    sys.exit(main())
