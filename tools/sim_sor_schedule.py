#!/usr/bin/env python3
"""CPU model of the schedule of the several-sweeps-in-flight SOR kernel (k_sor_pipe): W groups of nx
threads, thread (w, ci) on row r of sweep b*W + w + 1 at step 2w + b*P + ci + r.  Steps are executed
as on the device (all reads of a step see the state before it) and the model asserts that no cell
read in a step is written in the same step by another thread, that snapshots and maximum words are
not reused before their sweep has been judged, and that values, sweep count and last change equal
the serial lexicographic sweep's.  Runs without a GPU."""
import itertools
import sys

import numpy as np


def serial(p, coef, beta, maxit, tol):
    p = p.copy()
    ny, nx = p.shape[0] - 2, p.shape[1] - 2
    for it in range(1, maxit + 1):
        ch = 0.0
        for j in range(1, ny + 1):
            for i in range(1, nx + 1):
                old = p[j, i]
                nw = (1 - beta) * old + beta * coef[j, i] * (p[j, i + 1] + p[j, i - 1] + p[j + 1, i] + p[j - 1, i])
                p[j, i] = nw
                ch = max(ch, abs(old - nw))
        if ch < tol or it == maxit:
            return p, it, ch
    raise AssertionError


def plan(nx, ny, maxit):
    W = 1024 // nx if nx <= 512 else 0
    W = min(W, (ny + 1) // 2, maxit)
    P = max(ny, 2 * W)
    M = (nx + ny + P - 1) // P if W else 1
    return W, P, M


def piped(p, coef, beta, maxit, tol):
    ny, nx = p.shape[0] - 2, p.shape[1] - 2
    W, P, M = plan(nx, ny, maxit)
    assert W >= 2
    R, ND = W * M, nx + ny - 1
    a = p.copy()
    snap = [None] * R
    snap_owner = [0] * R
    smax = [0.0] * R
    judged = 0  # sweeps judged so far
    th = [dict(w=w, ci=ci, r=-(2 * w + ci), k=w + 1, mymax=0.0) for w in range(W) for ci in range(nx)]
    c, batch = -(ND - 1), 0
    limit = (maxit // W + 2) * P + 2 * W + ND
    for t in range(limit):
        reads, writes, new = {}, {}, {}
        for T in th:
            if 0 <= T["r"] < ny and T["k"] <= maxit:
                j, i, k = T["r"] + 1, T["ci"] + 1, T["k"]
                ks = k % R
                if T["r"] == 0:
                    T["mymax"] = 0.0
                    if T["ci"] == 0:
                        assert k - R <= judged, "maximum word reused before its sweep was judged"
                        smax[ks] = 0.0
                        snap[ks] = np.full_like(a, np.nan)
                        snap_owner[ks] = k
                assert snap_owner[ks] == k
                old = a[j, i]
                for cell in ((j, i + 1), (j, i - 1), (j + 1, i), (j - 1, i)):
                    reads[cell] = reads.get(cell, []) + [id(T)]
                nw = (1 - beta) * old + beta * coef[j, i] * (a[j, i + 1] + a[j, i - 1] + a[j + 1, i] + a[j - 1, i])
                assert (j, i) not in writes
                writes[(j, i)] = id(T)
                new[(j, i)] = (nw, ks)
                T["mymax"] = max(T["mymax"], abs(old - nw))
                if T["r"] == ny - 1:
                    smax[ks] = max(smax[ks], T["mymax"])
            T["r"] += 1
            if T["r"] == P:
                T["r"] = 0
                T["k"] += W
        for cell in writes:
            assert cell not in reads, f"step {t}: cell {cell} read and written in one step"
        for (j, i), (nw, ks) in new.items():
            a[j, i] = nw
            snap[ks][j, i] = nw
        if c >= 0 and c % 2 == 0 and c // 2 < W:
            kc = batch * W + c // 2 + 1
            if kc <= maxit:
                assert kc == judged + 1
                judged = kc
                assert snap_owner[kc % R] == kc
                ch = smax[kc % R]
                if ch < tol or kc == maxit:
                    out = p.copy()
                    out[1:-1, 1:-1] = snap[kc % R][1:-1, 1:-1]
                    return out, kc, ch, t + 1
        c += 1
        if c == P:
            c, batch = 0, batch + 1
    raise AssertionError("step limit reached")


def main():
    rng = np.random.default_rng(0)
    shapes = [(2, 4), (3, 3), (4, 9), (9, 4), (5, 5), (7, 2), (2, 7), (8, 8), (12, 5), (5, 12), (16, 16), (33, 6), (6, 33)]
    n = 0
    for (nx, ny), maxit in itertools.product(shapes, (2, 3, 5, 9, 40)):
        if plan(nx, ny, maxit)[0] < 2:
            continue
        p = rng.standard_normal((ny + 2, nx + 2))
        coef = 0.2 + 0.05 * rng.random((ny + 2, nx + 2))
        for tol in (0.0, 1e-2, 1e-1, 10.0):
            ref, its, ch = serial(p, coef, 1.2, maxit, tol)
            got, kc, ch2, steps = piped(p, coef, 1.2, maxit, tol)
            assert kc == its and ch == ch2 and np.array_equal(ref, got), (nx, ny, maxit, tol)
            n += 1
    print(f"{n} cases: the schedule reproduces the serial sweep and no step reads a cell it writes")


if __name__ == "__main__":
    sys.exit(main())
