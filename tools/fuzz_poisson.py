#!/usr/bin/env python3
"""Randomised parity sweep of the multi-pass Poisson kernel against the CPU oracle (test
infrastructure; tests/test_poisson_fuzz_gpu.py runs a few hundred cases of it, the tool any number):

    fuzz_poisson.py [--cases N] [--seed S] [--max-n 700]

Each case draws a grid (including the degenerate 3-wide ones and sizes around the 116/128-column
strip and the chunk heights), the storage form, the passes per launch K, the task height, the edge
strip percentage, XCD numbering on/off, the bc_x! row rule, a pass count, optionally a slab window of
a taller grid, a data scale (ordinary, tiny, huge, with exact zeros), and whether the call goes
through nsk_ns_poisson_iters_edges_first with random edge and skipped rows.  Values are compared as
bit patterns: signed zeros must agree as well, only NaN payloads are not compared."""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from navierstokes_amd import _lib, ns_params
from navierstokes_amd.ns import _pitch, c_params
from oracle.ns_oracle import Oracle
from util import Dev


def strict_same(a, b):
    """Equal bit patterns, any NaN equal to any NaN."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    if a.shape != b.shape:
        return False
    na, nb = np.isnan(a), np.isnan(b)
    if not np.array_equal(na, nb):
        return False
    ai, bi = a.view(np.uint64), b.view(np.uint64)
    return bool(np.array_equal(ai[~na], bi[~nb]))


def sweep(ctx, oracle, cases, seed, max_n, verbose=True):
    """Runs the cases on context `ctx`; returns the descriptions of the mismatching ones."""
    a = argparse.Namespace(cases=cases, seed=seed, max_n=max_n)
    rng = np.random.default_rng(a.seed)
    rng_pairs = np.random.default_rng(a.seed + 7919)      # (its own stream: the cases of a seed stay what they were)
    failures = []
    edge_stream = torch.cuda.Stream()
    interesting = [3, 4, 5, 6, 7, 113, 114, 115, 116, 117, 118, 119, 127, 128, 129, 130, 231, 232, 233, 234, 255, 256, 257]
    t_start, bad = time.time(), 0
    only = int(os.environ.get("FUZZ_ONLY", "-1"))         # run this case alone (the others only draw their random numbers)
    extra = [o.split("=") for o in os.environ.get("FUZZ_OPTS", "").split()]   # name=value options on top of the drawn ones
    for case in range(a.cases):
        skip = only >= 0 and case != only
        def dim():
            if rng.random() < 0.4:
                return int(rng.choice(interesting))
            return int(rng.integers(3, a.max_n + 1))
        nx, nyg = dim(), dim()
        if nx * nyg > 250_000:
            nyg = max(3, 250_000 // nx)
        pitched = bool(rng.integers(0, 2))
        K = int(rng.integers(0, 9))                       # 0 = the library's choice
        rows = int(rng.choice([0, 0, 2, 3, 5, 8, 13, 24, 48, 100]))
        edge = int(rng.choice([0, 0, 25, 50, 100]))
        xcd = int(rng.integers(0, 2))
        stream = int(rng.integers(-1, 2))
        full = int(rng.integers(0, 2))
        pairs = int(rng_pairs.choice([0, 1, 3, 3]))       # tasks in pairs: never, where estimated faster, wherever they fit
        count = int(rng.integers(1, 3 * max(K, 3) + 2))
        # whole runs of passes in one launch (grids whose tiles fit on chip): on or off, frame width, tile shape
        small = int(rng.integers(0, 2))
        s_halo = int(rng.choice([0, 0, 1, 2, 3, 5, 8, 12]))
        s_cols, s_rows = (int(rng.choice([0, 0, 2, 3, 7, 16, 40, 200])) for _ in range(2))
        if small and rng.random() < 0.6:
            count += int(rng.integers(8, 30))
        if small and rng.random() < 0.8:
            K = rows = 0                                  # (a pinned K or task height keeps the launch-per-K-passes kernel)
        scale, zero_frac = [(1.0, 0.0), (1.0, 0.0), (1e-2, 0.0), (1.0, 0.5), (1e-200, 0.0), (1e-300, 0.2), (1e-310, 0.3),
                            (1e200, 0.0), (1e300, 0.0)][int(rng.integers(0, 9))]
        if os.environ.get("FUZZ_SCALE"):                  # every case at this scale (subnormal: every task run twice)
            scale, zero_frac = float(os.environ["FUZZ_SCALE"]), 0.3
        p = ns_params("NavierStokes", nx=nx, ny=nyg)
        Pr0 = np.asfortranarray(rng.standard_normal((nx, nyg)) * scale)
        dP0 = np.asfortranarray(rng.standard_normal((nx - 2, nyg - 2)) * scale)
        dv = np.asfortranarray(rng.standard_normal((nx, nyg)) * scale * 10.0)
        if zero_frac:
            Pr0[rng.random((nx, nyg)) < zero_frac] = 0.0
            dP0[rng.random((nx - 2, nyg - 2)) < zero_frac] = 0.0
            if rng.random() < 0.5:
                Pr0[rng.random((nx, nyg)) < 0.05] = -0.0
        # window of the global grid: [lo, hi) rows, at least count+2 rows wide where halo rows exist
        lo, hi = 0, nyg
        if nyg >= 4 * count + 12 and rng.random() < 0.4:
            lo = int(rng.integers(0, nyg - 2 * count - 6))
            hi = int(rng.integers(lo + 2 * count + 4, nyg + 1))
            if rng.random() < 0.3:
                lo = 0
            if rng.random() < 0.3:
                hi = nyg
        Pr, dP = Pr0.copy(order="F"), dP0.copy(order="F")
        with np.errstate(all="ignore"):
            for _ in range(0 if skip else count):
                oracle.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
                oracle.update_Pr(Pr, dP, p.dtau)
                oracle.set_bc_Pr(Pr, 0.0, bool(full))
        wPr, wdv = np.asfortranarray(Pr0[:, lo:hi]), np.asfortranarray(dv[:, lo:hi])
        wdP = np.asfortranarray(dP0[:, lo:hi - 2])
        nyw = hi - lo
        for opt, v in (("poisson_k", K), ("poisson_rows", rows), ("poisson_edge_percent", edge), ("poisson_xcd", xcd),
                       ("poisson_stream", stream), ("poisson_pairs", pairs), ("poisson_small", small), ("poisson_small_halo", s_halo),
                       ("poisson_small_cols", s_cols), ("poisson_small_rows", s_rows)):
            try:
                ctx.set_option(opt, v)
            except Exception:
                if opt != "poisson_pairs":                # (an older build of the library, NSK_LIB: no such option)
                    raise
        for name, v in extra:
            ctx.set_option(name, int(v))
        if pitched:
            ld = _pitch(nx)
            dPr, dDv = Dev(wPr, ld, pad_value=np.nan), Dev(wdv, ld, pad_value=np.nan)
            emb = np.full((nx, nyw), np.nan, order="F")
            emb[1:-1, 1:-1] = wdP
            dDp = Dev(emb, ld, pad_value=np.nan)
            cp = c_params(p.with_(full_rows=bool(full)), ld_pr=ld, dp_embedded=1, iy0=lo, nyg=nyg, ny=nyw)
        else:
            dPr, dDv, dDp = Dev(wPr), Dev(wdv), Dev(wdP)
            cp = c_params(p.with_(full_rows=bool(full)), iy0=lo, nyg=nyg, ny=nyw)
        # sometimes through the edges-first entry point: random edge and skipped rows at both ends
        sk_lo = sk_hi = 0
        ef = None
        if rng.random() < 0.35 and nyw >= 8:
            cut = sorted(int(x) for x in rng.integers(0, nyw // 2, 2))
            sk_lo, e_lo = (cut[0], cut[1] - cut[0]) if lo > 0 else (0, int(rng.integers(0, nyw // 2)))
            cut = sorted(int(x) for x in rng.integers(0, nyw // 2, 2))
            sk_hi, e_hi = (cut[0], cut[1] - cut[0]) if hi < nyg else (0, int(rng.integers(0, nyw // 2)))
            ef = (sk_lo, e_lo, e_hi, sk_hi)
            ctx.set_option("poisson_edge_task_rows", int(rng.choice([0, 0, 2, 3, 12])))
            if skip:
                continue
            ctx.ns_poisson_iters_edges_first(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count, sk_lo, e_lo, e_hi, sk_hi,
                                             C.c_void_p(edge_stream.cuda_stream))
            edge_stream.synchronize()
        elif skip:
            continue
        else:
            ctx.ns_poisson_iters(dPr.ptr, dDp.ptr, dDv.ptr, C.byref(cp), count)
        gPr = dPr.get()
        gdP = dDp.get()
        if pitched:
            gdP = np.asfortranarray(gdP[1:-1, 1:-1])
        a0 = 0 if lo == 0 else count
        b0 = nyw if hi == nyg else nyw - count
        a0, b0 = max(a0, sk_lo), min(b0, nyw - sk_hi)     # skipped rows hold undefined values
        if b0 <= a0:
            continue
        ok = strict_same(gPr[:, a0:b0], Pr[:, lo + a0:lo + b0])
        pa, pb = max(a0, 1), min(b0, nyw - 1)
        ok = ok and strict_same(gdP[:, pa - 1:pb - 1], dP[:, lo + pa - 1:lo + pb - 1])
        if lo > 0 and sk_lo == 0:
            ok = ok and strict_same(gPr[:, 0], wPr[:, 0])
        if hi < nyg and sk_hi == 0:
            ok = ok and strict_same(gPr[:, -1], wPr[:, -1])
        desc = (f"case {case}: nx={nx} nyg={nyg} window=[{lo},{hi}) pitched={pitched} K={K}->{ctx.get_option('poisson_k')} rows={rows} "
                f"edge={edge} xcd={xcd} stream={stream} pairs={pairs} full={full} count={count} scale={scale} zeros={zero_frac} div={ctx.get_option('poisson_div_used')} "
                f"edges_first={ef} one_launch={small} frame={s_halo} tile={s_cols}x{s_rows}")
        if not ok and os.environ.get("FUZZ_SHOW"):        # where: first cells whose bit patterns differ
            for name, got, want in (("Pr", gPr[:, a0:b0], Pr[:, lo + a0:lo + b0]),
                                    ("dPrdtau", gdP[:, pa - 1:pb - 1], dP[:, lo + pa - 1:lo + pb - 1])):
                diff = np.argwhere(np.ascontiguousarray(got).view(np.uint64) != np.ascontiguousarray(want).view(np.uint64))
                print(f"  {name}: {len(diff)} cells differ; window rows {a0}..{b0}; first (ix, row in the compared range): "
                      + ", ".join(f"({i},{j}) got {got[i, j]!r} want {want[i, j]!r}" for i, j in diff[:8]), flush=True)
                if len(diff):
                    init = (Pr0 if name == "Pr" else dP0)[:, (lo + a0 if name == "Pr" else lo + pa - 1):]
                    same_as_input = sum(1 for i, j in diff if got[i, j] == init[i, j] or (got[i, j] != got[i, j] and init[i, j] != init[i, j]))
                    print(f"   {same_as_input} of the {len(diff)} hold the value the array had before the call", flush=True)
                    print("   rows:", sorted(set(int(j) for _, j in diff))[:40], flush=True)
                    for j in sorted(set(int(j) for _, j in diff))[:40]:
                        cols = sorted(int(i) for i, jj in diff if jj == j)
                        print(f"   row {j}: {len(cols)} columns {cols[0]}..{cols[-1]}", flush=True)
        if not ok:
            bad += 1
            failures.append(desc)
            print("MISMATCH", desc, flush=True)
        elif verbose and case % 25 == 0:
            print("ok", desc, f"[{time.time() - t_start:.0f} s]", flush=True)
    for opt in ("poisson_k", "poisson_rows", "poisson_edge_percent", "poisson_edge_task_rows", "poisson_small_halo",
                "poisson_small_cols", "poisson_small_rows"):
        ctx.set_option(opt, 0)
    ctx.set_option("poisson_small", 1)
    try:
        ctx.set_option("poisson_pairs", 1)
    except Exception:
        pass
    ctx.set_option("poisson_xcd", 1)
    ctx.set_option("poisson_stream", -1)
    if verbose:
        print(f"{a.cases} cases, {bad} mismatches, {time.time() - t_start:.0f} s")
    return failures


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=700)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    bad = sweep(ctx, Oracle(), a.cases, a.seed, a.max_n)
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
