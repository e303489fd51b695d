#!/usr/bin/env python3
"""Compares fixtures written by julia/make_fixture.jl (the REFERENCE's own kernels of
scripts/NavierStokes.jl:115-250 on ParallelStencil's Threads backend) with this repository.

    python tools/check_reference_fixture.py FIXTURE_DIR                 # the HIP library on cuda:0
    python tools/check_reference_fixture.py FIXTURE_DIR --backend oracle  # the CPU oracle (test infrastructure)
    python tools/check_reference_fixture.py OUT_DIR --write-from-oracle   # a fixture of the same format from the
                                                                          # oracle (self-test of this tool)

FIXTURE_DIR holds one directory per recipe:
  kernels_<nx>x<ny>      every kernel once on exactly representable pseudo-data, and the rows bc_x! changed
  steps_<nx>x<ny>_<n>    n whole time steps of runme() from its initial condition
each with manifest.txt (name, f64|i64, n1, n2) and one raw little-endian column-major file per array.

Report: per array bit-identical / max abs and relative L2 difference.  The launch range of bc_x! (one
parallel index, launched without ranges: scripts/NavierStokes.jl:155-159,181,187) is read from the
fixture and printed; time steps are run with both settings of `full_rows` so that the answer shows
in the fields as well.  Exit status 0 when everything is bit-identical with the setting the fixture's
bc_x! rows imply."""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from navierstokes_amd import ns_params  # noqa: E402


# ---- fixture files -------------------------------------------------------------------------------
def read_fixture(d: str) -> dict:
    out = {}
    with open(os.path.join(d, "manifest.txt")) as f:
        for line in f:
            if not line.strip():
                continue
            name, ty, n1, n2 = line.split()
            a = np.fromfile(os.path.join(d, name + ".bin"), dtype="<f8" if ty == "f64" else "<i8")
            out[name] = np.asfortranarray(a.reshape((int(n1), int(n2)), order="F"))
    return out


def write_fixture(d: str, arrays: dict) -> None:
    os.makedirs(d, exist_ok=True)
    lines = []
    for name, a in arrays.items():
        a = np.asarray(a)
        ty = "i64" if a.dtype.kind in "iu" else "f64"
        a = np.atleast_1d(a).astype("<i8" if ty == "i64" else "<f8")
        a2 = a.reshape((a.shape[0], -1), order="F") if a.ndim > 1 else a.reshape((-1, 1))
        np.asfortranarray(a2).ravel(order="F").tofile(os.path.join(d, name + ".bin"))
        lines.append(f"{name} {ty} {a2.shape[0]} {a2.shape[1]}")
    with open(os.path.join(d, "manifest.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


# ---- the pseudo-data of make_fixture.jl ----------------------------------------------------------
def pseudo(n1: int, n2: int, k: int, s: float = 1.0) -> np.ndarray:
    ix = np.arange(1, n1 + 1, dtype=np.int64)[:, None]
    iy = np.arange(1, n2 + 1, dtype=np.int64)[None, :]
    return np.asfortranarray(s * (((ix * 7919 + iy * 104729 + k * 1299709) % 2048) / 1024.0 - 1.0))


# ---- backends: the kernels on host arrays --------------------------------------------------------
class OracleKernels:
    name = "oracle"

    def __init__(self):
        from oracle.ns_oracle import Oracle
        self.o = Oracle()

    def update_tau(self, txx, tyy, txy, Vx, Vy, p): self.o.update_tau(txx, tyy, txy, Vx, Vy, p.mu, p.dx, p.dy)
    def predict_V(self, Vx, Vy, txx, tyy, txy, p): self.o.predict_V(Vx, Vy, txx, tyy, txy, p.rho, p.g, p.dt, p.dx, p.dy)
    def set_sphere(self, Cc, Vx, Vy, p): self.o.set_sphere(Cc, Vx, Vy, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.dx, p.dy)
    def update_divV(self, dv, Vx, Vy, p): self.o.update_divV(dv, Vx, Vy, p.dx, p.dy)
    def update_dPrdtau(self, Pr, dP, dv, p): self.o.update_dPrdtau(Pr, dP, dv, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy)
    def update_Pr(self, Pr, dP, p): self.o.update_Pr(Pr, dP, p.dtau)
    def set_bc_Pr(self, Pr, p): self.o.set_bc_Pr(Pr, 0.0, p.full_rows)
    def compute_res(self, Rp, Pr, dv, p): self.o.compute_res(Rp, Pr, dv, p.rho, p.dt, p.dx, p.dy)
    def maxabs(self, a): return self.o.maxabs(a)
    def correct_V(self, Vx, Vy, Pr, p): self.o.correct_V(Vx, Vy, Pr, p.dt, p.rho, p.dx, p.dy)
    def set_bc_Vel(self, Vx, Vy, vprof, p): self.o.set_bc_Vel(Vx, Vy, np.ascontiguousarray(vprof.ravel()), p.full_rows)
    def advect(self, Vx, Vx_o, Vy, Vy_o, Cc, C_o, p): self.o.advect(Vx, Vx_o, Vy, Vy_o, Cc, C_o, p.dt, p.dx, p.dy)

    def time_steps(self, p, nsteps):
        from oracle.ns_oracle import OracleSim
        s = OracleSim(p)
        iters, hs, hi, he = [], [], [], []
        for it in range(1, nsteps + 1):
            iters.append(s.step())
            hs += [it] * len(s.iter_evo)
            hi += s.iter_evo
            he += s.err_evo
        return dict(iters=iters, hist_step=hs, hist_iter=hi, hist_err=he, Pr=s.Pr, Vx=s.Vx, Vy=s.Vy, C=s.C,
                    dPrdtau=s.dPrdtau, Vprof=s.Vprof)


class HipKernels:
    """The C ABI of libnsk on cuda:0: arrays go up, one entry point runs, arrays come back."""
    name = "hip"

    def __init__(self):
        import torch
        from navierstokes_amd import _lib
        if not torch.cuda.is_available():
            raise SystemExit("the hip backend needs a GPU (there is no CPU path); use --backend oracle to check the oracle")
        self.torch = torch
        torch.cuda.set_device(0)
        self.ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)

    def _run(self, arrays, fn):
        torch = self.torch
        dev = [torch.from_numpy(np.ascontiguousarray(a.T)).to("cuda:0") for a in arrays]
        fn(*[C.c_void_p(t.data_ptr()) for t in dev])
        torch.cuda.synchronize()
        for a, t in zip(arrays, dev):
            a[...] = t.cpu().numpy().T

    def update_tau(self, txx, tyy, txy, Vx, Vy, p):
        nx, ny = txx.shape
        self._run((txx, tyy, txy, Vx, Vy), lambda a, b, c, d, e: self.ctx.update_tau(a, b, c, d, e, p.mu, p.dx, p.dy, nx, ny))

    def predict_V(self, Vx, Vy, txx, tyy, txy, p):
        nx, ny = txx.shape
        self._run((Vx, Vy, txx, tyy, txy), lambda a, b, c, d, e: self.ctx.predict_V(a, b, c, d, e, p.rho, p.g, p.dt, p.dx, p.dy, nx, ny))

    def set_sphere(self, Cc, Vx, Vy, p):
        nx, ny = Cc.shape
        self._run((Cc, Vx, Vy), lambda a, b, c: self.ctx.set_sphere(a, b, c, p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly,
                                                                    p.dx, p.dy, nx, ny, 0))

    def update_divV(self, dv, Vx, Vy, p):
        nx, ny = dv.shape
        self._run((dv, Vx, Vy), lambda a, b, c: self.ctx.update_divV(a, b, c, p.dx, p.dy, nx, ny))

    def update_dPrdtau(self, Pr, dP, dv, p):
        nx, ny = Pr.shape
        self._run((Pr, dP, dv), lambda a, b, c: self.ctx.update_dPrdtau(a, b, c, p.rho, p.dt, p.dtau, p.damp, p.dx, p.dy, nx, ny))

    def update_Pr(self, Pr, dP, p):
        nx, ny = Pr.shape
        self._run((Pr, dP), lambda a, b: self.ctx.update_Pr(a, b, p.dtau, nx, ny))

    def set_bc_Pr(self, Pr, p):
        nx, ny = Pr.shape
        self._run((Pr,), lambda a: self.ctx.set_bc_Pr(a, 0.0, nx, ny, int(p.full_rows), 0, ny))

    def compute_res(self, Rp, Pr, dv, p):
        nx, ny = Pr.shape
        self._run((Rp, Pr, dv), lambda a, b, c: self.ctx.compute_res(a, b, c, p.rho, p.dt, p.dx, p.dy, nx, ny))

    def maxabs(self, a):
        m = C.c_double(0.0)
        self._run((a,), lambda x: self.ctx.maxabs(x, a.size, C.byref(m)))
        return m.value

    def correct_V(self, Vx, Vy, Pr, p):
        nx, ny = Pr.shape
        self._run((Vx, Vy, Pr), lambda a, b, c: self.ctx.correct_V(a, b, c, p.dt, p.rho, p.dx, p.dy, nx, ny))

    def set_bc_Vel(self, Vx, Vy, vprof, p):
        nx, ny = Vy.shape[0], Vx.shape[1]
        self._run((Vx, Vy, vprof), lambda a, b, c: self.ctx.set_bc_Vel(a, b, c, nx, ny, int(p.full_rows), 0, ny))

    def advect(self, Vx, Vx_o, Vy, Vy_o, Cc, C_o, p):
        nx, ny = Cc.shape
        self._run((Vx, Vx_o, Vy, Vy_o, Cc, C_o),
                  lambda a, b, c, d, e, f: self.ctx.advect(a, b, c, d, e, f, p.dt, p.dx, p.dy, nx, ny, 0, ny))

    def time_steps(self, p, nsteps):
        from navierstokes_amd.ns import NavierStokes, init_vprof
        s = NavierStokes(p)             # the fused entry point nsk_ns_step, what runme() uses
        iters, hs, hi, he = [], [], [], []
        for it in range(1, nsteps + 1):
            iters.append(s.step())
            hs += [it] * len(s.iter_evo)
            hi += s.iter_evo
            he += s.err_evo
        f = s.host_fields()
        return dict(iters=iters, hist_step=hs, hist_iter=hi, hist_err=he, dPrdtau=s.dPrdtau_host(), Vprof=init_vprof(p), **f)


# ---- the recipes of make_fixture.jl --------------------------------------------------------------
def kernel_recipe(K, nx: int, ny: int, full_rows: bool) -> dict:
    """Same calls, same inputs, same names as kernel_dumps() of julia/make_fixture.jl."""
    p = ns_params("NavierStokes", nx=nx, ny=ny, full_rows=full_rows)
    z = lambda a, b: np.zeros((a, b), order="F")
    out = {}
    Vx, Vy = pseudo(nx + 1, ny, 1), pseudo(nx, ny + 1, 2)
    txx, tyy, txy = z(nx, ny), z(nx, ny), z(nx - 1, ny - 1)
    K.update_tau(txx, tyy, txy, Vx, Vy, p)
    out.update(tau_txx=txx.copy(), tau_tyy=tyy.copy(), tau_txy=txy.copy())
    txx, tyy, txy = pseudo(nx, ny, 3), pseudo(nx, ny, 4), pseudo(nx - 1, ny - 1, 5)
    K.predict_V(Vx, Vy, txx, tyy, txy, p)
    out.update(predict_Vx=Vx.copy(), predict_Vy=Vy.copy())
    Cc, Vx, Vy = pseudo(nx, ny, 6), pseudo(nx + 1, ny, 1), pseudo(nx, ny + 1, 2)
    K.set_sphere(Cc, Vx, Vy, p)
    out.update(sphere_C=Cc.copy(), sphere_Vx=Vx.copy(), sphere_Vy=Vy.copy())
    Vx, Vy, dv = pseudo(nx + 1, ny, 1), pseudo(nx, ny + 1, 2), z(nx, ny)
    K.update_divV(dv, Vx, Vy, p)
    out.update(div_divV=dv.copy())
    Pr, dP, dv = pseudo(nx, ny, 7), pseudo(nx - 2, ny - 2, 8), pseudo(nx, ny, 9)
    K.update_dPrdtau(Pr, dP, dv, p)
    out.update(dPrdtau_dPrdtau=dP.copy())
    K.update_Pr(Pr, dP, p)
    out.update(updatePr_Pr=Pr.copy())
    K.set_bc_Pr(Pr, p)
    out.update(bcPr_Pr=Pr.copy())
    Rp = z(nx - 2, ny - 2)
    K.compute_res(Rp, Pr, dv, p)
    out.update(res_Rp=Rp.copy(), res_maxabs=np.array([K.maxabs(Rp)]))
    Vx, Vy, Pr = pseudo(nx + 1, ny, 1), pseudo(nx, ny + 1, 2), pseudo(nx, ny, 7)
    K.correct_V(Vx, Vy, Pr, p)
    out.update(correct_Vx=Vx.copy(), correct_Vy=Vy.copy())
    ix = np.arange(1, nx + 1, dtype=np.int64)
    vprof = np.asfortranarray((((ix * 7919 + 10 * 1299709) % 2048) / 1024.0 - 1.0).reshape(nx, 1))
    K.set_bc_Vel(Vx, Vy, vprof, p)
    out.update(bcVel_Vx=Vx.copy(), bcVel_Vy=Vy.copy())
    Vx_o, Vy_o, C_o = pseudo(nx + 1, ny, 1, 2.5), pseudo(nx, ny + 1, 2, 2.5), pseudo(nx, ny, 6)
    Vx, Vy, Cc = Vx_o.copy(order="F"), Vy_o.copy(order="F"), C_o.copy(order="F")
    K.advect(Vx, Vx_o, Vy, Vy_o, Cc, C_o, p)
    out.update(advect_Vx=Vx, advect_Vy=Vy, advect_C=Cc)
    return out


def bc_x_rows(nx: int, ny: int, full_rows: bool) -> np.ndarray:
    return np.arange(1, (ny if full_rows else min(nx, ny)) + 1, dtype=np.int64).reshape(-1, 1)


def steps_recipe(K, nx: int, ny: int, nsteps: int, full_rows: bool) -> dict:
    p = ns_params("NavierStokes", nx=nx, ny=ny, full_rows=full_rows)
    r = K.time_steps(p, nsteps)
    out = {k: np.asarray(r[k]) for k in ("Pr", "Vx", "Vy", "C", "dPrdtau")}
    out["Vprof"] = np.asarray(r["Vprof"], dtype=np.float64).reshape(-1, 1)
    for k in ("iters", "hist_step", "hist_iter"):
        out[k] = np.asarray(r[k], dtype=np.int64).reshape(-1, 1)
    out["hist_err"] = np.asarray(r["hist_err"], dtype=np.float64).reshape(-1, 1)
    for k in ("ly", "rho", "vin", "psc", "lx", "ox", "oy", "mu", "g", "a2", "b2", "sinb", "cosb", "eps_it", "dx", "dy", "dt",
              "damp", "dtau"):
        out["scalar_" + k.replace("_", "")] = np.array([getattr(p, k)])
    return out


# ---- comparison ----------------------------------------------------------------------------------
def compare(name: str, ref: np.ndarray, got: np.ndarray) -> bool:
    ref, got = np.asarray(ref), np.asarray(got)
    if ref.size == got.size and ref.shape != got.shape:
        got = got.reshape(ref.shape, order="F")
    if ref.shape != got.shape:
        print(f"    {name:18s} SHAPE {ref.shape} (reference) vs {got.shape}")
        return False
    if ref.dtype.kind in "iu":
        same = bool(np.array_equal(ref, got))
        print(f"    {name:18s} {'identical' if same else f'DIFFERENT: reference {ref.ravel()[:12]} here {got.ravel()[:12]}'}")
        return same
    same = bool(np.array_equal(ref, got, equal_nan=True))       # +0 == -0, NaN == NaN
    if same:
        print(f"    {name:18s} bit-identical ({ref.size} values)")
        return True
    with np.errstate(invalid="ignore"):
        d = np.abs(ref - got)
        nrm = np.linalg.norm(ref.ravel())
        l2 = np.linalg.norm(np.nan_to_num(ref - got).ravel()) / (nrm if nrm > 0 else 1.0)
    bad = np.argwhere(~((ref == got) | (np.isnan(ref) & np.isnan(got))))
    rows = sorted({int(b[1]) + 1 for b in bad}) if bad.ndim == 2 and bad.shape[1] == 2 else []
    where = f", rows iy {rows[0]}..{rows[-1]}" if rows else ""
    print(f"    {name:18s} DIFFERENT in {len(bad)} of {ref.size} values, max |diff| {np.nanmax(d):.3e}, rel L2 {l2:.3e}{where}")
    return False


def check_dir(K, d: str) -> bool:
    base = os.path.basename(os.path.normpath(d))
    fx = read_fixture(d)
    nx, ny = int(fx["nx"][0, 0]), int(fx["ny"][0, 0])
    ok = True
    if base.startswith("kernels_"):
        rows = fx["bc_x_rows_Pr"].ravel()
        full = bool(len(rows) == ny and ny > nx)
        limited = bool(np.array_equal(rows, np.arange(1, min(nx, ny) + 1)))
        print(f"  reference bc_x!(Pr {nx} x {ny}) changed rows {rows[0] if len(rows) else '-'}..{rows[-1] if len(rows) else '-'}"
              f" ({len(rows)} of {ny}):", "rows 1..size(A,1) -> full_rows = 0 (this repository's default)" if limited and not full
              else "every row -> full_rows = 1" if full else "an unexpected set: look at bc_x_rows_Pr.bin")
        setting = full
        got = kernel_recipe(K, nx, ny, setting)
        got["bc_x_rows_Pr"] = bc_x_rows(nx, ny, setting)
        got["bc_x_rows_Vy"] = bc_x_rows(nx, ny + 1, setting)
        print(f"  every kernel once, {K.name} backend with full_rows = {int(setting)}:")
        for name in fx:
            if name in ("nx", "ny"):
                continue
            ok &= compare(name, fx[name], got[name]) if name in got else True
        return ok
    nsteps = int(fx["steps"][0, 0])
    verdicts = {}
    for full in (False, True):
        print(f"  {nsteps} time steps, {K.name} backend with full_rows = {int(full)}:")
        got = steps_recipe(K, nx, ny, nsteps, full)
        v = True
        for name in fx:
            if name in ("nx", "ny", "steps") or name not in got:
                continue
            v &= compare(name, fx[name], got[name])
        verdicts[full] = v
    which = [int(f) for f, v in verdicts.items() if v]
    print(f"  => bit-identical with full_rows in {which}" if which else "  => NOT bit-identical with either setting")
    return bool(which)


def write_from_oracle(out: str) -> None:
    K = OracleKernels()
    for nx, ny in ((21, 37),):
        a = kernel_recipe(K, nx, ny, False)
        a.update(nx=np.array([nx]), ny=np.array([ny]), bc_x_rows_Pr=bc_x_rows(nx, ny, False),
                 bc_x_rows_Vy=bc_x_rows(nx, ny + 1, False))
        write_fixture(os.path.join(out, f"kernels_{nx}x{ny}"), a)
    for nx, ny, n in ((21, 37, 3), (153, 255, 2)):
        a = steps_recipe(K, nx, ny, n, False)
        a.update(nx=np.array([nx]), ny=np.array([ny]), steps=np.array([n]))
        write_fixture(os.path.join(out, f"steps_{nx}x{ny}_{n}"), a)
    print("oracle fixture written to", out)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dir")
    ap.add_argument("--backend", choices=["hip", "oracle"], default="hip")
    ap.add_argument("--write-from-oracle", action="store_true")
    a = ap.parse_args(argv)
    if a.write_from_oracle:
        write_from_oracle(a.dir)
        return 0
    K = HipKernels() if a.backend == "hip" else OracleKernels()
    subdirs = sorted(s for s in os.listdir(a.dir) if os.path.exists(os.path.join(a.dir, s, "manifest.txt"))
                     and (s.startswith("kernels_") or s.startswith("steps_")))
    if not subdirs:
        raise SystemExit(f"no kernels_* / steps_* fixture under {a.dir}")
    ok = True
    for s in subdirs:
        print(f"{s}:")
        ok &= check_dir(K, os.path.join(a.dir, s))
    print("RESULT", "all bit-identical" if ok else "DIFFERENCES (see above)")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
