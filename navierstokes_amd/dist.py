"""Slab decomposition of the pressure-Poisson loop over the GPUs of one node.

The reference is single-device (no MPI/NCCL anywhere in /root/reference); this is the
multi-GPU form of its Poisson loop, scripts/NavierStokes.jl:84-95.  The global grid
(nx x nyg) is cut along iy (the slow index, so halo rows are contiguous memory): rank g owns
global rows g*ny_loc+1 .. (g+1)*ny_loc and holds `halo` extra rows on each interior side.

One pass of the iteration moves information by one row, so a window with `halo` valid halo
rows can run `halo` passes before it needs fresh halos: the ranks exchange `halo` rows of Pr
and dPrdτ once per `halo` passes (one message per field and neighbour, over xGMI through
torch.distributed = RCCL) instead of one row per pass; the result is bit-identical to the
single-domain iteration because every owned cell is computed by the same expression from the
same inputs.  The residual check (NavierStokes.jl:89-90) is a local max over owned rows
followed by one MAX all-reduce of the value's bit pattern (|x| >= 0, so the integer order is
the IEEE order and NaN stays the largest).

The compute backend is injected (`backend`): the product passes the HIP library binding; the
CPU tests pass a stand-in so that the decomposition and the exchange run under gloo.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist

from .params import NSParams


@dataclass
class Slab:
    rank: int
    world: int
    ny_loc: int      # rows owned by every rank
    halo: int        # halo rows on each interior side
    nyg: int         # global rows

    @property
    def lo_halo(self) -> int:
        return self.halo if self.rank > 0 else 0

    @property
    def hi_halo(self) -> int:
        return self.halo if self.rank < self.world - 1 else 0

    @property
    def iy0(self) -> int:
        """global 0-based row of the window's first row"""
        return self.rank * self.ny_loc - self.lo_halo

    @property
    def ny_win(self) -> int:
        return self.ny_loc + self.lo_halo + self.hi_halo

    @property
    def own_lo(self) -> int:
        """first owned window row, 0-based"""
        return self.lo_halo

    @property
    def own_hi(self) -> int:
        """one past the last owned window row, 0-based"""
        return self.lo_halo + self.ny_loc


def make_slab(rank: int, world: int, ny_loc: int, halo: int) -> Slab:
    if world > 1 and halo > ny_loc:
        raise ValueError(f"halo {halo} larger than the {ny_loc} rows a rank owns")
    return Slab(rank, world, ny_loc, halo if world > 1 else 0, ny_loc * world)


def exchange_rows(fields, slab: Slab, nrows: Optional[int] = None) -> None:
    """Refresh the halo rows of every field in `fields` (tensors shaped (ny_win, ld), row = iy)
    from the neighbours' owned rows.  One batched isend/irecv per neighbour and field."""
    if slab.world == 1:
        return
    h = slab.halo if nrows is None else nrows
    ops = []
    lo, hi = slab.own_lo, slab.own_hi
    for t in fields:
        if slab.rank > 0:                       # lower neighbour
            ops.append(dist.P2POp(dist.isend, t[lo:lo + h], slab.rank - 1))
            ops.append(dist.P2POp(dist.irecv, t[lo - h:lo], slab.rank - 1))
        if slab.rank < slab.world - 1:          # upper neighbour
            ops.append(dist.P2POp(dist.isend, t[hi - h:hi], slab.rank + 1))
            ops.append(dist.P2POp(dist.irecv, t[hi:hi + h], slab.rank + 1))
    for r in dist.batch_isend_irecv(ops):
        r.wait()


def allreduce_maxabs(value: float, device) -> float:
    """MAX all-reduce that keeps NaN: reduces the bit pattern of the non-negative double."""
    bits = struct.unpack("<q", struct.pack("<d", abs(value) if value == value else float("nan")))[0]
    t = torch.tensor([bits & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return struct.unpack("<d", struct.pack("<q", int(t.item())))[0]


class SlabPoisson:
    """Poisson loop of one rank.  `backend` provides
        poisson_iters(Pr, dP, divV, slab, count)   -- `count` passes on the window (in place)
        residual_max(Pr, divV, slab) -> float       -- max|Rp| over the owned rows
    on tensors shaped (ny_win, ld)."""

    def __init__(self, p: NSParams, slab: Slab, backend, Pr, dP, divV):
        self.p, self.slab, self.be = p, slab, backend
        self.Pr, self.dP, self.divV = Pr, dP, divV

    def iterate(self, count: int) -> None:
        """`count` passes of NavierStokes.jl:85-87 with a halo refresh every `halo` passes."""
        s = self.slab
        if s.world == 1:
            self.be.poisson_iters(self.Pr, self.dP, self.divV, s, count)
            return
        left = count
        while left > 0:
            n = min(left, s.halo)
            self.be.poisson_iters(self.Pr, self.dP, self.divV, s, n)
            exchange_rows((self.Pr, self.dP), s)
            left -= n

    def residual(self) -> float:
        """err of NavierStokes.jl:90 (before the ly^2/psc scaling), identical on every rank."""
        m = self.be.residual_max(self.Pr, self.divV, self.slab)
        return allreduce_maxabs(m, self.Pr.device)

    def solve(self):
        """NavierStokes.jl:84-95; returns (iterations, [(iter, err), ...])."""
        p = self.p
        done, hist = 0, []
        while done < p.niter:
            chunk = min(p.nchk - done % p.nchk if p.nchk > 0 else p.niter - done, p.niter - done)
            self.iterate(chunk)
            done += chunk
            if p.nchk > 0 and done % p.nchk == 0:
                err = self.residual() * p.ly ** 2 / p.psc
                hist.append((done, err))
                if err < p.eps_it or not (err - err == 0.0):
                    break
        return done, hist


class HipBackend:
    """The product backend: libnsk on this rank's GPU."""

    def __init__(self, ctx, p: NSParams, ld: int):
        import ctypes as C
        from .ns import c_params
        self.C, self.ctx, self.p, self.ld, self._c_params = C, ctx, p, ld, c_params

    def _cp(self, slab: Slab):
        return self._c_params(self.p, ld_pr=self.ld, dp_embedded=1, iy0=slab.iy0, nyg=slab.nyg,
                              ny=slab.ny_win)

    def poisson_iters(self, Pr, dP, divV, slab, count):
        C = self.C
        cp = self._cp(slab)
        self.ctx.ns_poisson_iters(C.c_void_p(Pr.data_ptr()), C.c_void_p(dP.data_ptr()),
                                  C.c_void_p(divV.data_ptr()), C.byref(cp), int(count))

    def residual_max(self, Pr, divV, slab):
        C = self.C
        cp = self._cp(slab)
        m = C.c_double(0.0)
        # owned rows, 1-based window rows; rows 1 and nyg of the global grid are never interior
        self.ctx.ns_residual(None, C.c_void_p(Pr.data_ptr()), C.c_void_p(divV.data_ptr()), C.byref(cp),
                             slab.own_lo + 1, slab.own_hi, C.byref(m))
        return m.value
