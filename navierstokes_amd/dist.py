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

Overlap ("edges" schedule, the default on the GPU): the rows a neighbour is waiting for are the
`halo` owned rows next to an interior edge, final only after the last pass of a block of `halo`
passes.  The library cuts the last launch of a block in three (nsk_ns_poisson_iters_edges_first):
short tasks produce those rows on a second stream, the halo rows -- about to be received -- are
not produced at all, and the rest of the window runs on the main stream; the exchange is ordered
behind the second stream only, so the rows travel while the last launch is still running.  No
extra arithmetic, same values.

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
    if world > 1 and halo < 1:
        raise ValueError(f"halo must be at least 1 row when the grid is cut into {world} slabs, got {halo}")
    if world > 1 and halo > ny_loc:
        raise ValueError(f"halo {halo} larger than the {ny_loc} rows a rank owns")
    return Slab(rank, world, ny_loc, halo if world > 1 else 0, ny_loc * world)


def exchange_rows(fields, slab: Slab, nrows: Optional[int] = None, faces: bool = False, wait: bool = True):
    """Refresh the halo rows of every field in `fields` (tensors shaped (rows, ld), row = iy)
    from the neighbours' owned rows.  One batched isend/irecv per neighbour and field.

    Cell fields have ny_win rows.  Face fields (faces=True: Vy) have ny_win+1 rows; face f is
    the lower face of cell row f, a rank owns the faces of its own cell rows (the top rank
    also the last face), so the window of the lower neighbour ends one face later than its
    cell halo: h+1 faces travel downwards, h upwards.

    wait=False (device-to-device transports only) returns the requests instead of waiting: the
    transfer is ordered behind the stream that is current now, and whoever calls .wait() on the
    requests later makes the stream current THEN wait for it."""
    if slab.world == 1:
        return []
    h = slab.halo if nrows is None else nrows
    up = h + 1 if faces else h            # rows that travel towards the lower neighbour
    lo, hi = slab.own_lo, slab.own_hi
    # A gloo group with device tensors (several ranks sharing one GPU in tests) stages through
    # host memory; RCCL sends the device rows directly.
    stage = fields[0].is_cuda and dist.get_backend() == "gloo"
    ops, landing = [], []

    def send(t, a, b, peer):
        ops.append(dist.P2POp(dist.isend, t[a:b].cpu() if stage else t[a:b], peer))

    def recv(t, a, b, peer):
        if stage:
            buf = torch.empty_like(t[a:b], device="cpu")
            landing.append((t, a, b, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        else:
            ops.append(dist.P2POp(dist.irecv, t[a:b], peer))

    for t in fields:
        if slab.rank > 0:                       # lower neighbour: send my first rows, get its last
            send(t, lo, lo + up, slab.rank - 1)
            recv(t, lo - h, lo, slab.rank - 1)
        if slab.rank < slab.world - 1:          # upper neighbour: send my last rows, get its first
            send(t, hi - h, hi, slab.rank + 1)
            recv(t, hi, hi + up, slab.rank + 1)
    reqs = dist.batch_isend_irecv(ops)
    if not wait and not stage:
        return reqs
    for r in reqs:
        r.wait()
    for t, a, b, buf in landing:
        t[a:b].copy_(buf)
    return []


def allreduce_maxabs(value: float, device) -> float:
    """MAX all-reduce that keeps NaN: reduces the bit pattern of the non-negative double."""
    bits = struct.unpack("<q", struct.pack("<d", abs(value) if value == value else float("nan")))[0]
    if dist.is_initialized() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([bits & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return struct.unpack("<d", struct.pack("<q", int(t.item())))[0]


class SlabPoisson:
    """Poisson loop of one rank.  `backend` provides
        poisson_iters(Pr, dP, divV, window, count)   -- `count` passes on the window (in place)
        residual_max(Pr, divV, slab) -> float        -- max|Rp| over the owned rows
    on tensors shaped (ny_win, ld); `window` is a Slab (iy0, nyg, ny_win).

    schedule: "plain" = passes, then the exchange; "edges" = the rows to be sent are produced first in
    the last launch of a block and travel beside the rest of it (needs a backend with
    poisson_iters_edges and more than 2*halo owned rows per rank).  None = "edges" where possible,
    else "plain"."""

    def __init__(self, p: NSParams, slab: Slab, backend, Pr, dP, divV, schedule=None):
        self.p, self.slab, self.be = p, slab, backend
        self.Pr, self.dP, self.divV = Pr, dP, divV
        can_edges = (slab.world > 1 and slab.ny_loc > 2 * slab.halo and Pr.is_cuda
                     and hasattr(backend, "poisson_iters_edges"))
        if schedule is None:
            schedule = "edges" if can_edges else "plain"
        if schedule not in ("plain", "edges"):
            raise ValueError(f"unknown schedule {schedule!r}")
        if schedule == "edges" and slab.world > 1 and not can_edges:
            raise ValueError("the edges-first schedule needs the HIP backend and more than 2*halo owned rows per rank")
        if slab.world == 1:
            schedule = "plain"
        self.schedule = schedule
        if schedule == "edges":
            self.edge_stream = torch.cuda.Stream(device=Pr.device)
        self.exchanges = 0
        self.timing = False          # True: keep an event pair per exchange (exposed_us())
        self.loopback = False        # True: no peers; halo rows are refilled by device copies of the
                                     # rank's own rows (timing of the compute side only, wrong values)
        self.loopback_wire = 0       # loopback only: spin this many torch.cuda._sleep cycles where the
                                     # transfer would run (a stand-in for latency + wire time)
        self._marks = []

    # -- one block of n <= halo passes followed by the halo refresh ------------------------------
    def _block_plain(self, n: int) -> None:
        self.be.poisson_iters(self.Pr, self.dP, self.divV, self.slab, n)
        mark = self._mark()
        if self.loopback:
            s, h = self.slab, self.slab.halo
            if self.loopback_wire:
                torch.cuda._sleep(self.loopback_wire)
            for t in (self.Pr, self.dP):
                if s.rank > 0:
                    t[:h].copy_(t[s.own_lo:s.own_lo + h])
                if s.rank < s.world - 1:
                    t[s.own_hi:].copy_(t[s.own_hi - h:s.own_hi])
        else:
            exchange_rows((self.Pr, self.dP), self.slab)
        self._mark(mark)

    def _block_edges(self, n: int) -> None:
        s, h = self.slab, self.slab.halo
        main = torch.cuda.current_stream(self.Pr.device)
        # main stream: all launches but the last, then the interior rows of the last one;
        # edge stream: the rows the neighbours wait for, in short tasks
        self.be.poisson_iters_edges(self.Pr, self.dP, self.divV, s, n, self.edge_stream)
        mark = self._mark()
        if self.loopback:
            # the stand-in for the transfer follows the edge tasks on their own stream: a third stream
            # may share a hardware queue with the main stream's and would then wait for the interior
            with torch.cuda.stream(self.edge_stream):
                if self.loopback_wire:
                    torch.cuda._sleep(self.loopback_wire)
                for t in (self.Pr, self.dP):
                    if s.rank > 0:
                        t[:h].copy_(t[s.own_lo:s.own_lo + h])
                    if s.rank < s.world - 1:
                        t[s.own_hi:].copy_(t[s.own_hi - h:s.own_hi])
            main.wait_stream(self.edge_stream)
        elif dist.get_backend() == "gloo":
            # host-staged transport (tests): nothing to overlap; the library has already ordered the
            # main stream behind the edge rows
            exchange_rows((self.Pr, self.dP), s)
        else:
            with torch.cuda.stream(self.edge_stream):      # RCCL: ordered behind the edge rows only
                reqs = exchange_rows((self.Pr, self.dP), s, wait=False)
            for r in reqs:
                r.wait()                                   # the main stream (not the host) waits
        self._mark(mark)

    def _mark(self, first=None):
        """Event pair around the part of an exchange the main stream cannot hide."""
        if not (self.timing and self.Pr.is_cuda):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        if first is not None:
            self._marks.append((first, e))
        return e

    def exposed_us(self) -> Optional[float]:
        """Mean time per exchange the main stream spent between its last pass of a block and the
        installed halo (waiting for the transfer + the install copies); None if nothing was timed."""
        if not self._marks:
            return None
        torch.cuda.synchronize()
        t = sum(a.elapsed_time(b) for a, b in self._marks) / len(self._marks) * 1e3
        self._marks = []
        return t

    def iterate(self, count: int) -> None:
        """`count` passes of NavierStokes.jl:85-87 with a halo refresh every `halo` passes."""
        s = self.slab
        if s.world == 1:
            self.be.poisson_iters(self.Pr, self.dP, self.divV, s, count)
            return
        left = count
        while left > 0:
            n = min(left, s.halo)
            if self.schedule == "edges":
                self._block_edges(n)
            else:
                self._block_plain(n)
            self.exchanges += 1
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
                err = self.residual() * (p.ly * p.ly) / p.psc
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

    @staticmethod
    def _P(t):
        import ctypes as C
        return C.c_void_p(t.data_ptr())

    # whole-step kernels on windows (argument lists of include/nsk.h) ---------------------
    def update_tau(self, txx, tyy, txy, Vx, Vy, s):
        p, P = self.p, self._P
        self.ctx.update_tau(P(txx), P(tyy), P(txy), P(Vx), P(Vy), p.mu, p.dx, p.dy, p.nx, s.ny_win)

    def predict_V(self, Vx, Vy, txx, tyy, txy, s):
        p, P = self.p, self._P
        self.ctx.predict_V(P(Vx), P(Vy), P(txx), P(tyy), P(txy), p.rho, p.g, p.dt, p.dx, p.dy, p.nx, s.ny_win)

    def set_sphere(self, Cc, Vx, Vy, s):
        p, P = self.p, self._P
        self.ctx.set_sphere(P(Cc), P(Vx), P(Vy), p.a2, p.b2, p.ox, p.oy, p.sinb, p.cosb, p.lx, p.ly, p.dx, p.dy,
                            p.nx, s.ny_win, s.iy0)

    def update_divV(self, divV, Vx, Vy, s):
        p, P = self.p, self._P
        self.ctx.update_divV_ld(P(divV), self.ld, P(Vx), P(Vy), p.dx, p.dy, p.nx, s.ny_win)

    def correct_V(self, Vx, Vy, Pr, s):
        p, P = self.p, self._P
        self.ctx.correct_V_ld(P(Vx), P(Vy), P(Pr), self.ld, p.dt, p.rho, p.dx, p.dy, p.nx, s.ny_win)

    def set_bc_Vel(self, Vx, Vy, Vprof, s):
        p, P = self.p, self._P
        self.ctx.set_bc_Vel(P(Vx), P(Vy), P(Vprof), p.nx, s.ny_win, int(p.full_rows), s.iy0, s.nyg)

    def advect(self, Vx, Vx_o, Vy, Vy_o, Cc, C_o, s):
        p, P = self.p, self._P
        self.ctx.advect(P(Vx), P(Vx_o), P(Vy), P(Vy_o), P(Cc), P(C_o), p.dt, p.dx, p.dy, p.nx, s.ny_win, s.iy0, s.nyg)

    def maxabs(self, t) -> float:
        m = self.C.c_double(0.0)
        self.ctx.maxabs(self._P(t), t.numel(), self.C.byref(m))
        return m.value

    def _cp(self, slab: Slab):
        return self._c_params(self.p, ld_pr=self.ld, dp_embedded=1, iy0=slab.iy0, nyg=slab.nyg,
                              ny=slab.ny_win)

    def poisson_iters(self, Pr, dP, divV, slab, count):
        C = self.C
        cp = self._cp(slab)
        self.ctx.ns_poisson_iters(C.c_void_p(Pr.data_ptr()), C.c_void_p(dP.data_ptr()),
                             C.c_void_p(divV.data_ptr()), C.byref(cp), int(count))

    def poisson_iters_edges(self, Pr, dP, divV, slab, count, edge_stream):
        """`count` passes; in the last launch the `halo` owned rows next to each interior edge come
        first, on `edge_stream`, and the halo rows are left to the exchange that follows."""
        C = self.C
        cp = self._cp(slab)
        h = slab.halo
        self.ctx.ns_poisson_iters_edges_first(C.c_void_p(Pr.data_ptr()), C.c_void_p(dP.data_ptr()),
                                              C.c_void_p(divV.data_ptr()), C.byref(cp), int(count),
                                              slab.lo_halo, h if slab.lo_halo else 0, h if slab.hi_halo else 0,
                                              slab.hi_halo, C.c_void_p(edge_stream.cuda_stream))

    def residual_max(self, Pr, divV, slab):
        C = self.C
        cp = self._cp(slab)
        m = C.c_double(0.0)
        # owned rows, 1-based window rows; rows 1 and nyg of the global grid are never interior
        self.ctx.ns_residual(None, C.c_void_p(Pr.data_ptr()), C.c_void_p(divV.data_ptr()), C.byref(cp),
                             slab.own_lo + 1, slab.own_hi, C.byref(m))
        return m.value


class SlabNavierStokes:
    """Whole time step of runme() (scripts/NavierStokes.jl:77-100) on one rank's slab.

    Every field is a window of the global field covering the rank's rows plus `halo` rows per
    interior side (Vy: one more face).  The kernels act on the whole window with their guards on
    window extents; a kernel that needs a neighbour row leaves the outermost halo row stale, so
    validity shrinks by one row per dependent kernel and is restored by two exchanges per step:
    divV before the Poisson loop (which needs all `halo` rows) and Vx, Vy, C after advect!.
    The back-trace of advect! reaches ceil(|V| dt / d)+1 rows; a device-side max|Vy| check
    (all-reduced) guards that this stays inside the valid halo.

    `backend` provides, on window tensors shaped (rows, ld):
        update_tau, predict_V, set_sphere, update_divV, correct_V, set_bc_Vel, advect, maxabs
        poisson_iters, residual_max                          (see SlabPoisson)
    """

    def __init__(self, p: NSParams, slab: Slab, backend, device, ld: int, vprof, schedule=None):
        self.p, self.slab, self.be, self.dev, self.ld = p, slab, backend, device, ld
        if slab.world > 1 and slab.halo < 4:
            # advect! needs reach + 1 <= halo - 1 valid rows and reach >= 2 (see step())
            raise ValueError(f"a slab time step needs at least 4 halo rows, got {slab.halo}")
        nx, nw = p.nx, slab.ny_win
        z = lambda rows, cols: torch.zeros((rows, cols), dtype=torch.float64, device=device)
        # NavierStokes.jl:58-70 on the window (Pr, dPrdtau, divV pitched; the rest dense)
        self.Pr, self.dP, self.divV = z(nw, ld), z(nw, ld), z(nw, ld)
        self.C, self.C_o = z(nw, nx), z(nw, nx)
        self.txx, self.tyy, self.txy = z(nw, nx), z(nw, nx), z(nw - 1, nx - 1)
        self.Vx, self.Vx_o = z(nw, nx + 1), z(nw, nx + 1)
        self.Vy, self.Vy_o = z(nw + 1, nx), z(nw + 1, nx)
        self.Vprof = torch.as_tensor(vprof, dtype=torch.float64).to(device)
        if slab.iy0 == 0:
            self.Vy[0, :] = self.Vprof          # NavierStokes.jl:73
        self.poisson = SlabPoisson(p, slab, backend, self.Pr, self.dP, self.divV, schedule=schedule)
        self.hist = []

    def step(self) -> int:
        p, s, be = self.p, self.slab, self.be
        be.update_tau(self.txx, self.tyy, self.txy, self.Vx, self.Vy, s)
        be.predict_V(self.Vx, self.Vy, self.txx, self.tyy, self.txy, s)
        be.set_sphere(self.C, self.Vx, self.Vy, s)
        be.update_divV(self.divV, self.Vx, self.Vy, s)
        exchange_rows((self.divV,), s)
        iters, self.hist = self.poisson.solve()
        be.correct_V(self.Vx, self.Vy, self.Pr, s)
        be.set_sphere(self.C, self.Vx, self.Vy, s)
        be.set_bc_Vel(self.Vx, self.Vy, self.Vprof, s)
        if s.world > 1:
            # rows the back-trace may reach: |Vy| dt/dy + 1 below/above; one halo row is already
            # stale (predict_V!/correct_V! do not update a window's outermost row)
            reach = allreduce_maxabs(be.maxabs(self.Vy), self.dev) * p.dt / p.dy + 2
            if not reach <= s.halo - 1:
                raise RuntimeError(f"advect! reaches {reach:.1f} rows but only {s.halo - 1} valid halo rows "
                                   f"exist: raise the halo width")
        self.Vx_o.copy_(self.Vx)
        self.Vy_o.copy_(self.Vy)
        self.C_o.copy_(self.C)
        be.advect(self.Vx, self.Vx_o, self.Vy, self.Vy_o, self.C, self.C_o, s)
        exchange_rows((self.Vx, self.C), s)
        exchange_rows((self.Vy,), s, faces=True)
        return iters

    def owned(self, name: str):
        """Rows of a field this rank owns (cells; for Vy its faces, the top rank one more)."""
        s, t = self.slab, getattr(self, name)
        hi = s.own_hi + (1 if name in ("Vy", "Vy_o") and s.rank == s.world - 1 else 0)
        return t[s.own_lo:hi]
