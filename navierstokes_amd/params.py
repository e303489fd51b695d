"""Parameter block of the reference drivers.

Mirrors ``runme()`` of /root/reference/scripts/NavierStokes.jl:14-56 and
scripts/NavierStokes_Re6.jl:14-56 expression by expression (Python floats are
IEEE doubles, and the association order of the Julia source is kept, so every
scalar is bit-identical to what Julia computes).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class NSParams:
    # numerics (NavierStokes.jl:39-49)
    nx: int
    ny: int
    niter: int
    nchk: int
    nvis: int
    nt: int
    nsave: int
    eps_it: float
    # physics (NavierStokes.jl:15-37)
    lx: float
    ly: float
    rho: float
    vin: float
    psc: float
    Re: float
    mu: float
    g: float
    a2: float
    b2: float
    ox: float
    oy: float
    sinb: float
    cosb: float
    # preprocessing (NavierStokes.jl:51-54)
    dx: float
    dy: float
    dt: float
    damp: float
    dtau: float
    # launch range of bc_x! (NavierStokes.jl:155,181,187): False = as the reference runs it
    full_rows: bool = False

    def with_(self, **kw) -> "NSParams":
        return replace(self, **kw)


def ns_params(script: str = "NavierStokes", n: int = 1, nx: int | None = None,
              ny: int | None = None, **overrides) -> NSParams:
    """Scalars of ``runme()``.

    script="NavierStokes"     -> scripts/NavierStokes.jl     (Re=1e4, 153 x 255, no ``n``)
    script="NavierStokes_Re6" -> scripts/NavierStokes_Re6.jl (Re=1e6, nx=n*160-1, ny=n*256-1;
                                 the script itself runs n=16: 2559 x 4095)
    ``nx``/``ny`` override the grid (all derived scalars follow), ``overrides``
    replace fields of the finished block (e.g. niter, nchk for short tests).
    """
    if script not in ("NavierStokes", "NavierStokes_Re6"):
        raise ValueError(f"unknown script {script!r}")
    re6 = script == "NavierStokes_Re6"
    # physics -- dimensionally independent
    ly = 1.0
    rho = 1.0
    vin = 1.0
    psc = rho * (vin * vin)            # ρ*vin^2: Julia lowers x^2 to x*x
    # nondimensional
    Re = 1e6 if re6 else 1e4
    Fr = math.inf
    lx_ly = 0.6
    a_ly = 0.05
    b_ly = 0.05
    ox_ly = 0.05
    oy_ly = -0.4
    beta = 0 * math.pi / 6
    # dimensionally dependent
    lx = lx_ly * ly
    ox = ox_ly * ly
    oy = oy_ly * ly
    mu = 1 / Re * rho * vin * ly
    g = 1 / (Fr * Fr) * (vin * vin) / ly
    a2 = (a_ly * ly) * (a_ly * ly)
    b2 = (b_ly * ly) * (b_ly * ly)
    sinb, cosb = math.sin(beta), math.cos(beta)
    # numerics
    if re6:
        ny_ = n * 256 - 1          # NavierStokes_Re6.jl:39
        nx_ = n * 160 - 1          # NavierStokes_Re6.jl:40
        nt, nvis, nsave = 10000, 50, 50
    else:
        ny_ = 255                  # NavierStokes.jl:39
        nx_ = math.ceil(ny_ * lx_ly)  # NavierStokes.jl:40 -> 153
        nt, nvis, nsave = 2000, 10, 10
    if nx is not None:
        nx_ = nx
    if ny is not None:
        ny_ = ny
    eps_it = 1e-3
    niter = 50 * nx_
    nchk = 4 * (nx_ - 1) if re6 else 1 * (nx_ - 1)
    CFLtau = 0.9 / math.sqrt(2)
    CFL_visc = 1 / 4.1
    CFL_adv = 1.0
    dx, dy = lx / nx_, ly / ny_
    dt = min(CFL_visc * (dy * dy) * rho / mu, CFL_adv * dy / vin)
    damp = 2 / ny_
    dtau = CFLtau * dy
    p = NSParams(nx=nx_, ny=ny_, niter=niter, nchk=nchk, nvis=nvis, nt=nt, nsave=nsave,
                 eps_it=eps_it, lx=lx, ly=ly, rho=rho, vin=vin, psc=psc, Re=Re, mu=mu, g=g,
                 a2=a2, b2=b2, ox=ox, oy=oy, sinb=sinb, cosb=cosb, dx=dx, dy=dy, dt=dt,
                 damp=damp, dtau=dtau)
    return p.with_(**overrides) if overrides else p
