"""A second, structurally independent restatement of the NavierStokes.jl kernels: whole-array
numpy slicing that spells out ParallelStencil.FiniteDifferences2D's macros
(@all, @inn, @d_xa, @d_ya, @d_xi, @d_yi, @d2_xi, @d2_yi) instead of the oracle's index loops.
Used only to cross-check the C oracle (tests/test_oracle_ns.py)."""
import numpy as np


def inn(A): return A[1:-1, 1:-1]
def d_xa(A): return A[1:, :] - A[:-1, :]
def d_ya(A): return A[:, 1:] - A[:, :-1]
def d_xi(A): return A[1:, 1:-1] - A[:-1, 1:-1]
def d_yi(A): return A[1:-1, 1:] - A[1:-1, :-1]
def d2_xi(A): return (A[2:, 1:-1] - A[1:-1, 1:-1]) - (A[1:-1, 1:-1] - A[:-2, 1:-1])
def d2_yi(A): return (A[1:-1, 2:] - A[1:-1, 1:-1]) - (A[1:-1, 1:-1] - A[1:-1, :-2])


def divV(Vx, Vy, dx, dy):                      # macro ∇V, NavierStokes.jl:115
    return d_xa(Vx) / dx + d_ya(Vy) / dy


def update_tau(Vx, Vy, mu, dx, dy):            # :116-121
    dv = divV(Vx, Vy, dx, dy)
    txx = (2 * mu) * (d_xa(Vx) / dx - dv / 3.0)
    tyy = (2 * mu) * (d_ya(Vy) / dy - dv / 3.0)
    txy = mu * (d_yi(Vx) / dy + d_xi(Vy) / dx)
    return txx, tyy, txy


def predict_V(Vx, Vy, txx, tyy, txy, rho, g, dt, dx, dy):   # :123-127
    Vx, Vy = Vx.copy(), Vy.copy()
    Vx[1:-1, 1:-1] = inn(Vx) + dt / rho * (d_xi(txx) / dx + d_ya(txy) / dy)
    Vy[1:-1, 1:-1] = inn(Vy) + dt / rho * (d_yi(tyy) / dy + d_xa(txy) / dx - rho * g)
    return Vx, Vy


def residual(Pr, dv, rho, dt, dx, dy):         # :145
    return d2_xi(Pr) / dx / dx + d2_yi(Pr) / dy / dy - rho / dt * inn(dv)


def poisson_pass(Pr, dP, dv, rho, dt, dtau, damp, dx, dy, full_rows=False):   # :85-87
    dP = dP * (1.0 - damp) + dtau * residual(Pr, dv, rho, dt, dx, dy)
    Pr = Pr.copy()
    Pr[1:-1, 1:-1] = inn(Pr) + dtau * dP
    n1, n2 = Pr.shape
    lim = n2 if full_rows else min(n1, n2)      # bc_x! launch range
    Pr[0, :lim] = Pr[1, :lim]
    Pr[-1, :lim] = Pr[-2, :lim]
    Pr[:, 0] = Pr[:, 1]                          # bc_yval!
    Pr[:, -1] = 0.0
    return Pr, dP


def correct_V(Vx, Vy, Pr, dt, rho, dx, dy):    # :149-153
    Vx, Vy = Vx.copy(), Vy.copy()
    Vx[1:-1, 1:-1] = inn(Vx) - dt / rho * d_xi(Pr) / dx
    Vy[1:-1, 1:-1] = inn(Vy) - dt / rho * d_yi(Pr) / dy
    return Vx, Vy


def set_bc_Vel(Vx, Vy, Vprof, full_rows=False):  # :179-184
    Vx, Vy = Vx.copy(), Vy.copy()
    Vx[:, 0] = Vx[:, 1]
    Vx[:, -1] = Vx[:, -2]
    n1, n2 = Vy.shape
    lim = n2 if full_rows else min(n1, n2)
    Vy[0, :lim] = Vy[1, :lim]
    Vy[-1, :lim] = Vy[-2, :lim]
    Vy[:, 0] = Vprof
    Vy[:, -1] = Vy[:, -2]
    return Vx, Vy


def set_sphere(C, Vx, Vy, a2, b2, ox, oy, sinb, cosb, lx, ly, dx, dy):   # :225-250
    C, Vx, Vy = C.copy(), Vx.copy(), Vy.copy()
    nx, ny = C.shape
    xv = (np.arange(nx + 1) * dx - lx / 2)[:, None]
    yv = (np.arange(ny + 1) * dy - ly / 2)[None, :]
    xc, yc = xv + dx / 2, yv + dx / 2            # dx for yc, as in the reference

    def inside(x, y, thr):
        xr = (x - ox) * cosb - (y - oy) * sinb
        yr = (x - ox) * sinb + (y - oy) * cosb
        return xr * xr / a2 + yr * yr / b2 < thr

    C[inside(xc[:nx], yc[:, :ny], 1.05)] = 1.0
    Vx[inside(xv, yc[:, :ny], 1.0)] = 0.0
    Vy[inside(xc[:nx], yv, 1.0)] = 0.0
    return C, Vx, Vy


def backtrack(A_o, vxc, vyc, dt, dx, dy, IX, IY):   # :192-202 for index grids IX, IY (1-based)
    n1, n2 = A_o.shape
    ddx, ddy = dt * vxc / dx, dt * vyc / dy
    ix1 = np.clip(np.floor(IX - ddx).astype(np.int64), 1, n1)
    iy1 = np.clip(np.floor(IY - ddy).astype(np.int64), 1, n2)
    ix2, iy2 = np.clip(ix1 + 1, 1, n1), np.clip(iy1 + 1, 1, n2)
    tx = (ddx > 0).astype(float) - np.fmod(ddx, 1.0)
    ty = (ddy > 0).astype(float) - np.fmod(ddy, 1.0)
    lerp = lambda a, b, t: b * t + a * (1 - t)
    fx1 = lerp(A_o[ix1 - 1, iy1 - 1], A_o[ix2 - 1, iy1 - 1], tx)
    fx2 = lerp(A_o[ix1 - 1, iy2 - 1], A_o[ix2 - 1, iy2 - 1], tx)
    return lerp(fx1, fx2, ty)


def advect(Vx_o, Vy_o, C_o, dt, dx, dy):       # :206-223
    nx, ny = C_o.shape
    Vx, Vy, C = Vx_o.copy(), Vy_o.copy(), C_o.copy()
    # Vx: 1 < ix < nx+1, iy <= ny
    IX, IY = np.meshgrid(np.arange(2, nx + 1), np.arange(1, ny + 1), indexing="ij")
    vxc = Vx_o[1:nx, :]
    vyc = 0.25 * (Vy_o[0:nx - 1, 0:ny] + Vy_o[0:nx - 1, 1:ny + 1] + Vy_o[1:nx, 0:ny] + Vy_o[1:nx, 1:ny + 1])
    Vx[1:nx, :] = backtrack(Vx_o, vxc, vyc, dt, dx, dy, IX, IY)
    # Vy: 1 < iy < ny+1, ix <= nx
    IX, IY = np.meshgrid(np.arange(1, nx + 1), np.arange(2, ny + 1), indexing="ij")
    vxc = 0.25 * (Vx_o[0:nx, 0:ny - 1] + Vx_o[1:nx + 1, 0:ny - 1] + Vx_o[0:nx, 1:ny] + Vx_o[1:nx + 1, 1:ny])
    vyc = Vy_o[:, 1:ny]
    Vy[:, 1:ny] = backtrack(Vy_o, vxc, vyc, dt, dx, dy, IX, IY)
    # C: everywhere
    IX, IY = np.meshgrid(np.arange(1, nx + 1), np.arange(1, ny + 1), indexing="ij")
    vxc = 0.5 * (Vx_o[0:nx, :] + Vx_o[1:nx + 1, :])
    vyc = 0.5 * (Vy_o[:, 0:ny] + Vy_o[:, 1:ny + 1])
    C[:, :] = backtrack(C_o, vxc, vyc, dt, dx, dy, IX, IY)
    return Vx, Vy, C
