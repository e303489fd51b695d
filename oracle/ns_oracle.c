/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's
 * NavierStokes.jl projection-method time step.  Nothing in the product path
 * may link or call this file; it is the checker used by tests/, the smoke
 * check and the CPU-baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference holds no golden vectors or tests for this
 * path (no test touches update_τ!/predict_V!/update_dPrdτ!/advect!), the
 * reference cannot run here (no Julia, ParallelStencil.jl not vendored), and
 * the finite-difference macros live in the third-party ParallelStencil.jl
 * (version unpinned by the reference: Manifest.toml is git-ignored).  Their
 * published semantics are restated below.
 *
 * Source followed: /root/reference/scripts/NavierStokes.jl (identical kernels in
 * scripts/NavierStokes_Re6.jl).  Line numbers below refer to NavierStokes.jl.
 *
 * Conventions
 *   - Arrays are Julia column-major A[ix,iy] -> a[(iy-1)*ld + (ix-1)], ld = size(A,1).
 *   - All loops are written with 1-based (ix,iy) exactly as the parallel indices
 *     of ParallelStencil so each guard can be compared with the macro rules:
 *       @all(A)  = A[ix,iy]       guard ix<=size(A,1)   && iy<=size(A,2)
 *       @inn(A)  = A[ix+1,iy+1]   guard ix<=size(A,1)-2 && iy<=size(A,2)-2
 *       @d_xa(A) = A[ix+1,iy]-A[ix,iy]        @d_ya(A) = A[ix,iy+1]-A[ix,iy]
 *       @d_xi(A) = A[ix+1,iy+1]-A[ix,iy+1]    @d_yi(A) = A[ix+1,iy+1]-A[ix+1,iy]
 *       @d2_xi(A)= (A[ix+2,iy+1]-A[ix+1,iy+1])-(A[ix+1,iy+1]-A[ix,iy+1])
 *       @d2_yi(A)= (A[ix+1,iy+2]-A[ix+1,iy+1])-(A[ix+1,iy+1]-A[ix+1,iy])
 *   - Julia never contracts a*b+c into an FMA and never re-associates:
 *     build with -ffp-contract=off and without -ffast-math.
 *   - Every kernel reads only values that the same kernel does not write at a
 *     different index (or reads a different array), so loop order is free and
 *     the OpenMP build (outer index threaded, like the Threads backend of
 *     ParallelStencil) gives bit-identical results to the serial one.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define A2(a, ld, ix, iy) (a)[(size_t)((iy) - 1) * (size_t)(ld) + (size_t)((ix) - 1)]

#ifdef _OPENMP
#define PAR_FOR _Pragma("omp parallel for schedule(static)")
#else
#define PAR_FOR
#endif

/* NavierStokes.jl:115-121  update_τ! (with macro ∇V at :115) */
void ns_update_tau(double *txx, double *tyy, double *txy, const double *Vx,
                   const double *Vy, double mu, double dx, double dy, int nx,
                   int ny) {
  const double mu2 = 2 * mu; /* "2μ*(...)" parses as (2*μ)*(...) */
  PAR_FOR
  for (int iy = 1; iy <= ny; iy++) {
    for (int ix = 1; ix <= nx; ix++) {
      double dVx = A2(Vx, nx + 1, ix + 1, iy) - A2(Vx, nx + 1, ix, iy);
      double dVy = A2(Vy, nx, ix, iy + 1) - A2(Vy, nx, ix, iy);
      double div = dVx / dx + dVy / dy;
      A2(txx, nx, ix, iy) = mu2 * (dVx / dx - div / 3.0);
      A2(tyy, nx, ix, iy) = mu2 * (dVy / dy - div / 3.0);
      if (ix <= nx - 1 && iy <= ny - 1) {
        double dyVx = A2(Vx, nx + 1, ix + 1, iy + 1) - A2(Vx, nx + 1, ix + 1, iy);
        double dxVy = A2(Vy, nx, ix + 1, iy + 1) - A2(Vy, nx, ix, iy + 1);
        A2(txy, nx - 1, ix, iy) = mu * (dyVx / dy + dxVy / dx);
      }
    }
  }
}

/* NavierStokes.jl:123-127  predict_V! */
void ns_predict_V(double *Vx, double *Vy, const double *txx, const double *tyy,
                  const double *txy, double rho, double g, double dt, double dx,
                  double dy, int nx, int ny) {
  const double dtr = dt / rho;
  const double rg = rho * g;
  PAR_FOR
  for (int iy = 1; iy <= ny - 1; iy++) {
    for (int ix = 1; ix <= nx - 1; ix++) {
      if (iy <= ny - 2) { /* @inn(Vx): size (nx+1,ny) */
        double a = A2(txx, nx, ix + 1, iy + 1) - A2(txx, nx, ix, iy + 1);
        double b = A2(txy, nx - 1, ix, iy + 1) - A2(txy, nx - 1, ix, iy);
        A2(Vx, nx + 1, ix + 1, iy + 1) =
            A2(Vx, nx + 1, ix + 1, iy + 1) + dtr * (a / dx + b / dy);
      }
      if (ix <= nx - 2) { /* @inn(Vy): size (nx,ny+1) */
        double a = A2(tyy, nx, ix + 1, iy + 1) - A2(tyy, nx, ix + 1, iy);
        double b = A2(txy, nx - 1, ix + 1, iy) - A2(txy, nx - 1, ix, iy);
        A2(Vy, nx, ix + 1, iy + 1) =
            A2(Vy, nx, ix + 1, iy + 1) + dtr * (a / dy + b / dx - rg);
      }
    }
  }
}

/* NavierStokes.jl:129-132  update_∇V! */
void ns_update_divV(double *divV, const double *Vx, const double *Vy, double dx,
                    double dy, int nx, int ny) {
  PAR_FOR
  for (int iy = 1; iy <= ny; iy++)
    for (int ix = 1; ix <= nx; ix++)
      A2(divV, nx, ix, iy) =
          (A2(Vx, nx + 1, ix + 1, iy) - A2(Vx, nx + 1, ix, iy)) / dx +
          (A2(Vy, nx, ix, iy + 1) - A2(Vy, nx, ix, iy)) / dy;
}

static inline double ns_residual(const double *Pr, const double *divV,
                                 double rho_dt, double dx, double dy, int nx,
                                 int ix, int iy) {
  double c = A2(Pr, nx, ix + 1, iy + 1);
  double d2x = (A2(Pr, nx, ix + 2, iy + 1) - c) - (c - A2(Pr, nx, ix, iy + 1));
  double d2y = (A2(Pr, nx, ix + 1, iy + 2) - c) - (c - A2(Pr, nx, ix + 1, iy));
  return d2x / dx / dx + d2y / dy / dy - rho_dt * A2(divV, nx, ix + 1, iy + 1);
}

/* NavierStokes.jl:134-137  update_dPrdτ! */
void ns_update_dPrdtau(const double *Pr, double *dPrdtau, const double *divV,
                       double rho, double dt, double dtau, double damp,
                       double dx, double dy, int nx, int ny) {
  const double rho_dt = rho / dt;
  const double keep = 1.0 - damp;
  PAR_FOR
  for (int iy = 1; iy <= ny - 2; iy++)
    for (int ix = 1; ix <= nx - 2; ix++)
      A2(dPrdtau, nx - 2, ix, iy) =
          A2(dPrdtau, nx - 2, ix, iy) * keep +
          dtau * ns_residual(Pr, divV, rho_dt, dx, dy, nx, ix, iy);
}

/* NavierStokes.jl:139-142  update_Pr! */
void ns_update_Pr(double *Pr, const double *dPrdtau, double dtau, int nx,
                  int ny) {
  PAR_FOR
  for (int iy = 1; iy <= ny - 2; iy++)
    for (int ix = 1; ix <= nx - 2; ix++)
      A2(Pr, nx, ix + 1, iy + 1) =
          A2(Pr, nx, ix + 1, iy + 1) + dtau * A2(dPrdtau, nx - 2, ix, iy);
}

/* NavierStokes.jl:144-147  compute_res! */
void ns_compute_res(double *Rp, const double *Pr, const double *divV, double rho,
                    double dt, double dx, double dy, int nx, int ny) {
  const double rho_dt = rho / dt;
  PAR_FOR
  for (int iy = 1; iy <= ny - 2; iy++)
    for (int ix = 1; ix <= nx - 2; ix++)
      A2(Rp, nx - 2, ix, iy) = ns_residual(Pr, divV, rho_dt, dx, dy, nx, ix, iy);
}

/* NavierStokes.jl:90  maximum(abs.(Rp)) -- Julia's maximum propagates NaN */
double ns_maxabs(const double *a, size_t n) {
  double m = 0.0;
  int nan = 0;
  if (n == 0) return 0.0; /* Julia would throw on an empty array */
  for (size_t i = 0; i < n; i++) {
    double v = fabs(a[i]);
    if (v != v) nan = 1;
    if (v > m) m = v;
  }
  return nan ? NAN : m;
}

/* NavierStokes.jl:149-153  correct_V! */
void ns_correct_V(double *Vx, double *Vy, const double *Pr, double dt,
                  double rho, double dx, double dy, int nx, int ny) {
  const double dtr = dt / rho;
  PAR_FOR
  for (int iy = 1; iy <= ny - 1; iy++) {
    for (int ix = 1; ix <= nx - 1; ix++) {
      if (iy <= ny - 2)
        A2(Vx, nx + 1, ix + 1, iy + 1) =
            A2(Vx, nx + 1, ix + 1, iy + 1) -
            dtr * (A2(Pr, nx, ix + 1, iy + 1) - A2(Pr, nx, ix, iy + 1)) / dx;
      if (ix <= nx - 2)
        A2(Vy, nx, ix + 1, iy + 1) =
            A2(Vy, nx, ix + 1, iy + 1) -
            dtr * (A2(Pr, nx, ix + 1, iy + 1) - A2(Pr, nx, ix + 1, iy)) / dy;
    }
  }
}

/*
 * NavierStokes.jl:155-159  bc_x!  declared "@parallel_indices (iy)" and launched
 * at :181/:187 without explicit ranges.  ParallelStencil binds a single parallel
 * index to the FIRST launch range, 1:size(A,1), so only rows iy <= size(A,1)
 * receive the condition (both reference grids have nx < ny).  `full_rows`!=0
 * selects the presumably intended 1:size(A,2) instead.
 */
void ns_bc_x(double *A, int n1, int n2, int full_rows) {
  int lim = full_rows ? n2 : (n1 < n2 ? n1 : n2);
  for (int iy = 1; iy <= lim; iy++) {
    A2(A, n1, 1, iy) = A2(A, n1, 2, iy);
    A2(A, n1, n1, iy) = A2(A, n1, n1 - 1, iy);
  }
}

/* NavierStokes.jl:161-165  bc_y! */
void ns_bc_y(double *A, int n1, int n2) {
  for (int ix = 1; ix <= n1; ix++) {
    A2(A, n1, ix, 1) = A2(A, n1, ix, 2);
    A2(A, n1, ix, n2) = A2(A, n1, ix, n2 - 1);
  }
}

/* NavierStokes.jl:167-171  bc_yV! */
void ns_bc_yV(double *A, const double *V, int n1, int n2) {
  for (int ix = 1; ix <= n1; ix++) {
    A2(A, n1, ix, 1) = V[ix - 1];
    A2(A, n1, ix, n2) = A2(A, n1, ix, n2 - 1);
  }
}

/* NavierStokes.jl:173-177  bc_yval! */
void ns_bc_yval(double *A, double val, int n1, int n2) {
  for (int ix = 1; ix <= n1; ix++) {
    A2(A, n1, ix, 1) = A2(A, n1, ix, 2);
    A2(A, n1, ix, n2) = val;
  }
}

/* NavierStokes.jl:179-184  set_bc_Vel! */
void ns_set_bc_Vel(double *Vx, double *Vy, const double *Vprof, int nx, int ny,
                   int full_rows) {
  ns_bc_y(Vx, nx + 1, ny);
  ns_bc_x(Vy, nx, ny + 1, full_rows);
  ns_bc_yV(Vy, Vprof, nx, ny + 1);
}

/* NavierStokes.jl:186-190  set_bc_Pr! */
void ns_set_bc_Pr(double *Pr, double val, int nx, int ny, int full_rows) {
  ns_bc_x(Pr, nx, ny, full_rows);
  ns_bc_yval(Pr, val, nx, ny);
}

/* NavierStokes.jl:204  lerp(a,b,t) = b*t + a*(1-t) */
static inline double ns_lerp(double a, double b, double t) {
  return b * t + a * (1 - t);
}

static inline int ns_clampi(double f, int lo, int hi) {
  /* clamp(floor(Int,f),lo,hi); the comparison is done in double so a huge or
     non-finite f (Julia would throw InexactError) maps to an edge. */
  if (!(f >= (double)lo)) return lo;
  if (f >= (double)hi) return hi;
  return (int)f; /* f is already floored and within [lo,hi) */
}

/* NavierStokes.jl:192-202  backtrack! */
static inline double ns_backtrack(const double *A_o, int n1, int n2, double vxc,
                                  double vyc, double dt, double dx, double dy,
                                  int ix, int iy) {
  double ddx = dt * vxc / dx, ddy = dt * vyc / dy;
  int ix1 = ns_clampi(floor((double)ix - ddx), 1, n1);
  int iy1 = ns_clampi(floor((double)iy - ddy), 1, n2);
  int ix2 = ix1 + 1 > n1 ? n1 : ix1 + 1;
  int iy2 = iy1 + 1 > n2 ? n2 : iy1 + 1;
  ddx = (double)(ddx > 0) - fmod(ddx, 1.0);
  ddy = (double)(ddy > 0) - fmod(ddy, 1.0);
  double fx1 = ns_lerp(A2(A_o, n1, ix1, iy1), A2(A_o, n1, ix2, iy1), ddx);
  double fx2 = ns_lerp(A2(A_o, n1, ix1, iy2), A2(A_o, n1, ix2, iy2), ddx);
  return ns_lerp(fx1, fx2, ddy);
}

/* NavierStokes.jl:206-223  advect! */
void ns_advect(double *Vx, const double *Vx_o, double *Vy, const double *Vy_o,
               double *C, const double *C_o, double dt, double dx, double dy,
               int nx, int ny) {
  PAR_FOR
  for (int iy = 1; iy <= ny + 1; iy++) {
    for (int ix = 1; ix <= nx + 1; ix++) {
      if (ix > 1 && ix < nx + 1 && iy <= ny) {
        double vxc = A2(Vx_o, nx + 1, ix, iy);
        double vyc = 0.25 * (A2(Vy_o, nx, ix - 1, iy) + A2(Vy_o, nx, ix - 1, iy + 1) +
                             A2(Vy_o, nx, ix, iy) + A2(Vy_o, nx, ix, iy + 1));
        A2(Vx, nx + 1, ix, iy) =
            ns_backtrack(Vx_o, nx + 1, ny, vxc, vyc, dt, dx, dy, ix, iy);
      }
      if (iy > 1 && iy < ny + 1 && ix <= nx) {
        double vxc = 0.25 * (A2(Vx_o, nx + 1, ix, iy - 1) + A2(Vx_o, nx + 1, ix + 1, iy - 1) +
                             A2(Vx_o, nx + 1, ix, iy) + A2(Vx_o, nx + 1, ix + 1, iy));
        double vyc = A2(Vy_o, nx, ix, iy);
        A2(Vy, nx, ix, iy) =
            ns_backtrack(Vy_o, nx, ny + 1, vxc, vyc, dt, dx, dy, ix, iy);
      }
      if (ix <= nx && iy <= ny) {
        double vxc = 0.5 * (A2(Vx_o, nx + 1, ix, iy) + A2(Vx_o, nx + 1, ix + 1, iy));
        double vyc = 0.5 * (A2(Vy_o, nx, ix, iy) + A2(Vy_o, nx, ix, iy + 1));
        A2(C, nx, ix, iy) = ns_backtrack(C_o, nx, ny, vxc, vyc, dt, dx, dy, ix, iy);
      }
    }
  }
}

/* NavierStokes.jl:225-250  set_sphere!  (yc uses dx, as the reference does).
 * iy0 is the global index offset of local row 1 (0 for a single domain). */
void ns_set_sphere(double *C, double *Vx, double *Vy, double a2, double b2,
                   double ox, double oy, double sinb, double cosb, double lx,
                   double ly, double dx, double dy, int nx, int ny, int iy0) {
  PAR_FOR
  for (int iy = 1; iy <= ny + 1; iy++) {
    for (int ix = 1; ix <= nx + 1; ix++) {
      double xv = (double)(ix - 1) * dx - lx / 2, yv = (double)(iy + iy0 - 1) * dy - ly / 2;
      double xc = xv + dx / 2, yc = yv + dx / 2;
      if (ix <= nx && iy <= ny) {
        double xr = (xc - ox) * cosb - (yc - oy) * sinb;
        double yr = (xc - ox) * sinb + (yc - oy) * cosb;
        if (xr * xr / a2 + yr * yr / b2 < 1.05) A2(C, nx, ix, iy) = 1.0;
      }
      if (iy <= ny) {
        double xr = (xv - ox) * cosb - (yc - oy) * sinb;
        double yr = (xv - ox) * sinb + (yc - oy) * cosb;
        if (xr * xr / a2 + yr * yr / b2 < 1.0) A2(Vx, nx + 1, ix, iy) = 0.0;
      }
      if (ix <= nx) {
        double xr = (xc - ox) * cosb - (yv - oy) * sinb;
        double yr = (xc - ox) * sinb + (yv - oy) * cosb;
        if (xr * xr / a2 + yr * yr / b2 < 1.0) A2(Vy, nx, ix, iy) = 0.0;
      }
    }
  }
}

/* NavierStokes.jl:72  inflow profile on LinRange(0.5dx, lx-0.5dx, nx);
 * LinRange element i is (1-t)*a + t*b with t=(i-1)/(n-1). */
void ns_init_vprof(double *Vprof, double vin, double lx, double dx, int nx) {
  double a = 0.5 * dx, b = lx - 0.5 * dx;
  for (int i = 1; i <= nx; i++) {
    double t = nx > 1 ? (double)(i - 1) / (double)(nx - 1) : 0.0;
    double x = (1 - t) * a + t * b;
    Vprof[i - 1] = 4 * vin * x / lx * (1.0 - x / lx);
  }
}

/* Parameter block of runme(), NavierStokes.jl:14-56. */
typedef struct {
  int nx, ny, niter, nchk;
  int full_rows; /* 0: bc_x! limited to size(A,1) rows (reference behaviour) */
  double lx, ly, rho, vin, psc, mu, g, a2, b2, ox, oy, sinb, cosb;
  double eps_it, dx, dy, dt, damp, dtau;
} ns_params;

/* Poisson loop, NavierStokes.jl:84-95.  Returns the number of iterations run;
 * err_hist/iter_hist receive one entry per check (capacity nhist_cap). */
int ns_poisson(double *Pr, double *dPrdtau, double *Rp, const double *divV,
               const ns_params *p, int niter, int nchk, double *err_hist,
               int *iter_hist, int nhist_cap, int *nhist_out) {
  int nh = 0, iter;
  const int nx = p->nx, ny = p->ny;
  for (iter = 1; iter <= niter; iter++) {
    ns_update_dPrdtau(Pr, dPrdtau, divV, p->rho, p->dt, p->dtau, p->damp, p->dx, p->dy, nx, ny);
    ns_update_Pr(Pr, dPrdtau, p->dtau, nx, ny);
    ns_set_bc_Pr(Pr, 0.0, nx, ny, p->full_rows);
    if (nchk > 0 && iter % nchk == 0) {
      ns_compute_res(Rp, Pr, divV, p->rho, p->dt, p->dx, p->dy, nx, ny);
      double err = ns_maxabs(Rp, (size_t)(nx - 2) * (size_t)(ny - 2)) * (p->ly * p->ly) / p->psc;  /* m*ly^2/psc: the power binds first */
      if (nh < nhist_cap) {
        if (err_hist) err_hist[nh] = err;
        if (iter_hist) iter_hist[nh] = iter;
      }
      nh++;
      if (err < p->eps_it || !isfinite(err)) { iter++; break; }
    }
  }
  if (nhist_out) *nhist_out = nh;
  return iter - 1;
}

/* Workspace-carrying state for whole time steps (NavierStokes.jl:58-74). */
typedef struct {
  double *Pr, *dPrdtau, *C, *C_o, *txx, *tyy, *txy, *Vx, *Vy, *Vx_o, *Vy_o, *divV, *Rp, *Vprof;
} ns_state;

/* One pass of the body of the time loop, NavierStokes.jl:79-100. */
int ns_step(ns_state *s, const ns_params *p, double *err_hist, int *iter_hist,
            int nhist_cap, int *nhist_out) {
  const int nx = p->nx, ny = p->ny;
  ns_update_tau(s->txx, s->tyy, s->txy, s->Vx, s->Vy, p->mu, p->dx, p->dy, nx, ny);
  ns_predict_V(s->Vx, s->Vy, s->txx, s->tyy, s->txy, p->rho, p->g, p->dt, p->dx, p->dy, nx, ny);
  ns_set_sphere(s->C, s->Vx, s->Vy, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx, p->ly, p->dx, p->dy, nx, ny, 0);
  ns_update_divV(s->divV, s->Vx, s->Vy, p->dx, p->dy, nx, ny);
  int it = ns_poisson(s->Pr, s->dPrdtau, s->Rp, s->divV, p, p->niter, p->nchk,
                      err_hist, iter_hist, nhist_cap, nhist_out);
  ns_correct_V(s->Vx, s->Vy, s->Pr, p->dt, p->rho, p->dx, p->dy, nx, ny);
  ns_set_sphere(s->C, s->Vx, s->Vy, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx, p->ly, p->dx, p->dy, nx, ny, 0);
  ns_set_bc_Vel(s->Vx, s->Vy, s->Vprof, nx, ny, p->full_rows);
  memcpy(s->Vx_o, s->Vx, sizeof(double) * (size_t)(nx + 1) * ny);
  memcpy(s->Vy_o, s->Vy, sizeof(double) * (size_t)nx * (ny + 1));
  memcpy(s->C_o, s->C, sizeof(double) * (size_t)nx * ny);
  ns_advect(s->Vx, s->Vx_o, s->Vy, s->Vy_o, s->C, s->C_o, p->dt, p->dx, p->dy, nx, ny);
  return it;
}
