/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the reference's bubble/drop
 * solvers: the constant-viscosity code1 family that scripts/DNS.jl transcribes, and the
 * front-tracking code3 (scripts/code3_fun.jl).  Nothing in the product path may link or
 * call this file.
 *
 * Sources followed (file:line of /root/reference):
 *   code1 / DNS:  scripts/code1.m:33-101 is the ground truth (Tryggvason's code1);
 *                 scripts/code1_loop_fun.jl:7-115 states the same kernels as Julia
 *                 functions and scripts/DNS.jl:118-220 gives the kernel/argument split.
 *                 DNS.jl itself does not run as written (temp_uvel!/temp_vvel! use
 *                 undefined variables, three statements are cut short by line breaks);
 *                 what is restated here is the algorithm all three files intend,
 *                 i.e. code1.m, including its "+" in the density flux (code1.m:90-93;
 *                 code1_loop_fun.jl:106-109 has "-").
 *   code3:        scripts/code3_fun.jl:8-293 (functions) and :354-494 (time loop,
 *                 surface tension spreading, RK2 averaging, front regridding).
 *
 * Pinning: the reference's kernel tests give closed-form known answers for
 * temp_uvel_adv! (tests/test_temp_uvel_gpu.jl:47-65), set_density_viscosity!
 * (tests/test_set_density.jl:43-72) and set_tangential! (tests/
 * test_tangential_velocity.jl:84-87) and move_front! (scripts/tests/
 * test_move_front_gpu.jl:74-127, rigid rotation), and input recipes without stored outputs
 * for advect_front! and the pressure solvers (scripts/tests/test_advect_front_gpu.jl:80-123,
 * test_solve_pressure_gpu.jl:142-180); tests/test_oracle_bubble.py checks this file against
 * them.  No stored outputs exist for anything else: PARITY UNPINNED beyond those.
 *
 * Arrays are Julia/MATLAB column-major, 1-based: A[i,j] -> a[(j-1)*ld + (i-1)].
 * Julia semantics: x^2 is x*x, a/b*c is (a/b)*c, no FMA contraction
 * (build with -ffp-contract=off).
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#define B2(a, ld, i, j) (a)[(size_t)((j) - 1) * (size_t)(ld) + (size_t)((i) - 1)]
#define SQ(x) ((x) * (x))

/* code1.m:25-28, DNS.jl:128-137, code3_fun.jl:8-18 (interior cells only; the loop bounds of
 * code3_fun.jl:9-10 run one past the array and the function is never called there). */
void bub_set_density(double *r, double *m, double xc, double yc, const double *x,
                     const double *y, double rad, double rho2, double m2, int nx, int ny) {
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx + 1; i++)
      if (SQ(x[i - 1] - xc) + SQ(y[j - 1] - yc) < SQ(rad)) {
        B2(r, nx + 2, i, j) = rho2;
        if (m) B2(m, nx + 2, i, j) = m2;
      }
}

/* code1.m:33-34, code3_fun.jl:28-35 */
void bub_set_tangential(double *u, double *v, double usouth, double unorth, double vwest,
                        double veast, int nx, int ny) {
  for (int i = 1; i <= nx + 1; i++) {
    B2(u, nx + 1, i, 1) = 2 * usouth - B2(u, nx + 1, i, 2);
    B2(u, nx + 1, i, ny + 2) = 2 * unorth - B2(u, nx + 1, i, ny + 1);
  }
  for (int j = 1; j <= ny + 1; j++) {
    B2(v, nx + 2, 1, j) = 2 * vwest - B2(v, nx + 2, 2, j);
    B2(v, nx + 2, nx + 2, j) = 2 * veast - B2(v, nx + 2, nx + 1, j);
  }
}

#define U(i, j) B2(u, nx + 1, i, j)
#define V(i, j) B2(v, nx + 2, i, j)
#define R(i, j) B2(r, nx + 2, i, j)
#define M(i, j) B2(m, nx + 2, i, j)

static inline double adv_u(const double *u, const double *v, double dx, double dy, int nx, int i, int j) {
  return -0.25 * ((SQ(U(i + 1, j) + U(i, j)) - SQ(U(i, j) + U(i - 1, j))) / dx +
                  ((U(i, j + 1) + U(i, j)) * (V(i + 1, j) + V(i, j)) -
                   (U(i, j) + U(i, j - 1)) * (V(i + 1, j - 1) + V(i, j - 1))) / dy);
}
static inline double adv_v(const double *u, const double *v, double dx, double dy, int nx, int i, int j) {
  return -0.25 * (((U(i, j + 1) + U(i, j)) * (V(i + 1, j) + V(i, j)) -
                   (U(i - 1, j + 1) + U(i - 1, j)) * (V(i, j) + V(i - 1, j))) / dx +
                  (SQ(V(i, j + 1) + V(i, j)) - SQ(V(i, j) + V(i, j - 1))) / dy);
}

/* code1.m:36-43, code1_loop_fun.jl:26-36 */
void bub_temp_uvel(double *ut, const double *u, const double *v, const double *r, double dt,
                   double dx, double dy, double gx, double m0, int nx, int ny) {
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx; i++)
      B2(ut, nx + 1, i, j) =
          U(i, j) + dt * (adv_u(u, v, dx, dy, nx, i, j) +
                          m0 / (0.5 * (R(i + 1, j) + R(i, j))) *
                              ((U(i + 1, j) - 2 * U(i, j) + U(i - 1, j)) / SQ(dx) +
                               (U(i, j + 1) - 2 * U(i, j) + U(i, j - 1)) / SQ(dy)) +
                          gx);
}

/* code1.m:45-52, code1_loop_fun.jl:38-49 */
void bub_temp_vvel(double *vt, const double *u, const double *v, const double *r, double dt,
                   double dx, double dy, double gy, double m0, int nx, int ny) {
  for (int j = 2; j <= ny; j++)
    for (int i = 2; i <= nx + 1; i++)
      B2(vt, nx + 2, i, j) =
          V(i, j) + dt * (adv_v(u, v, dx, dy, nx, i, j) +
                          m0 / (0.5 * (R(i, j + 1) + R(i, j))) *
                              ((V(i + 1, j) - 2 * V(i, j) + V(i - 1, j)) / SQ(dx) +
                               (V(i, j + 1) - 2 * V(i, j) + V(i, j - 1)) / SQ(dy)) +
                          gy);
}

/* code3_fun.jl:37-49 */
void bub_temp_uvel_adv(double *ut, const double *u, const double *v, const double *r, double dt,
                       double dx, double dy, const double *fx, double gx, double rro, int nx,
                       int ny) {
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx; i++) {
      double rb = 0.5 * (R(i + 1, j) + R(i, j));
      B2(ut, nx + 1, i, j) = U(i, j) + dt * (adv_u(u, v, dx, dy, nx, i, j) +
                                             B2(fx, nx + 2, i, j) / rb - (1.0 - rro / rb) * gx);
    }
}

/* code3_fun.jl:51-62 */
void bub_temp_vvel_adv(double *vt, const double *u, const double *v, const double *r, double dt,
                       double dx, double dy, const double *fy, double gy, double rro, int nx,
                       int ny) {
  for (int j = 2; j <= ny; j++)
    for (int i = 2; i <= nx + 1; i++) {
      double rb = 0.5 * (R(i, j + 1) + R(i, j));
      B2(vt, nx + 2, i, j) = V(i, j) + dt * (adv_v(u, v, dx, dy, nx, i, j) +
                                             B2(fy, nx + 2, i, j) / rb - (1.0 - rro / rb) * gy);
    }
}

/* code3_fun.jl:64-76.  As written there, only the second (cross) term is divided by the
 * mean density:  ut += dt*( T1 + (1/dy)*(X - Y)/rbar ). */
void bub_temp_uvel_diff(double *ut, const double *u, const double *v, const double *r, double dt,
                        double dx, double dy, const double *m, int nx, int ny) {
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx; i++) {
      double t1 = (1. / dx) * 2. * (M(i + 1, j) * (1. / dx) * (U(i + 1, j) - U(i, j)) -
                                    M(i, j) * (1. / dx) * (U(i, j) - U(i - 1, j)));
      double X = 0.25 * (M(i, j) + M(i + 1, j) + M(i + 1, j + 1) + M(i, j + 1)) *
                 ((1. / dy) * (U(i, j + 1) - U(i, j)) + (1. / dx) * (V(i + 1, j) - V(i, j)));
      double Y = 0.25 * (M(i, j) + M(i + 1, j) + M(i + 1, j - 1) + M(i, j - 1)) *
                 ((1. / dy) * (U(i, j) - U(i, j - 1)) + (1. / dx) * (V(i + 1, j - 1) - V(i, j - 1)));
      double t2 = (1. / dy) * (X - Y) / (0.5 * (R(i + 1, j) + R(i, j)));
      B2(ut, nx + 1, i, j) = B2(ut, nx + 1, i, j) + dt * (t1 + t2);
    }
}

/* code3_fun.jl:78-90.  Here the whole bracket is divided:  vt += dt*( T1 + T2 )/rbar. */
void bub_temp_vvel_diff(double *vt, const double *u, const double *v, const double *r, double dt,
                        double dx, double dy, const double *m, int nx, int ny) {
  for (int j = 2; j <= ny; j++)
    for (int i = 2; i <= nx + 1; i++) {
      double X = 0.25 * (M(i, j) + M(i + 1, j) + M(i + 1, j + 1) + M(i, j + 1)) *
                 ((1.0 / dy) * (U(i, j + 1) - U(i, j)) + (1.0 / dx) * (V(i + 1, j) - V(i, j)));
      double Y = 0.25 * (M(i, j) + M(i, j + 1) + M(i - 1, j + 1) + M(i - 1, j)) *
                 ((1.0 / dy) * (U(i - 1, j + 1) - U(i - 1, j)) + (1.0 / dx) * (V(i, j) - V(i - 1, j)));
      double t1 = (1.0 / dx) * (X - Y);
      double t2 = (1.0 / dy) * 2.0 * (M(i, j + 1) * (1.0 / dy) * (V(i, j + 1) - V(i, j)) -
                                      M(i, j) * (1.0 / dy) * (V(i, j) - V(i, j - 1)));
      B2(vt, nx + 2, i, j) = B2(vt, nx + 2, i, j) + dt * (t1 + t2) / (0.5 * (R(i, j + 1) + R(i, j)));
    }
}

/* code1.m:55-57, code3_fun.jl:415-420: rt = r with lrg on the four borders */
void bub_set_boundary(double *rt, const double *r, double lrg, int nx, int ny) {
  memcpy(rt, r, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  for (int i = 1; i <= nx + 2; i++) B2(rt, nx + 2, i, 1) = B2(rt, nx + 2, i, ny + 2) = lrg;
  for (int j = 1; j <= ny + 2; j++) B2(rt, nx + 2, 1, j) = B2(rt, nx + 2, nx + 2, j) = lrg;
}

#define RT(i, j) B2(rt, nx + 2, i, j)
#define P(i, j) B2(p, nx + 2, i, j)

/* code1.m:59-65, code3_fun.jl:92-102 */
void bub_compute_sources(double *tmp1, double *tmp2, const double *rt, const double *ut,
                         const double *vt, double dt, double dx, double dy, int nx, int ny) {
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx + 1; i++) {
      B2(tmp1, nx + 2, i, j) = (0.5 / dt) * ((B2(ut, nx + 1, i, j) - B2(ut, nx + 1, i - 1, j)) / dx +
                                             (B2(vt, nx + 2, i, j) - B2(vt, nx + 2, i, j - 1)) / dy);
      B2(tmp2, nx + 2, i, j) =
          1.0 / ((1.0 / dx) * (1.0 / (dx * (RT(i + 1, j) + RT(i, j))) + 1.0 / (dx * (RT(i - 1, j) + RT(i, j)))) +
                 (1.0 / dy) * (1.0 / (dy * (RT(i, j + 1) + RT(i, j))) + 1.0 / (dy * (RT(i, j - 1) + RT(i, j)))));
    }
}

static double max_abs_diff(const double *a, const double *b, size_t n) {
  /* maximum(abs.(a .- b)): NaN-propagating like Julia's maximum */
  double mx = 0.0;
  int nan = 0;
  for (size_t k = 0; k < n; k++) {
    double d = fabs(a[k] - b[k]);
    if (d != d) nan = 1;
    if (d > mx) mx = d;
  }
  return nan ? NAN : mx;
}

/* code1.m:68-78, code3_fun.jl:104-124: lexicographic in-place SOR (i outer, j inner), stop
 * when max|old-new| < maxError.  Returns the number of sweeps performed. */
int bub_solve_pressure(double *p, const double *tmp1, const double *tmp2, const double *rt,
                       double beta, double dx, double dy, int nx, int ny, int maxit,
                       double maxError, double *last_change) {
  size_t n = (size_t)(nx + 2) * (ny + 2);
  double *old = (double *)malloc(n * sizeof(double));
  int it;
  double ch = 0.0;
  for (it = 1; it <= maxit; it++) {
    memcpy(old, p, n * sizeof(double));
    for (int i = 2; i <= nx + 1; i++)
      for (int j = 2; j <= ny + 1; j++)
        P(i, j) = (1.0 - beta) * P(i, j) +
                  beta * B2(tmp2, nx + 2, i, j) *
                      ((1.0 / dx) * (P(i + 1, j) / (dx * (RT(i + 1, j) + RT(i, j))) +
                                     P(i - 1, j) / (dx * (RT(i - 1, j) + RT(i, j)))) +
                       (1.0 / dy) * (P(i, j + 1) / (dy * (RT(i, j + 1) + RT(i, j))) +
                                     P(i, j - 1) / (dy * (RT(i, j - 1) + RT(i, j)))) -
                       B2(tmp1, nx + 2, i, j));
    ch = max_abs_diff(old, p, n);
    if (ch < maxError) break;
  }
  free(old);
  if (last_change) *last_change = ch;
  return it > maxit ? maxit : it;
}

/* code3_perf_gpu.jl:129-183 (red then black colour sweep) with its loop :609-617 (and
 * scripts/tests/test_solve_pressure_gpu.jl:6-61): in-place SOR where all cells with i+j even
 * are updated first, then all cells with i+j odd; stop when max|old-new| < maxError.  In double
 * precision (the reference's CUDA.zeros arrays are Float32).  Not the lexicographic order of
 * bub_solve_pressure: results differ from it by the iteration error. */
int bub_solve_pressure_rb(double *p, const double *tmp1, const double *tmp2, const double *rt,
                          double beta, double dx, double dy, int nx, int ny, int maxit,
                          double maxError, double *last_change) {
  size_t n = (size_t)(nx + 2) * (ny + 2);
  double *old = (double *)malloc(n * sizeof(double));
  int it;
  double ch = 0.0;
  for (it = 1; it <= maxit; it++) {
    memcpy(old, p, n * sizeof(double));
    for (int colour = 0; colour < 2; colour++)
      for (int j = 2; j <= ny + 1; j++)
        for (int i = 2; i <= nx + 1; i++) {
          if ((i + j) % 2 != colour) continue;
          P(i, j) = (1.0 - beta) * P(i, j) +
                    beta * B2(tmp2, nx + 2, i, j) *
                        ((1.0 / dx) * (P(i + 1, j) / (dx * (RT(i + 1, j) + RT(i, j))) +
                                       P(i - 1, j) / (dx * (RT(i - 1, j) + RT(i, j)))) +
                         (1.0 / dy) * (P(i, j + 1) / (dy * (RT(i, j + 1) + RT(i, j))) +
                                       P(i, j - 1) / (dy * (RT(i, j - 1) + RT(i, j)))) -
                         B2(tmp1, nx + 2, i, j));
        }
    ch = max_abs_diff(old, p, n);
    if (ch < maxError) break;
  }
  free(old);
  if (last_change) *last_change = ch;
  return it > maxit ? maxit : it;
}

/* code1_perf_gpu.jl:98-118 (update_dPdtau!, update_p!) with its loop :228-242: damped
 * pseudo-transient iteration for the variable-density pressure equation; every iteration is
 * two whole-array maps.  dPdtau persists across calls like the reference's array.  In double
 * precision.  Returns the iterations performed. */
int bub_pt_pressure(double *p, double *dPdtau, const double *tmp1, const double *rt, double dtau,
                    double damp, double dx, double dy, int nx, int ny, int maxit, double maxError,
                    double *last_change) {
  size_t n = (size_t)(nx + 2) * (ny + 2);
  double *old = (double *)malloc(n * sizeof(double));
  int it;
  double ch = 0.0;
  for (it = 1; it <= maxit; it++) {
    memcpy(old, p, n * sizeof(double));
    for (int j = 2; j <= ny + 1; j++)
      for (int i = 2; i <= nx + 1; i++)
        B2(dPdtau, nx + 2, i, j) =
            B2(dPdtau, nx + 2, i, j) * (1.0 - damp) +
            dtau * ((1.0 / dx) * ((P(i + 1, j) - P(i, j)) / (dx * (RT(i + 1, j) + RT(i, j))) -
                                  (P(i, j) - P(i - 1, j)) / (dx * (RT(i - 1, j) + RT(i, j)))) +
                    (1.0 / dy) * ((P(i, j + 1) - P(i, j)) / (dy * (RT(i, j + 1) + RT(i, j))) -
                                  (P(i, j) - P(i, j - 1)) / (dy * (RT(i, j - 1) + RT(i, j)))) -
                    B2(tmp1, nx + 2, i, j));
    for (int j = 2; j <= ny + 1; j++)
      for (int i = 2; i <= nx + 1; i++) P(i, j) = P(i, j) + dtau * B2(dPdtau, nx + 2, i, j);
    ch = max_abs_diff(old, p, n);
    if (ch < maxError) break;
  }
  free(old);
  if (last_change) *last_change = ch;
  return it > maxit ? maxit : it;
}

/* code1.m:80-86, code3_fun.jl:126-143 */
void bub_correct_uvel(double *u, const double *ut, const double *p, const double *r, double dt,
                      double dx, int nx, int ny) {
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx; i++)
      U(i, j) = B2(ut, nx + 1, i, j) - dt * (2.0 / dx) * (P(i + 1, j) - P(i, j)) / (R(i + 1, j) + R(i, j));
}
void bub_correct_vvel(double *v, const double *vt, const double *p, const double *r, double dt,
                      double dy, int nx, int ny) {
  for (int j = 2; j <= ny; j++)
    for (int i = 2; i <= nx + 1; i++)
      V(i, j) = B2(vt, nx + 2, i, j) - dt * (2.0 / dy) * (P(i, j + 1) - P(i, j)) / (R(i, j + 1) + R(i, j));
}

/* code1.m:88-96 (conservative centred flux, "+", and m0 diffusion) */
void bub_advect_density(double *r, const double *ro, const double *u, const double *v, double dt,
                        double dx, double dy, double m0, int nx, int ny) {
#define RO(i, j) B2(ro, nx + 2, i, j)
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx + 1; i++)
      R(i, j) = RO(i, j) -
                (0.5 * dt / dx) * (U(i, j) * (RO(i + 1, j) + RO(i, j)) - U(i - 1, j) * (RO(i - 1, j) + RO(i, j))) -
                (0.5 * dt / dy) * (V(i, j) * (RO(i, j + 1) + RO(i, j)) - V(i, j - 1) * (RO(i, j - 1) + RO(i, j))) +
                (m0 * dt / dx / dx) * (RO(i + 1, j) - 2.0 * RO(i, j) + RO(i - 1, j)) +
                (m0 * dt / dy / dy) * (RO(i, j + 1) - 2.0 * RO(i, j) + RO(i, j - 1));
#undef RO
}

/* code1.m:100-101, DNS.jl:214-220: velocities at the cell corners for output */
void bub_velocity(double *uu, double *vv, const double *u, const double *v, int nx, int ny) {
  for (int j = 1; j <= ny + 1; j++)
    for (int i = 1; i <= nx + 1; i++) {
      B2(uu, nx + 1, i, j) = 0.5 * (U(i, j + 1) + U(i, j));
      B2(vv, nx + 1, i, j) = 0.5 * (V(i + 1, j) + V(i, j));
    }
}

/* One time step of code1.m:31-96.  Returns the SOR sweeps used. */
int bub_code1_step(double *u, double *v, double *p, double *r, double *ut, double *vt,
                   double *tmp1, double *tmp2, double *rt, double *ro, double dt, double dx,
                   double dy, double gx, double gy, double m0, double usouth, double unorth,
                   double vwest, double veast, double beta, int maxit, double maxError, int nx,
                   int ny) {
  bub_set_tangential(u, v, usouth, unorth, vwest, veast, nx, ny);
  bub_temp_uvel(ut, u, v, r, dt, dx, dy, gx, m0, nx, ny);
  bub_temp_vvel(vt, u, v, r, dt, dx, dy, gy, m0, nx, ny);
  bub_set_boundary(rt, r, 1000.0, nx, ny);
  bub_compute_sources(tmp1, tmp2, rt, ut, vt, dt, dx, dy, nx, ny);
  int it = bub_solve_pressure(p, tmp1, tmp2, rt, beta, dx, dy, nx, ny, maxit, maxError, NULL);
  bub_correct_uvel(u, ut, p, r, dt, dx, nx, ny);
  bub_correct_vvel(v, vt, p, r, dt, dy, nx, ny);
  memcpy(ro, r, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  bub_advect_density(r, ro, u, v, dt, dx, dy, m0, nx, ny);
  return it;
}

/* ------------------------------- code3: the front ---------------------------------- */

/* code3_fun.jl:372-378: unit tangents of the front segments, l = 1..Nf+1, and the wrap */
void bub_front_tangents(double *tx, double *ty, const double *xf, const double *yf, int Nf) {
  for (int l = 1; l <= Nf + 1; l++) {
    double ds = sqrt(SQ(xf[l] - xf[l - 1]) + SQ(yf[l] - yf[l - 1])); /* xf[l+1]-xf[l], 0-based here */
    tx[l - 1] = (xf[l] - xf[l - 1]) / ds;
    ty[l - 1] = (yf[l] - yf[l - 1]) / ds;
  }
  tx[Nf + 1] = tx[1];
  ty[Nf + 1] = ty[1];
}

/* Bilinear spreading of one front quantity onto the x-face grid (fx) and the y-face grid
 * (fy); code3_fun.jl:384-400 and :203-219 share it.  Accumulation is sequential in l. */
static void spread_point(double *fx, double *fy, double xl, double yl, double nfx, double nfy,
                         double dx, double dy, int nx) {
  int ip = (int)floor(xl / dx) + 1;
  int jp = (int)floor((yl + 0.5 * dy) / dy) + 1;
  double ax = xl / dx - ip + 1;
  double ay = (yl + 0.5 * dy) / dy - jp + 1;
  B2(fx, nx + 2, ip, jp) += (1.0 - ax) * (1.0 - ay) * nfx / dx / dy;
  B2(fx, nx + 2, ip + 1, jp) += ax * (1.0 - ay) * nfx / dx / dy;
  B2(fx, nx + 2, ip, jp + 1) += (1.0 - ax) * ay * nfx / dx / dy;
  B2(fx, nx + 2, ip + 1, jp + 1) += ax * ay * nfx / dx / dy;
  ip = (int)floor((xl + 0.5 * dx) / dx) + 1;
  jp = (int)floor(yl / dy) + 1;
  ax = (xl + 0.5 * dx) / dx - ip + 1;
  ay = yl / dy - jp + 1;
  B2(fy, nx + 2, ip, jp) += (1.0 - ax) * (1.0 - ay) * nfy / dx / dy;
  B2(fy, nx + 2, ip + 1, jp) += ax * (1.0 - ay) * nfy / dx / dy;
  B2(fy, nx + 2, ip, jp + 1) += (1.0 - ax) * ay * nfy / dx / dy;
  B2(fy, nx + 2, ip + 1, jp + 1) += ax * ay * nfy / dx / dy;
}

/* code3_fun.jl:365-366,380-401: fx, fy zeroed, then the surface tension of every front
 * point spread to the grid. */
void bub_surface_tension(double *fx, double *fy, const double *xf, const double *yf,
                         const double *tx, const double *ty, double sigma, double dx, double dy,
                         int nx, int ny, int Nf) {
  memset(fx, 0, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  memset(fy, 0, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  for (int l = 2; l <= Nf + 1; l++) {
    double nfx = sigma * (tx[l - 1] - tx[l - 2]);
    double nfy = sigma * (ty[l - 1] - ty[l - 2]);
    spread_point(fx, fy, xf[l - 1], yf[l - 1], nfx, nfy, dx, dy, nx);
  }
}

/* code3_fun.jl:433-435 + :197-222: fx, fy zeroed, then the density jump distributed */
void bub_distribute(double *fx, double *fy, const double *xf, const double *yf, double rho1,
                    double rho2, double dx, double dy, int nx, int ny, int Nf) {
  memset(fx, 0, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  memset(fy, 0, sizeof(double) * (size_t)(nx + 2) * (ny + 2));
  for (int l = 2; l <= Nf + 1; l++) {
    double nfx = -0.5 * (yf[l] - yf[l - 2]) * (rho2 - rho1);
    double nfy = 0.5 * (xf[l] - xf[l - 2]) * (rho2 - rho1);
    spread_point(fx, fy, xf[l - 1], yf[l - 1], nfx, nfy, dx, dy, nx);
  }
}

/* code3_fun.jl:144-179 */
void bub_advect_front(double *uf, double *vf, const double *xf, const double *yf, const double *u,
                      const double *v, double dx, double dy, int nx, int Nf) {
  for (int l = 2; l <= Nf + 1; l++) {
    double xl = xf[l - 1], yl = yf[l - 1];
    int ip = (int)floor(xl / dx) + 1;
    int jp = (int)floor((yl + 0.5 * dy) / dy) + 1;
    double ax = xl / dx - ip + 1;
    double ay = (yl + 0.5 * dy) / dy - jp + 1;
    double v1 = (1.0 - ax) * (1.0 - ay) * U(ip, jp);
    double v2 = ax * (1.0 - ay) * U(ip + 1, jp);
    double v3 = (1.0 - ax) * ay * U(ip, jp + 1);
    double v4 = ax * ay * U(ip + 1, jp + 1);
    uf[l - 1] = v1 + v2 + v3 + v4;
    ip = (int)floor((xl + 0.5 * dx) / dx) + 1;
    jp = (int)floor(yl / dy) + 1;
    ax = (xl + 0.5 * dx) / dx - ip + 1;
    ay = yl / dy - jp + 1;
    v1 = (1.0 - ax) * (1.0 - ay) * V(ip, jp);
    v2 = ax * (1.0 - ay) * V(ip + 1, jp);
    v3 = (1.0 - ax) * ay * V(ip, jp + 1);
    v4 = ax * ay * V(ip + 1, jp + 1);
    vf[l - 1] = v1 + v2 + v3 + v4;
  }
}

/* code3_fun.jl:181-195.  Returns the number of points that left the box (the reference
 * prints a line for each). */
int bub_move_front(double *xf, double *yf, const double *uf, const double *vf, int Nf, double dt,
                   double Lx, double Ly) {
  int out = 0;
  for (int i = 2; i <= Nf + 1; i++) {
    xf[i - 1] = xf[i - 1] + dt * uf[i - 1];
    yf[i - 1] = yf[i - 1] + dt * vf[i - 1];
    if (xf[i - 1] < 0.0 || xf[i - 1] > Lx || yf[i - 1] < 0.0 || yf[i - 1] > Ly) out++;
  }
  xf[0] = xf[Nf];
  yf[0] = yf[Nf];
  xf[Nf + 1] = xf[1];
  yf[Nf + 1] = yf[1];
  return out;
}

/* code3_fun.jl:224-240: lexicographic SOR for the density; returns the sweeps performed */
int bub_construct_den(double *r, const double *fx, const double *fy, double beta, int maxit,
                      double maxError, double dx, double dy, int nx, int ny) {
  size_t n = (size_t)(nx + 2) * (ny + 2);
  double *old = (double *)malloc(n * sizeof(double));
  int it;
  for (it = 1; it <= maxit; it++) {
    memcpy(old, r, n * sizeof(double));
    for (int i = 2; i <= nx + 1; i++)
      for (int j = 2; j <= ny + 1; j++)
        R(i, j) = (1.0 - beta) * R(i, j) +
                  beta * (0.25 * (R(i + 1, j) + R(i - 1, j) + R(i, j + 1) + R(i, j - 1) +
                                  dx * B2(fx, nx + 2, i - 1, j) - dx * B2(fx, nx + 2, i, j) +
                                  dy * B2(fy, nx + 2, i, j - 1) - dy * B2(fy, nx + 2, i, j)));
    if (max_abs_diff(old, r, n) < maxError) break;
  }
  free(old);
  return it > maxit ? maxit : it;
}

/* code3_fun.jl:440-441 + :242-249: m = m1 everywhere, then the interior from the density */
void bub_update_visco(const double *r, double *m, double m1, double m2, double rho1, double rho2,
                      int nx, int ny) {
  for (size_t k = 0; k < (size_t)(nx + 2) * (ny + 2); k++) m[k] = 0.0 + m1;
  for (int j = 2; j <= ny + 1; j++)
    for (int i = 2; i <= nx + 1; i++)
      M(i, j) = m1 + (m2 - m1) * (R(i, j) - rho1) / (rho2 - rho1);
}

/* code3_fun.jl:455-494: front regridding (sequential: it reads points it has just written).
 * xf, yf have capacity cap; returns the new Nf, or -1 if the capacity would be exceeded
 * (the reference resizes its vectors at that point). */
int bub_regrid_front(double *xf, double *yf, double dx, double dy, int Nf, int cap) {
  double *xo = (double *)malloc(sizeof(double) * (size_t)(Nf + 2));
  double *yo = (double *)malloc(sizeof(double) * (size_t)(Nf + 2));
  memcpy(xo, xf, sizeof(double) * (size_t)(Nf + 2));
  memcpy(yo, yf, sizeof(double) * (size_t)(Nf + 2));
  int j = 1, ok = 1;
  for (int l = 2; l <= Nf + 1 && ok; l++) {
    double ds = sqrt(SQ((xo[l - 1] - xf[j - 1]) / dx) + SQ((yo[l - 1] - yf[j - 1]) / dy));
    if (ds > 0.5) {
      j += 1;
      if (j + 2 > cap) { ok = 0; break; }
      xf[j - 1] = 0.5 * (xo[l - 1] + xf[j - 2]);
      yf[j - 1] = 0.5 * (yo[l - 1] + yf[j - 2]);
      j += 1;
      xf[j - 1] = xo[l - 1];
      yf[j - 1] = yo[l - 1];
    } else if (ds < 0.25) {
      /* drop the point */
    } else {
      j += 1;
      if (j + 1 > cap) { ok = 0; break; }
      xf[j - 1] = xo[l - 1];
      yf[j - 1] = yo[l - 1];
    }
  }
  free(xo);
  free(yo);
  if (!ok || j + 1 > cap) return -1;
  int Nn = j - 1;
  xf[0] = xf[Nn];
  yf[0] = yf[Nn];
  xf[Nn + 1] = xf[1];
  yf[Nn + 1] = yf[1];
  return Nn;
}

/* State of the code3 time loop (code3_fun.jl:311-349). */
typedef struct {
  double *u, *v, *p, *ut, *vt, *tmp1, *tmp2, *fx, *fy, *un, *vn, *r, *m, *rn, *mn, *rt;
  double *xf, *yf, *xfn, *yfn, *uf, *vf, *tx, *ty; /* capacity cap */
  int Nf, cap;
} c3_state;

typedef struct {
  int nx, ny, maxit;
  double Lx, Ly, gx, gy, rho1, rho2, m1, m2, sigma, rro, usouth, unorth, vwest, veast;
  double dt, dx, dy, maxError, beta;
} c3_params;

/* One pass of the time loop body, code3_fun.jl:356-494.  sweeps[0..3] receive the SOR sweep
 * counts (pressure, density) of the two substeps.  Returns 0, -1 if the front outgrew cap, or -2
 * as soon as a front point has left the box: the reference prints a message there
 * (code3_fun.jl:186-188) and its next array access by that point is a BoundsError; this
 * restatement indexes unchecked, so it stops instead. */
int bub_code3_step(c3_state *s, const c3_params *q, int *sweeps) {
  const int nx = q->nx, ny = q->ny;
  const size_t nu = (size_t)(nx + 1) * (ny + 2), nv = (size_t)(nx + 2) * (ny + 1),
               nc = (size_t)(nx + 2) * (ny + 2);
  memcpy(s->un, s->u, nu * sizeof(double));
  memcpy(s->vn, s->v, nv * sizeof(double));
  memcpy(s->rn, s->r, nc * sizeof(double));
  memcpy(s->mn, s->m, nc * sizeof(double));
  memcpy(s->xfn, s->xf, sizeof(double) * (size_t)(s->Nf + 2));
  memcpy(s->yfn, s->yf, sizeof(double) * (size_t)(s->Nf + 2));
  for (int sub = 0; sub < 2; sub++) {
    bub_front_tangents(s->tx, s->ty, s->xf, s->yf, s->Nf);
    bub_surface_tension(s->fx, s->fy, s->xf, s->yf, s->tx, s->ty, q->sigma, q->dx, q->dy, nx, ny, s->Nf);
    bub_set_tangential(s->u, s->v, q->usouth, q->unorth, q->vwest, q->veast, nx, ny);
    bub_temp_uvel_adv(s->ut, s->u, s->v, s->r, q->dt, q->dx, q->dy, s->fx, q->gx, q->rro, nx, ny);
    bub_temp_vvel_adv(s->vt, s->u, s->v, s->r, q->dt, q->dx, q->dy, s->fy, q->gy, q->rro, nx, ny);
    bub_temp_uvel_diff(s->ut, s->u, s->v, s->r, q->dt, q->dx, q->dy, s->m, nx, ny);
    bub_temp_vvel_diff(s->vt, s->u, s->v, s->r, q->dt, q->dx, q->dy, s->m, nx, ny);
    bub_set_boundary(s->rt, s->r, 1000.0, nx, ny);
    bub_compute_sources(s->tmp1, s->tmp2, s->rt, s->ut, s->vt, q->dt, q->dx, q->dy, nx, ny);
    int a = bub_solve_pressure(s->p, s->tmp1, s->tmp2, s->rt, q->beta, q->dx, q->dy, nx, ny, q->maxit, q->maxError, NULL);
    bub_correct_uvel(s->u, s->ut, s->p, s->r, q->dt, q->dx, nx, ny);
    bub_correct_vvel(s->v, s->vt, s->p, s->r, q->dt, q->dy, nx, ny);
    bub_advect_front(s->uf, s->vf, s->xf, s->yf, s->u, s->v, q->dx, q->dy, nx, s->Nf);
    bub_move_front(s->xf, s->yf, s->uf, s->vf, s->Nf, q->dt, q->Lx, q->Ly);
    for (int l = 1; l <= s->Nf; l++)  /* (NaN positions count as outside here) */
      if (!(s->xf[l] >= 0.0 && s->xf[l] <= q->Lx && s->yf[l] >= 0.0 && s->yf[l] <= q->Ly)) return -2;
    bub_distribute(s->fx, s->fy, s->xf, s->yf, q->rho1, q->rho2, q->dx, q->dy, nx, ny, s->Nf);
    int b = bub_construct_den(s->r, s->fx, s->fy, q->beta, q->maxit, q->maxError, q->dx, q->dy, nx, ny);
    bub_update_visco(s->r, s->m, q->m1, q->m2, q->rho1, q->rho2, nx, ny);
    if (sweeps) { sweeps[2 * sub] = a; sweeps[2 * sub + 1] = b; }
  }
  /* code3_fun.jl:445-450: second-order averaging */
  for (size_t k = 0; k < nu; k++) s->u[k] = 0.5 * (s->u[k] + s->un[k]);
  for (size_t k = 0; k < nv; k++) s->v[k] = 0.5 * (s->v[k] + s->vn[k]);
  for (size_t k = 0; k < nc; k++) s->r[k] = 0.5 * (s->r[k] + s->rn[k]);
  for (size_t k = 0; k < nc; k++) s->m[k] = 0.5 * (s->m[k] + s->mn[k]);
  for (int k = 0; k < s->Nf + 2; k++) {
    s->xf[k] = 0.5 * (s->xf[k] + s->xfn[k]);
    s->yf[k] = 0.5 * (s->yf[k] + s->yfn[k]);
  }
  int Nn = bub_regrid_front(s->xf, s->yf, q->dx, q->dy, s->Nf, s->cap);
  if (Nn < 0) return -1;
  s->Nf = Nn;
  return 0;
}
