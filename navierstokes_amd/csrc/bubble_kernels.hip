// Bubble / drop kernels: the constant-viscosity code1 family that scripts/DNS.jl transcribes
// (ground truth scripts/code1.m:33-101, Julia statement scripts/code1_loop_fun.jl:7-115,
// kernel split scripts/DNS.jl:118-220) and the front-tracking code3
// (scripts/code3_fun.jl:8-293, time loop :354-494).
//
// Grid arrays are (nx+2) x (ny+2) cell-centred with one ghost layer, u is (nx+1) x (ny+2),
// v is (nx+2) x (ny+1), column-major, i contiguous.  One thread per grid point, 64 consecutive
// i per wavefront row.  The two lexicographic Gauss-Seidel/SOR solvers (pressure, density) run
// in ONE workgroup that sweeps anti-diagonals i+j = const: every cell of a diagonal depends only
// on the two previous diagonals (new values) and the two next ones (old values), exactly the
// operands of the reference's in-place i-outer/j-inner loop, so the result is bit-identical; the
// whole iteration loop with its max|old-new| stopping test stays inside the one launch.
// The front-to-grid spreading is a gather: each grid point scans the front points in ascending
// order and adds the contributions that land on it, which reproduces the reference's
// sequential accumulation order without atomics.
#include <cmath>
#include <cstring>
#include <vector>

#include "nsk_internal.h"

namespace {

#define B2(a, ld, i, j) (a)[(size_t)((j)-1) * (size_t)(ld) + (size_t)((i)-1)]
#define SQ(x) ((x) * (x))
#define U(i, j) B2(u, nx + 1, i, j)
#define V(i, j) B2(v, nx + 2, i, j)
#define R(i, j) B2(r, nx + 2, i, j)
#define M(i, j) B2(m, nx + 2, i, j)
#define RT(i, j) B2(rt, nx + 2, i, j)

constexpr int BX = 64, BY = 4;
inline dim3 grid2(int n1, int n2) { return dim3(nsk_cdiv(n1, BX), nsk_cdiv(n2, BY)); }
inline dim3 block2() { return dim3(BX, BY); }
#define IJ                                           \
  const int i = blockIdx.x * BX + threadIdx.x + 1;   \
  const int j = blockIdx.y * BY + threadIdx.y + 1

typedef const double *__restrict__ cdp;
typedef double *__restrict__ dp;

// code1.m:25-28 / DNS.jl:128-137 / code3_fun.jl:8-18
__global__ void k_set_density(dp r, dp m, double xc, double yc, cdp x, cdp y, double rad,
                              double rho2, double m2, int nx, int ny) {
  IJ;
  if (i < 2 || i > nx + 1 || j < 2 || j > ny + 1) return;
  if (SQ(x[i - 1] - xc) + SQ(y[j - 1] - yc) < SQ(rad)) {
    R(i, j) = rho2;
    if (m) M(i, j) = m2;
  }
}

// code1.m:33-34 / code3_fun.jl:28-35
__device__ __forceinline__ void c_set_tangential(int t, dp u, dp v, double usouth, double unorth, double vwest,
                                 double veast, int nx, int ny) {
  if (t <= nx + 1) {
    U(t, 1) = 2 * usouth - U(t, 2);
    U(t, ny + 2) = 2 * unorth - U(t, ny + 1);
  }
  if (t <= ny + 1) {
    V(1, t) = 2 * vwest - V(2, t);
    V(nx + 2, t) = 2 * veast - V(nx + 1, t);
  }
}
__global__ void k_set_tangential(dp u, dp v, double usouth, double unorth, double vwest, double veast, int nx, int ny) {
  c_set_tangential(blockIdx.x * blockDim.x + threadIdx.x + 1, u, v, usouth, unorth, vwest, veast, nx, ny);
}

__device__ __forceinline__ double adv_u(cdp u, cdp v, double dx, double dy, int nx, int i, int j) {
  return -0.25 * ((SQ(U(i + 1, j) + U(i, j)) - SQ(U(i, j) + U(i - 1, j))) / dx +
                  ((U(i, j + 1) + U(i, j)) * (V(i + 1, j) + V(i, j)) -
                   (U(i, j) + U(i, j - 1)) * (V(i + 1, j - 1) + V(i, j - 1))) / dy);
}
__device__ __forceinline__ double adv_v(cdp u, cdp v, double dx, double dy, int nx, int i, int j) {
  return -0.25 * (((U(i, j + 1) + U(i, j)) * (V(i + 1, j) + V(i, j)) -
                   (U(i - 1, j + 1) + U(i - 1, j)) * (V(i, j) + V(i - 1, j))) / dx +
                  (SQ(V(i, j + 1) + V(i, j)) - SQ(V(i, j) + V(i, j - 1))) / dy);
}

// code1.m:36-52: both temporary velocities in one launch
__device__ __forceinline__ void c_temp_vel(int i, int j, dp ut, dp vt, cdp u, cdp v, cdp r, double dt, double dx, double dy,
                           double gx, double gy, double m0, int nx, int ny) {
  if (ut && i >= 2 && i <= nx && j >= 2 && j <= ny + 1)
    B2(ut, nx + 1, i, j) =
        U(i, j) + dt * (adv_u(u, v, dx, dy, nx, i, j) +
                        m0 / (0.5 * (R(i + 1, j) + R(i, j))) *
                            ((U(i + 1, j) - 2 * U(i, j) + U(i - 1, j)) / SQ(dx) +
                             (U(i, j + 1) - 2 * U(i, j) + U(i, j - 1)) / SQ(dy)) +
                        gx);
  if (vt && i >= 2 && i <= nx + 1 && j >= 2 && j <= ny)
    B2(vt, nx + 2, i, j) =
        V(i, j) + dt * (adv_v(u, v, dx, dy, nx, i, j) +
                        m0 / (0.5 * (R(i, j + 1) + R(i, j))) *
                            ((V(i + 1, j) - 2 * V(i, j) + V(i - 1, j)) / SQ(dx) +
                             (V(i, j + 1) - 2 * V(i, j) + V(i, j - 1)) / SQ(dy)) +
                        gy);
}
__global__ void k_temp_vel(dp ut, dp vt, cdp u, cdp v, cdp r, double dt, double dx, double dy, double gx, double gy, double m0, int nx, int ny) {
  IJ;
  c_temp_vel(i, j, ut, vt, u, v, r, dt, dx, dy, gx, gy, m0, nx, ny);
}

// code3_fun.jl:37-62
__device__ __forceinline__ void c_temp_vel_adv(int i, int j, dp ut, dp vt, cdp u, cdp v, cdp r, double dt, double dx, double dy,
                               cdp fx, cdp fy, double gx, double gy, double rro, int nx, int ny) {
  if (ut && i >= 2 && i <= nx && j >= 2 && j <= ny + 1) {
    const double rb = 0.5 * (R(i + 1, j) + R(i, j));
    B2(ut, nx + 1, i, j) = U(i, j) + dt * (adv_u(u, v, dx, dy, nx, i, j) + B2(fx, nx + 2, i, j) / rb -
                                           (1.0 - rro / rb) * gx);
  }
  if (vt && i >= 2 && i <= nx + 1 && j >= 2 && j <= ny) {
    const double rb = 0.5 * (R(i, j + 1) + R(i, j));
    B2(vt, nx + 2, i, j) = V(i, j) + dt * (adv_v(u, v, dx, dy, nx, i, j) + B2(fy, nx + 2, i, j) / rb -
                                           (1.0 - rro / rb) * gy);
  }
}
__global__ void k_temp_vel_adv(dp ut, dp vt, cdp u, cdp v, cdp r, double dt, double dx, double dy, cdp fx, cdp fy, double gx, double gy, double rro, int nx, int ny) {
  IJ;
  c_temp_vel_adv(i, j, ut, vt, u, v, r, dt, dx, dy, fx, fy, gx, gy, rro, nx, ny);
}

// code3_fun.jl:64-90.  The two expressions differ in what the mean density divides, as written
// in the reference: ut += dt*(T1 + (1/dy)*(X-Y)/rbar);  vt += dt*(T1 + T2)/rbar.
__device__ __forceinline__ void c_temp_vel_diff(int i, int j, dp ut, dp vt, cdp u, cdp v, cdp r, double dt, double dx, double dy,
                                cdp m, int nx, int ny) {
  if (ut && i >= 2 && i <= nx && j >= 2 && j <= ny + 1) {
    const double t1 = (1. / dx) * 2. * (M(i + 1, j) * (1. / dx) * (U(i + 1, j) - U(i, j)) -
                                        M(i, j) * (1. / dx) * (U(i, j) - U(i - 1, j)));
    const double X = 0.25 * (M(i, j) + M(i + 1, j) + M(i + 1, j + 1) + M(i, j + 1)) *
                     ((1. / dy) * (U(i, j + 1) - U(i, j)) + (1. / dx) * (V(i + 1, j) - V(i, j)));
    const double Y = 0.25 * (M(i, j) + M(i + 1, j) + M(i + 1, j - 1) + M(i, j - 1)) *
                     ((1. / dy) * (U(i, j) - U(i, j - 1)) + (1. / dx) * (V(i + 1, j - 1) - V(i, j - 1)));
    const double t2 = (1. / dy) * (X - Y) / (0.5 * (R(i + 1, j) + R(i, j)));
    B2(ut, nx + 1, i, j) = B2(ut, nx + 1, i, j) + dt * (t1 + t2);
  }
  if (vt && i >= 2 && i <= nx + 1 && j >= 2 && j <= ny) {
    const double X = 0.25 * (M(i, j) + M(i + 1, j) + M(i + 1, j + 1) + M(i, j + 1)) *
                     ((1.0 / dy) * (U(i, j + 1) - U(i, j)) + (1.0 / dx) * (V(i + 1, j) - V(i, j)));
    const double Y = 0.25 * (M(i, j) + M(i, j + 1) + M(i - 1, j + 1) + M(i - 1, j)) *
                     ((1.0 / dy) * (U(i - 1, j + 1) - U(i - 1, j)) + (1.0 / dx) * (V(i, j) - V(i - 1, j)));
    const double t1 = (1.0 / dx) * (X - Y);
    const double t2 = (1.0 / dy) * 2.0 * (M(i, j + 1) * (1.0 / dy) * (V(i, j + 1) - V(i, j)) -
                                          M(i, j) * (1.0 / dy) * (V(i, j) - V(i, j - 1)));
    B2(vt, nx + 2, i, j) = B2(vt, nx + 2, i, j) + dt * (t1 + t2) / (0.5 * (R(i, j + 1) + R(i, j)));
  }
}
__global__ void k_temp_vel_diff(dp ut, dp vt, cdp u, cdp v, cdp r, double dt, double dx, double dy, cdp m, int nx, int ny) {
  IJ;
  c_temp_vel_diff(i, j, ut, vt, u, v, r, dt, dx, dy, m, nx, ny);
}

// code1.m:55-57: rt = r with lrg on the four borders
__device__ __forceinline__ void c_set_boundary(int i, int j, dp rt, cdp r, double lrg, int nx, int ny) {
  if (i > nx + 2 || j > ny + 2) return;
  const bool border = i == 1 || i == nx + 2 || j == 1 || j == ny + 2;
  RT(i, j) = border ? lrg : R(i, j);
}
__global__ void k_set_boundary(dp rt, cdp r, double lrg, int nx, int ny) {
  IJ;
  c_set_boundary(i, j, rt, r, lrg, nx, ny);
}

// code1.m:59-65 / code3_fun.jl:92-102
__device__ __forceinline__ void c_compute_sources(int i, int j, dp tmp1, dp tmp2, cdp rt, cdp ut, cdp vt, double dt, double dx,
                                  double dy, int nx, int ny) {
  if (i < 2 || i > nx + 1 || j < 2 || j > ny + 1) return;
  B2(tmp1, nx + 2, i, j) = (0.5 / dt) * ((B2(ut, nx + 1, i, j) - B2(ut, nx + 1, i - 1, j)) / dx +
                                         (B2(vt, nx + 2, i, j) - B2(vt, nx + 2, i, j - 1)) / dy);
  B2(tmp2, nx + 2, i, j) =
      1.0 / ((1.0 / dx) * (1.0 / (dx * (RT(i + 1, j) + RT(i, j))) + 1.0 / (dx * (RT(i - 1, j) + RT(i, j)))) +
             (1.0 / dy) * (1.0 / (dy * (RT(i, j + 1) + RT(i, j))) + 1.0 / (dy * (RT(i, j - 1) + RT(i, j)))));
}
__global__ void k_compute_sources(dp tmp1, dp tmp2, cdp rt, cdp ut, cdp vt, double dt, double dx, double dy, int nx, int ny) {
  IJ;
  c_compute_sources(i, j, tmp1, tmp2, rt, ut, vt, dt, dx, dy, nx, ny);
}

// code1.m:80-86
__device__ __forceinline__ void c_correct_vel(int i, int j, dp u, dp v, cdp ut, cdp vt, cdp p, cdp r, double dt, double dx,
                              double dy, int nx, int ny) {
  if (u && i >= 2 && i <= nx && j >= 2 && j <= ny + 1)
    U(i, j) = B2(ut, nx + 1, i, j) -
              dt * (2.0 / dx) * (B2(p, nx + 2, i + 1, j) - B2(p, nx + 2, i, j)) / (R(i + 1, j) + R(i, j));
  if (v && i >= 2 && i <= nx + 1 && j >= 2 && j <= ny)
    V(i, j) = B2(vt, nx + 2, i, j) -
              dt * (2.0 / dy) * (B2(p, nx + 2, i, j + 1) - B2(p, nx + 2, i, j)) / (R(i, j + 1) + R(i, j));
}
__global__ void k_correct_vel(dp u, dp v, cdp ut, cdp vt, cdp p, cdp r, double dt, double dx, double dy, int nx, int ny) {
  IJ;
  c_correct_vel(i, j, u, v, ut, vt, p, r, dt, dx, dy, nx, ny);
}

// code1.m:88-96
__device__ __forceinline__ void c_advect_density(int i, int j, dp r, cdp ro, cdp u, cdp v, double dt, double dx, double dy,
                                 double m0, int nx, int ny) {
  if (i < 2 || i > nx + 1 || j < 2 || j > ny + 1) return;
#define RO(i, j) B2(ro, nx + 2, i, j)
  R(i, j) = RO(i, j) -
            (0.5 * dt / dx) * (U(i, j) * (RO(i + 1, j) + RO(i, j)) - U(i - 1, j) * (RO(i - 1, j) + RO(i, j))) -
            (0.5 * dt / dy) * (V(i, j) * (RO(i, j + 1) + RO(i, j)) - V(i, j - 1) * (RO(i, j - 1) + RO(i, j))) +
            (m0 * dt / dx / dx) * (RO(i + 1, j) - 2.0 * RO(i, j) + RO(i - 1, j)) +
            (m0 * dt / dy / dy) * (RO(i, j + 1) - 2.0 * RO(i, j) + RO(i, j - 1));
#undef RO
}
__global__ void k_advect_density(dp r, cdp ro, cdp u, cdp v, double dt, double dx, double dy, double m0, int nx, int ny) {
  IJ;
  c_advect_density(i, j, r, ro, u, v, dt, dx, dy, m0, nx, ny);
}

// code1.m:100-101 / DNS.jl:214-220
__device__ __forceinline__ void c_velocity(int i, int j, dp uu, dp vv, cdp u, cdp v, int nx, int ny) {
  if (i > nx + 1 || j > ny + 1) return;
  B2(uu, nx + 1, i, j) = 0.5 * (U(i, j + 1) + U(i, j));
  B2(vv, nx + 1, i, j) = 0.5 * (V(i + 1, j) + V(i, j));
}
__global__ void k_velocity(dp uu, dp vv, cdp u, cdp v, int nx, int ny) {
  IJ;
  c_velocity(i, j, uu, vv, u, v, nx, ny);
}

// code3_fun.jl:440-441 + :242-249
__device__ __forceinline__ void c_update_visco(int i, int j, cdp r, dp m, double m1, double m2, double rho1, double rho2, int nx,
                               int ny) {
  if (i > nx + 2 || j > ny + 2) return;
  const bool inner = i >= 2 && i <= nx + 1 && j >= 2 && j <= ny + 1;
  M(i, j) = inner ? m1 + (m2 - m1) * (R(i, j) - rho1) / (rho2 - rho1) : 0.0 + m1;
}
__global__ void k_update_visco(cdp r, dp m, double m1, double m2, double rho1, double rho2, int nx, int ny) {
  IJ;
  c_update_visco(i, j, r, m, m1, m2, rho1, rho2, nx, ny);
}

// code3_fun.jl:445-450: a = 0.5*(a + b)
__global__ void k_average(dp a, cdp b, size_t n) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) a[k] = 0.5 * (a[k] + b[k]);
}

// ---- single-workgroup lexicographic SOR ------------------------------------------------
// MODE 0: pressure (code1.m:68-78 / code3_fun.jl:104-124); MODE 1: density (code3_fun.jl:224-240).
// `a` is the unknown ((nx+2)x(ny+2), updated in place).  LDS copy of `a` when USE_LDS; the
// coefficient arrays as well when COEF_LDS (everything fits: the reference sizes), so that a
// diagonal costs LDS latency instead of a round trip to L2.
struct SorArgs {
  double *a;
  const double *c1, *c2, *c3;  // pressure: tmp1, tmp2, rt;  density: fx, fy, unused
  double beta, dx, dy, maxError;
  int nx, ny, maxit;
  int *sweeps_out;             // sweeps performed
  double *change_out;          // last max|old-new|
};

template <int MODE, bool USE_LDS, bool COEF_LDS>
__global__ void __launch_bounds__(1024) k_sor(SorArgs A) {
  extern __shared__ double lds[];
  __shared__ unsigned long long red[16];
  __shared__ unsigned long long sweep_max;
  const int nx = A.nx, ny = A.ny, ld = nx + 2;
  const int tid = threadIdx.x, nth = blockDim.x;
  const size_t ncell = (size_t)ld * (ny + 2);
  double *a = USE_LDS ? lds : A.a;
  if (USE_LDS) {
    for (size_t k = tid; k < ncell; k += nth) lds[k] = A.a[k];
    __syncthreads();
  }
  const double beta = A.beta, dx = A.dx, dy = A.dy;
  const double *c1 = A.c1, *c2 = A.c2, *rt = A.c3;
  if (COEF_LDS) {
    double *l1 = lds + ncell, *l2 = lds + 2 * ncell, *l3 = lds + 3 * ncell;
    for (size_t k = tid; k < ncell; k += nth) {
      l1[k] = A.c1[k];
      l2[k] = A.c2[k];
      if (MODE == 0) l3[k] = A.c3[k];
    }
    __syncthreads();
    c1 = l1;
    c2 = l2;
    rt = l3;
  }
  int it = 0;
  unsigned long long last = 0;
  for (it = 1; it <= A.maxit; it++) {
    unsigned long long mymax = 0;
    // anti-diagonals d = i + j, i, j in 2..n+1
    for (int d = 4; d <= nx + ny + 2; d++) {
      const int ilo = max(2, d - (ny + 1)), ihi = min(nx + 1, d - 2);
      for (int i = ilo + tid; i <= ihi; i += nth) {
        const int j = d - i;
        const double old = B2(a, ld, i, j);
        double nw;
        if (MODE == 0) {
          nw = (1.0 - beta) * old +
               beta * B2(c2, ld, i, j) *
                   ((1.0 / dx) * (B2(a, ld, i + 1, j) / (dx * (RT(i + 1, j) + RT(i, j))) +
                                  B2(a, ld, i - 1, j) / (dx * (RT(i - 1, j) + RT(i, j)))) +
                    (1.0 / dy) * (B2(a, ld, i, j + 1) / (dy * (RT(i, j + 1) + RT(i, j))) +
                                  B2(a, ld, i, j - 1) / (dy * (RT(i, j - 1) + RT(i, j)))) -
                    B2(c1, ld, i, j));
        } else {
          nw = (1.0 - beta) * old +
               beta * (0.25 * (B2(a, ld, i + 1, j) + B2(a, ld, i - 1, j) + B2(a, ld, i, j + 1) +
                               B2(a, ld, i, j - 1) + dx * B2(c1, ld, i - 1, j) - dx * B2(c1, ld, i, j) +
                               dy * B2(c2, ld, i, j - 1) - dy * B2(c2, ld, i, j)));
        }
        B2(a, ld, i, j) = nw;
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(old - nw));
        mymax = b > mymax ? b : mymax;
      }
      __syncthreads();  // also makes the stores visible to the workgroup (global or LDS)
    }
    // maximum(abs.(old .- new)) over the whole array (ghost cells do not change): integer
    // max of the bit patterns keeps NaN like Julia's maximum
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(mymax, off, 64);
      mymax = o > mymax ? o : mymax;
    }
    if ((tid & 63) == 0) red[tid >> 6] = mymax;
    __syncthreads();
    if (tid == 0) {
      unsigned long long m = 0;
      for (int w = 0; w < (nth + 63) / 64; w++) m = red[w] > m ? red[w] : m;
      sweep_max = m;
    }
    __syncthreads();
    last = sweep_max;
    const double ch = __longlong_as_double((long long)last);
    if (ch < A.maxError) break;  // false for NaN: keeps sweeping, as the reference does
  }
  if (USE_LDS) {
    __syncthreads();
    for (size_t k = tid; k < ncell; k += nth) A.a[k] = lds[k];
  }
  if (tid == 0) {
    if (A.sweeps_out) *A.sweeps_out = it > A.maxit ? A.maxit : it;
    if (A.change_out) *A.change_out = __longlong_as_double((long long)last);
  }
}

// ---- several sweeps in flight ------------------------------------------------------------
// A lexicographic sweep is a chain of nx+ny-1 anti-diagonals, but sweep k+1 may work on diagonal e
// as soon as sweep k has finished diagonal e+1: its cells need the values of THIS sweep to the left
// and below (diagonal e-1) and the values of the PREVIOUS sweep of the cell itself, to the right
// and above (diagonals e, e+1).  So sweeps follow each other through the diagonals two steps apart,
// all on ONE array in LDS updated in place: in a step they write diagonals e, e-2, e-4, ... and read
// e-1..e+1, e-3..e-1, ..., which never meet, and every value is formed from exactly the operands
// the serial sweep uses -- the results are identical bit for bit.
//
// Threads: W groups of nx threads, a thread per column.  A thread walks up its column, one row a
// step: thread (w, ci) works on row r of sweep b W + w + 1 at step 2w + b P + ci + r, then rests
// until P = max(ny, 2W) steps are over and starts the column again for the group's next sweep.
// Within a sweep the columns are one step apart (the diagonal); with 2W >= ny a new sweep starts
// every other step and no thread ever rests for long.
//
// The stopping test max|old-new| < maxError belongs to whole sweeps, and by the time sweep k is
// known to have met it the sweeps behind it have already changed the array.  Every sweep therefore
// also stores its values into a snapshot of its own in global memory (number k mod R; stores only,
// nothing waits for them), and the snapshot of the final sweep is what is copied out; the sweeps
// behind it were speculation and are dropped.  A thread adds its column's max|old-new| to the
// sweep's word in LDS (number k mod R as well) when it reaches the top row; sweep k is complete at
// step 2w + b P + nx+ny-2.  R = W M with M P >= nx+ny steps between sweeps k and k+R, so a
// snapshot and a word are reused only after their sweep has been judged.
// All counters are kept incrementally (no integer division in the loop).
//
// What does not change from sweep to sweep is formed once, in LDS, by the expressions of the
// reference: beta*tmp2 for the pressure and, when a fifth array fits (PRE), the face denominators
// dx*(rt[i+1,j]+rt[i,j]) and dy*(rt[i,j+1]+rt[i,j]) (the sum is commutative, so the same array serves
// the left / lower face of the next cell); dx*fx and dy*fy for the density.
//
// FAST (two more arrays): the four divisions of a cell by face denominators are replaced by five
// fused operations each with the denominator's correctly rounded reciprocal r = RN(1/d), formed
// once per face:  q0 = RN(x r), then twice q <- RN(q - RN(q d - x) r).  The first correction makes
// q a faithful rounding of x/d whatever q0's error, the second is then Markstein's correction, which
// returns the correctly rounded quotient -- the bits of x/d, signed zeros included
// (tools/micro/fma_division_check.c compares 4 10^8 cases on the CPU).  It holds while no
// intermediate leaves the normal range: denominators in [2^-100, 2^100], dividends zero or in
// [2^-900, 2^900].  Every thread checks the denominators it forms and every value it loads or
// produces; if one falls outside, the whole solve is run again from its inputs with IEEE divisions.
__device__ __forceinline__ bool fast_dividend_ok(double x) {
  const unsigned e = ((unsigned)(__double_as_longlong(x) >> 32) >> 20) & 0x7ffu;  // biased exponent
  return e - 123u <= 1800u || x == 0.0;
}
__device__ __forceinline__ bool fast_divisor_ok(double d) {
  const int hi = (int)(__double_as_longlong(d) >> 32);
  return (unsigned)((hi >> 20) - 923) <= 200u;  // positive, 2^-100 <= d < 2^101
}
__device__ __forceinline__ double fast_div(double x, double d, double r) {
  const double q0 = x * r;
  const double t0 = __builtin_fma(q0, d, -x);
  const double q1 = __builtin_fma(-t0, r, q0);
  const double t1 = __builtin_fma(q1, d, -x);
  return __builtin_fma(-t1, r, q1);
}

// Debug builds only (tools/debug_sor_call.sh): NSK_SOR_NOINLINE makes the solver a real call from the
// merged-step kernels, NSK_SOR_DEBUG prints every judged sweep.
#ifdef NSK_SOR_NOINLINE
#define NSK_SOR_INLINE __noinline__
#else
#define NSK_SOR_INLINE __forceinline__
#endif
template <int MODE, bool PRE, bool FAST = false>
__device__ NSK_SOR_INLINE void sor_pipe(const SorArgs &A, double *snap, int W, int P, int M) {
  static_assert(!FAST || (MODE == 0 && PRE), "the fast division belongs to the pressure solve with face arrays");
  extern __shared__ double lds[];
  const int nx = A.nx, ny = A.ny, ld = nx + 2;
  const int tid = threadIdx.x, nth = blockDim.x;
  const int ncell = ld * (ny + 2), R = W * M;
  const double beta = A.beta, dx = A.dx, dy = A.dy;
  // pressure: a, tmp1, beta*tmp2, then rt or the two face arrays (and their reciprocals); density: a, dx*fx, dy*fy
  constexpr int NARR = MODE == 0 ? (FAST ? 7 : PRE ? 5 : 4) : 3;
  double *a = lds, *c1 = lds + ncell, *c2 = c1 + ncell, *rt = c2 + ncell, *fy = rt + ncell;
  double *fxd = rt;  // PRE: x-face denominators in rt's place, y-face denominators behind them
  double *rfx = fy + ncell, *rfy = rfx + ncell;  // FAST
  unsigned long long *sweep_max = reinterpret_cast<unsigned long long *>(lds + (size_t)NARR * ncell);  // [R]
  int *unsafe = reinterpret_cast<int *>(sweep_max + R);  // FAST: somebody met an operand the fast division does not cover
  int final_sweep = A.maxit;
  unsigned long long last = 0;
  // One run of the whole solve from its inputs; returns whether anybody met an operand the fast
  // division does not cover (the result is then discarded and the solve run again with IEEE divisions).
  auto solve = [&](auto fast_tag) -> bool {
    constexpr bool fast = decltype(fast_tag)::value;
    bool bad = false;
    for (int k = tid; k < ncell; k += nth) {
      const double ak = A.a[k];
      a[k] = ak;
      if (fast) bad |= !fast_dividend_ok(ak);
      if (MODE == 0) {
        c1[k] = A.c1[k];
        c2[k] = beta * A.c2[k];
        if (!PRE) {
          rt[k] = A.c3[k];
        } else {
          const int i0 = k % ld, j0 = k / ld;  // 0-based
          const double dxf = i0 + 1 < ld ? dx * (A.c3[k + 1] + A.c3[k]) : 0.0;
          const double dyf = j0 + 1 < ny + 2 ? dy * (A.c3[k + ld] + A.c3[k]) : 0.0;
          fxd[k] = dxf;
          fy[k] = dyf;
          if (fast) {
            rfx[k] = 1.0 / dxf;
            rfy[k] = 1.0 / dyf;
            // the faces of interior cells are the ones that divide
            if (i0 <= nx && j0 >= 1 && j0 <= ny) bad |= !fast_divisor_ok(dxf);
            if (j0 <= ny && i0 >= 1 && i0 <= nx) bad |= !fast_divisor_ok(dyf);
          }
        }
      } else {
        c1[k] = dx * A.c1[k];
        c2[k] = dy * A.c2[k];
      }
    }
    for (int k = tid; k < R; k += nth) sweep_max[k] = 0;
    if (fast && tid == 0) *unsafe = 0;
    __syncthreads();
    const int ND = nx + ny - 1;
    const int w = tid / nx, ci = tid - w * nx;  // this thread's group and column (0-based)
    const bool alive = w < W;
    int r = -(2 * w + ci);                      // this thread's row (0-based; negative: not started)
    int k = w + 1, ks = k == R ? 0 : k;         // its sweep, and the sweep modulo R
    int q = ld + ci + 1;                        // its cell; q +- 1, q +- ld the neighbours
    double *my_snap = snap + (size_t)ks * ncell;
    int c = -(ND - 1), batch = 0, bm = 0;       // step counter of the completion test, see below
    unsigned long long mymax = 0;
    last = 0;
    final_sweep = A.maxit;
    // every sweep up to maxit is complete after this many steps; the loop ends earlier by itself
    const long long step_limit = ((long long)A.maxit / W + 2) * P + 2 * W + ND;
    for (long long t = 0; t < step_limit; t++) {
      if (alive && r >= 0 && r < ny && k <= A.maxit) {
        // everything the cell needs is asked of LDS at once, ahead of any branch
        const double old = a[q], east = a[q + 1], west = a[q - 1], north = a[q + ld], south = a[q - ld];
        double nw;
        if (MODE == 0) {
          const double s1 = c1[q], s2 = c2[q];
          double ex, ey;  // the two brackets of code1.m:70-73
          if (PRE) {
            const double de = fxd[q], dw = fxd[q - 1], dn = fy[q], ds = fy[q - ld];
            if (fast) {
              const double re = rfx[q], rw = rfx[q - 1], rn = rfy[q], rs = rfy[q - ld];
              ex = fast_div(east, de, re) + fast_div(west, dw, rw);
              ey = fast_div(north, dn, rn) + fast_div(south, ds, rs);
            } else {
              ex = east / de + west / dw;
              ey = north / dn + south / ds;
            }
          } else {
            const double rc = rt[q], re = rt[q + 1], rw = rt[q - 1], rn = rt[q + ld], rs = rt[q - ld];
            ex = east / (dx * (re + rc)) + west / (dx * (rw + rc));
            ey = north / (dy * (rn + rc)) + south / (dy * (rs + rc));
          }
          nw = (1.0 - beta) * old + s2 * ((1.0 / dx) * ex + (1.0 / dy) * ey - s1);
        } else {
          nw = (1.0 - beta) * old + beta * (0.25 * (east + west + north + south + c1[q - 1] - c1[q] + c2[q - ld] - c2[q]));
        }
        a[q] = nw;
        my_snap[q] = nw;
        if (fast) bad |= !fast_dividend_ok(nw);
        const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(old - nw));
        mymax = (r == 0 || b > mymax) ? b : mymax;  // a new column starts a new maximum
        // the word's previous sweep was judged at least two steps ago
        if (r == 0 && ci == 0) sweep_max[ks] = 0;
        q += ld;
        // top of the column: its max|old-new| (integer max of the bit patterns keeps NaN)
        if (r == ny - 1) atomicMax(&sweep_max[ks], mymax);
      }
      if (++r == P) {
        r = 0;
        q = ld + ci + 1;
        k += W;
        ks += W;
        if (ks >= R) ks -= R;
        my_snap = snap + (size_t)ks * ncell;
      }
      __syncthreads();  // the step's values and, if a sweep ended, its maximum are visible to everybody
      // Group g completes a sweep at steps 2g + ND-1 + b P: c counts steps from ND-1 modulo P, so an
      // even c below 2W names the group (at most one per step), `batch` its b, bm = b modulo M.
      if (c >= 0 && !(c & 1) && (c >> 1) < W) {
        const int kc = batch * W + (c >> 1) + 1;
        if (kc <= A.maxit) {
          int idx = bm * W + (c >> 1) + 1;
          if (idx == R) idx = 0;
          last = sweep_max[idx];
          const double ch = __longlong_as_double((long long)last);
#ifdef NSK_SOR_DEBUG
          if (tid == 0) printf("SOR judge mode=%d fast=%d t=%lld kc=%d idx=%d last=%016llx lds=%u W=%d P=%d M=%d maxit=%d\n", MODE, (int)fast, t, kc, idx, last, (unsigned)(size_t)lds, W, P, M, A.maxit);
#endif
          if (ch < A.maxError || kc == A.maxit) {  // (false for NaN: keeps sweeping, as the reference does)
            final_sweep = kc;
            break;
          }
        }
      }
      if (++c == P) {
        c = 0;
        batch++;
        if (++bm == M) bm = 0;
      }
    }
    if (!fast) return false;
    if (bad) *unsafe = 1;
    __syncthreads();
    const bool again = *unsafe != 0;
    __syncthreads();  // everybody has read the flag before the next round clears it
    return again;
  };
  if (FAST) {
    if (solve(std::true_type{})) solve(std::false_type{});
  } else {
    solve(std::false_type{});
  }
  // interior cells of the final sweep's snapshot (ghost cells are never written by a sweep)
  const double *fin = snap + (size_t)(final_sweep % R) * ncell;
  for (int i = tid; i < nx * ny; i += nth) {
    const int j = i / nx, i0 = i - j * nx;
    B2(A.a, ld, i0 + 2, j + 2) = B2(fin, ld, i0 + 2, j + 2);
  }
  if (tid == 0) {
    if (A.sweeps_out) *A.sweeps_out = final_sweep;
    if (A.change_out) *A.change_out = __longlong_as_double((long long)last);
  }
}
template <int MODE, bool PRE, bool FAST>
__global__ void __launch_bounds__(1024) k_sor_pipe(SorArgs A, double *snap, int W, int P, int M) {
  sor_pipe<MODE, PRE, FAST>(A, snap, W, P, M);
}

// ---- front kernels ----------------------------------------------------------------------

// code3_fun.jl:372-378: tangents of segments l = 1..Nf+1 (1-based), then tx[Nf+2] = tx[2]
__device__ __forceinline__ void c_front_tangents(int l, dp tx, dp ty, cdp xf, cdp yf, int Nf) {
  if (l <= Nf + 1) {
    const double ex = xf[l] - xf[l - 1], ey = yf[l] - yf[l - 1];
    const double ds = sqrt(SQ(ex) + SQ(ey));
    tx[l - 1] = ex / ds;
    ty[l - 1] = ey / ds;
  }
  if (l == Nf + 2) {  // same quantities as l = 2, recomputed to avoid a second launch
    const double ex = xf[2] - xf[1], ey = yf[2] - yf[1];
    const double ds = sqrt(SQ(ex) + SQ(ey));
    tx[Nf + 1] = ex / ds;
    ty[Nf + 1] = ey / ds;
  }
}
__global__ void k_front_tangents(dp tx, dp ty, cdp xf, cdp yf, int Nf) {
  c_front_tangents(blockIdx.x * blockDim.x + threadIdx.x + 1, tx, ty, xf, yf, Nf);
}

// Spreading as a gather (code3_fun.jl:380-401 for KIND 0, :197-222 for KIND 1): grid point
// (i,j) of fx / fy sums, for l ascending, the weight with which front point l lands on it.
// The cell a front point lands in, its offsets in that cell and its strength do not depend on the
// grid point, so the workgroup works them out once per point (SPREAD_CHUNK points at a time) and
// keeps, in LDS and still in ascending l, only the points that land on its tile of the grid or on
// the ring around it; every grid point then only compares cell indices with those few, point after
// point in the order of the reference's scatter loop.
constexpr int SPREAD_CHUNK = BX * BY;
template <int KIND>
__global__ void __launch_bounds__(BX * BY) k_spread(dp fx, dp fy, cdp xf, cdp yf, cdp tx, cdp ty, double c0,
                                                    double c1, double dx, double dy, int nx, int ny, int Nf) {
  // cell of a point as one 64-bit key (ip low, jp high): a grid point is hit when the key is one of
  // its four (i or i-1, j or j-1)
  __shared__ unsigned long long s_key[2][SPREAD_CHUNK];
  __shared__ double s_ax[2][SPREAD_CHUNK], s_ay[2][SPREAD_CHUNK], s_nf[2][SPREAD_CHUNK];
  __shared__ int s_cnt[2][SPREAD_CHUNK / 64];
  IJ;
  const int tid = threadIdx.y * BX + threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // points whose cell (ip, jp) has ip in [i_lo-1, i_hi] and jp in [j_lo-1, j_hi] touch this tile
  const int i_lo = blockIdx.x * BX + 1, i_hi = i_lo + BX - 1, j_lo = blockIdx.y * BY + 1, j_hi = j_lo + BY - 1;
  auto key = [](int ip, int jp) { return (unsigned long long)(unsigned)ip | ((unsigned long long)(unsigned)jp << 32); };
  const unsigned long long k00 = key(i, j), k10 = key(i - 1, j), k01 = key(i, j - 1), k11 = key(i - 1, j - 1);
  double sx = 0.0, sy = 0.0;
  for (int l0 = 2; l0 <= Nf + 1; l0 += SPREAD_CHUNK) {
    const int l = l0 + tid;
    int ip[2] = {0, 0}, jp[2] = {0, 0};
    double ax[2], ay[2], nf[2];
    bool rel[2] = {false, false};
    if (l <= Nf + 1) {
      const double xl = xf[l - 1], yl = yf[l - 1];
      if (KIND == 0) {  // surface tension, c0 = sigma
        nf[0] = c0 * (tx[l - 1] - tx[l - 2]);
        nf[1] = c0 * (ty[l - 1] - ty[l - 2]);
      } else {          // density jump, c0 = rho1, c1 = rho2
        nf[0] = -0.5 * (yf[l] - yf[l - 2]) * (c1 - c0);
        nf[1] = 0.5 * (xf[l] - xf[l - 2]) * (c1 - c0);
      }
      ip[0] = (int)floor(xl / dx) + 1;
      jp[0] = (int)floor((yl + 0.5 * dy) / dy) + 1;
      ax[0] = xl / dx - ip[0] + 1;
      ay[0] = (yl + 0.5 * dy) / dy - jp[0] + 1;
      ip[1] = (int)floor((xl + 0.5 * dx) / dx) + 1;
      jp[1] = (int)floor(yl / dy) + 1;
      ax[1] = (xl + 0.5 * dx) / dx - ip[1] + 1;
      ay[1] = yl / dy - jp[1] + 1;
      for (int f = 0; f < 2; f++) rel[f] = ip[f] >= i_lo - 1 && ip[f] <= i_hi && jp[f] >= j_lo - 1 && jp[f] <= j_hi;
    }
    // ordered compaction: place among the wavefront's points, then the wavefronts one after the other
    int pos[2];
    for (int f = 0; f < 2; f++) {
      const unsigned long long mask = __ballot(rel[f]);
      pos[f] = __popcll(mask & ((1ull << lane) - 1));
      if (lane == 0) s_cnt[f][wave] = __popcll(mask);
    }
    __syncthreads();
    int n[2];
    for (int f = 0; f < 2; f++) {
      int base = 0, total = 0;
      for (int w = 0; w < SPREAD_CHUNK / 64; w++) {
        const int cw = s_cnt[f][w];
        base += w < wave ? cw : 0;
        total += cw;
      }
      n[f] = total;
      if (rel[f]) {
        const int at = base + pos[f];
        s_key[f][at] = key(ip[f], jp[f]);
        s_ax[f][at] = ax[f];
        s_ay[f][at] = ay[f];
        s_nf[f][at] = nf[f];
      }
    }
    __syncthreads();
    for (int q = 0; q < n[0]; q++) {
      const unsigned long long kx = s_key[0][q];
      if (kx == k00 || kx == k10 || kx == k01 || kx == k11) {
        const double axq = s_ax[0][q], ayq = s_ay[0][q];
        const double wx = (unsigned)kx == (unsigned)i ? 1.0 - axq : axq, wy = (unsigned)(kx >> 32) == (unsigned)j ? 1.0 - ayq : ayq;
        sx = sx + wx * wy * s_nf[0][q] / dx / dy;
      }
    }
    for (int q = 0; q < n[1]; q++) {
      const unsigned long long ky = s_key[1][q];
      if (ky == k00 || ky == k10 || ky == k01 || ky == k11) {
        const double axq = s_ax[1][q], ayq = s_ay[1][q];
        const double wx = (unsigned)ky == (unsigned)i ? 1.0 - axq : axq, wy = (unsigned)(ky >> 32) == (unsigned)j ? 1.0 - ayq : ayq;
        sy = sy + wx * wy * s_nf[1][q] / dx / dy;
      }
    }
    __syncthreads();
  }
  if (i > nx + 2 || j > ny + 2) return;
  B2(fx, nx + 2, i, j) = sx;
  B2(fy, nx + 2, i, j) = sy;
}

// The same spreading inside a launch of ONE workgroup (the merged time step below).  A point adds
// to the four grid points around it (1-ax)(1-ay) nf/dx/dy, ax (1-ay) nf/dx/dy, ... -- expressions
// that do not depend on the grid point, so the point's thread evaluates all four (per field) into
// LDS once (`lds`: 10 arrays of `cap` values).  Then every wavefront takes tiles of 16 x 4 grid
// points; a lane tests one point per round against the tile and its ring, and the wavefront visits
// the points of the ballot in ascending order: a lane whose grid point is one of the four adds that
// corner's value.
template <int KIND>
__device__ __forceinline__ void spread_tiles(double *lds, int cap, dp fx, dp fy, cdp xf, cdp yf, cdp tx, cdp ty, double c0,
                                             double c1, double dx, double dy, int nx, int ny, int Nf) {
  unsigned long long *s_key = reinterpret_cast<unsigned long long *>(lds);  // [2][cap]
  double *s_val = lds + 2 * (size_t)cap;                                    // [2][cap][4]: corners 00, 10, 01, 11
  auto key = [](int ip, int jp) { return (unsigned long long)(unsigned)ip | ((unsigned long long)(unsigned)jp << 32); };
  for (int idx = threadIdx.x; idx < Nf; idx += blockDim.x) {
    const int l = idx + 2;
    const double xl = xf[l - 1], yl = yf[l - 1];
    double nf[2];
    if (KIND == 0) {  // surface tension, c0 = sigma
      nf[0] = c0 * (tx[l - 1] - tx[l - 2]);
      nf[1] = c0 * (ty[l - 1] - ty[l - 2]);
    } else {          // density jump, c0 = rho1, c1 = rho2
      nf[0] = -0.5 * (yf[l] - yf[l - 2]) * (c1 - c0);
      nf[1] = 0.5 * (xf[l] - xf[l - 2]) * (c1 - c0);
    }
#pragma unroll
    for (int f = 0; f < 2; f++) {
      const double xs = f == 0 ? xl : xl + 0.5 * dx, ys = f == 0 ? yl + 0.5 * dy : yl;
      const int ip = (int)floor(xs / dx) + 1, jp = (int)floor(ys / dy) + 1;
      const double ax = xs / dx - ip + 1, ay = ys / dy - jp + 1;
      s_key[f * cap + idx] = key(ip, jp);
      double *v = s_val + ((size_t)f * cap + idx) * 4;
      v[0] = (1.0 - ax) * (1.0 - ay) * nf[f] / dx / dy;
      v[1] = ax * (1.0 - ay) * nf[f] / dx / dy;
      v[2] = (1.0 - ax) * ay * nf[f] / dx / dy;
      v[3] = ax * ay * nf[f] / dx / dy;
    }
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
  const int tiles_x = (nx + 2 + 15) / 16, tiles_y = (ny + 2 + 3) / 4;
  for (int tile = wave; tile < tiles_x * tiles_y; tile += nwaves) {
    const int tj = tile / tiles_x, ti = tile - tj * tiles_x;
    const int i_lo = ti * 16 + 1, j_lo = tj * 4 + 1, i = i_lo + (lane & 15), j = j_lo + (lane >> 4);
    const unsigned long long k00 = key(i, j), k10 = key(i - 1, j), k01 = key(i, j - 1), k11 = key(i - 1, j - 1);
    double acc[2] = {0.0, 0.0};
#pragma unroll
    for (int f = 0; f < 2; f++) {
      const int o = f * cap;
      for (int base = 0; base < Nf; base += 64) {
        const int idx = base + lane;
        bool rel = false;
        if (idx < Nf) {
          const unsigned long long kq = s_key[o + idx];
          const int ip = (int)(unsigned)kq, jp = (int)(unsigned)(kq >> 32);
          rel = ip >= i_lo - 1 && ip <= i_lo + 15 && jp >= j_lo - 1 && jp <= j_lo + 3;
        }
        unsigned long long mask = __ballot(rel);
        while (mask) {
          const int qd = o + base + __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const unsigned long long kq = s_key[qd];
          const double *v = s_val + (size_t)qd * 4;
          const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
          if (kq == k00 || kq == k10 || kq == k01 || kq == k11)
            acc[f] = acc[f] + (kq == k00 ? v0 : kq == k10 ? v1 : kq == k01 ? v2 : v3);
        }
      }
    }
    if (i <= nx + 2 && j <= ny + 2) {
      B2(fx, nx + 2, i, j) = acc[0];
      B2(fy, nx + 2, i, j) = acc[1];
    }
  }
}

// code3_fun.jl:144-195: interpolate the velocity to the front and move it; one thread per point,
// then thread 0 of the last block closes the periodic ghost points.
__device__ __forceinline__ void c_advect_move_front(int l, dp uf, dp vf, dp xf, dp yf, cdp u, cdp v, double dx, double dy,
                                    double dt, double Lx, double Ly, int nx, int ny, int Nf,
                                    int *out_count, bool nan_is_outside = false) {
  if (l > Nf + 1) return;
  const double xl = xf[l - 1], yl = yf[l - 1];
  // A front point outside the box is an error in the reference (BoundsError); the index clamps
  // below only keep the loads inside the arrays in that case.
  int ip = (int)floor(xl / dx) + 1, jp = (int)floor((yl + 0.5 * dy) / dy) + 1;
  double ax = xl / dx - ip + 1, ay = (yl + 0.5 * dy) / dy - jp + 1;
  ip = min(max(ip, 1), nx);
  jp = min(max(jp, 1), ny + 1);
  double v1 = (1.0 - ax) * (1.0 - ay) * U(ip, jp), v2 = ax * (1.0 - ay) * U(ip + 1, jp);
  double v3 = (1.0 - ax) * ay * U(ip, jp + 1), v4 = ax * ay * U(ip + 1, jp + 1);
  const double ufl = v1 + v2 + v3 + v4;
  ip = (int)floor((xl + 0.5 * dx) / dx) + 1;
  jp = (int)floor(yl / dy) + 1;
  ax = (xl + 0.5 * dx) / dx - ip + 1;
  ay = yl / dy - jp + 1;
  ip = min(max(ip, 1), nx + 1);
  jp = min(max(jp, 1), ny);
  v1 = (1.0 - ax) * (1.0 - ay) * V(ip, jp);
  v2 = ax * (1.0 - ay) * V(ip + 1, jp);
  v3 = (1.0 - ax) * ay * V(ip, jp + 1);
  v4 = ax * ay * V(ip + 1, jp + 1);
  const double vfl = v1 + v2 + v3 + v4;
  uf[l - 1] = ufl;
  vf[l - 1] = vfl;
  const double xn = xl + dt * ufl, yn = yl + dt * vfl;
  xf[l - 1] = xn;
  yf[l - 1] = yn;
  // the reference's message (code3_fun.jl:186-188) tests xn < 0 || xn > Lx || ...; a NaN position fails
  // that test but is no position inside the box either: the time steps count it as outside
  const bool outside = nan_is_outside ? !(xn >= 0.0 && xn <= Lx && yn >= 0.0 && yn <= Ly)
                                      : (xn < 0.0 || xn > Lx || yn < 0.0 || yn > Ly);
  if (out_count && outside) atomicAdd(out_count, 1);
}
__global__ void k_advect_move_front(dp uf, dp vf, dp xf, dp yf, cdp u, cdp v, double dx, double dy, double dt, double Lx, double Ly, int nx, int ny, int Nf, int *out_count, bool nan_is_outside) {
  c_advect_move_front(blockIdx.x * blockDim.x + threadIdx.x + 2, uf, vf, xf, yf, u, v, dx, dy, dt, Lx, Ly, nx, ny, Nf, out_count, nan_is_outside);
}

__global__ void k_close_front(dp xf, dp yf, int Nf) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    xf[0] = xf[Nf];
    yf[0] = yf[Nf];
    xf[Nf + 1] = xf[1];
    yf[Nf + 1] = yf[1];
  }
}

// code3_fun.jl:455-494: regridding compares every old point with the point written last, so it is
// one chain, walked by one lane.  Unless old point l-1 was dropped, the point written last IS old
// point l-1 (kept, or the far end of a split segment), so the distance the chain needs at l is the
// one between old neighbours: when `dist` is given it holds those (worked out by all lanes
// beforehand) and the chain computes only the distances after a dropped point.
// xo/yo: the old front; returns the new size, or -1 when `cap` points do not hold it.
template <bool HAVE_DIST>
__device__ __forceinline__ int regrid_chain(dp xf, dp yf, const double *xo, const double *yo, const double *dist,
                                            double dx, double dy, int Nf, int cap) {
  int j = 1;
  bool last_is_prev = HAVE_DIST;
  double lx = xo[0], ly = yo[0];  // xf[j-1], yf[j-1]: the point written last
  double nxt_x = xo[1], nxt_y = yo[1], nxt_d = HAVE_DIST ? dist[2] : 0.0;
  for (int l = 2; l <= Nf + 1; l++) {
    const double ox = nxt_x, oy = nxt_y, od = nxt_d;  // xo[l-1], yo[l-1], dist[l]
    nxt_x = xo[l];                                    // (l <= Nf+1: in range) read ahead of the chain
    nxt_y = yo[l];
    if (HAVE_DIST && l < Nf + 1) nxt_d = dist[l + 1];
    const double ds = last_is_prev ? od : sqrt(SQ((ox - lx) / dx) + SQ((oy - ly) / dy));
    if (ds > 0.5) {
      j += 1;
      if (j + 2 > cap) return -1;
      xf[j - 1] = 0.5 * (ox + lx);
      yf[j - 1] = 0.5 * (oy + ly);
      j += 1;
      xf[j - 1] = lx = ox;
      yf[j - 1] = ly = oy;
      last_is_prev = HAVE_DIST;
    } else if (ds < 0.25) {
      last_is_prev = false;
    } else {
      j += 1;
      if (j + 1 > cap) return -1;
      xf[j - 1] = lx = ox;
      yf[j - 1] = ly = oy;
      last_is_prev = HAVE_DIST;
    }
  }
  if (j + 1 > cap) return -1;
  const int Nn = j - 1;
  xf[0] = lx;  // xf[Nn]
  yf[0] = ly;
  // xf[1] is the first point written (j = 2), or untouched when nothing was written
  xf[Nn + 1] = xf[1];
  yf[Nn + 1] = yf[1];
  return Nn;
}

// One wavefront: the old front goes to xo/yo (scratch copies) and, when it fits, to LDS.
// *Nf_io is updated; -1 signals cap exceeded.
__global__ void __launch_bounds__(64) k_regrid_front(dp xf, dp yf, dp xo, dp yo, double dx, double dy, int *Nf_io, int cap) {
  constexpr int LDS_POINTS = 2048;  // fronts up to this size are walked in LDS
  __shared__ double s_x[LDS_POINTS], s_y[LDS_POINTS], s_d[LDS_POINTS];
  if (blockIdx.x != 0) return;
  const int Nf = *Nf_io;
  const bool in_lds = Nf + 2 <= LDS_POINTS;
  for (int k = threadIdx.x; k < Nf + 2; k += 64) {
    const double x = xf[k], y = yf[k];
    xo[k] = x;
    yo[k] = y;
    if (in_lds) {
      s_x[k] = x;
      s_y[k] = y;
    }
  }
  __syncthreads();
  if (in_lds) {
    for (int l = 2 + threadIdx.x; l <= Nf + 1; l += 64)
      s_d[l] = sqrt(SQ((s_x[l - 1] - s_x[l - 2]) / dx) + SQ((s_y[l - 1] - s_y[l - 2]) / dy));
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  *Nf_io = in_lds ? regrid_chain<true>(xf, yf, s_x, s_y, s_d, dx, dy, Nf, cap)
                  : regrid_chain<false>(xf, yf, xo, yo, nullptr, dx, dy, Nf, cap);
}

// Launch plan of k_sor_pipe / sor_pipe for a grid, or ok = false when the grid does not fit.
struct PipePlan {
  bool ok, pre, fast;
  int W, P, M, nth;
  size_t pbytes;
};
PipePlan pipe_plan(const nsk_ctx *ctx, int mode, int nx, int ny, int maxit) {
  PipePlan pl{};
  const size_t bytes = sizeof(double) * (size_t)(nx + 2) * (ny + 2);
  // W groups of nx threads, each sweep P steps after the group's previous one (see sor_pipe)
  int W = nx <= 512 ? 1024 / nx : 0;
  if (W > (ny + 1) / 2) W = (ny + 1) / 2;  // a new sweep every other step already
  if (W > maxit) W = maxit;
  pl.W = W;
  pl.P = ny > 2 * W ? ny : 2 * W;
  pl.M = W >= 1 ? (nx + ny + pl.P - 1) / pl.P : 1;  // M P >= nx+ny-1 + 1 steps before a snapshot is reused
  const size_t R = (size_t)W * pl.M;
  // pressure: four arrays, five when the face denominators fit as well; density: three
  // ... seven with the reciprocals of the fast division
  const size_t words = R * 8 + 8;  // the sweeps' maxima and the fast division's flag
  pl.pre = mode == 0 && bytes * 5 + words <= 156 * 1024;
  pl.fast = mode == 0 && ctx->sor_fast_division && bytes * 7 + words <= 156 * 1024;
  pl.pbytes = bytes * (mode == 0 ? (pl.fast ? 7 : pl.pre ? 5 : 4) : 3) + words;
  pl.nth = (W * nx + 63) / 64 * 64;  // threads past W*nx belong to no group
  pl.ok = ctx->sor_pipeline && W >= 2 && pl.pbytes <= 156 * 1024;
  return pl;
}
// the snapshots of a plan's sweeps in flight
int pipe_reserve(nsk_ctx *ctx, const PipePlan &pl, int nx, int ny) {
  const size_t snap_elems = (size_t)pl.W * pl.M * (nx + 2) * (ny + 2);
  if (ctx->sor_snap_elems < snap_elems) {
    if (ctx->sor_snap) NSK_HIP(hipFree(ctx->sor_snap));
    ctx->sor_snap = nullptr;
    ctx->sor_snap_elems = 0;
    NSK_HIP(hipMalloc(&ctx->sor_snap, snap_elems * sizeof(double)));
    ctx->sor_snap_elems = snap_elems;
  }
  return 0;
}

// dev_sweeps: device word that receives the sweep count instead of the default slot (the fused
// steps read all their counters back once, at the end).
int launch_sor(nsk_ctx *ctx, int mode, double *a, const double *c1, const double *c2,
               const double *c3, double beta, double dx, double dy, int nx, int ny, int maxit,
               double maxError, int *sweeps, double *change, int *dev_sweeps = nullptr) {
  // device-side result slots live in the context's reduction scratch
  int *d_sweeps = dev_sweeps ? dev_sweeps : reinterpret_cast<int *>(ctx->red_dev + 2);
  double *d_change = reinterpret_cast<double *>(ctx->red_dev + 3);
  SorArgs A{a, c1, c2, c3, beta, dx, dy, maxError, nx, ny, maxit, d_sweeps, d_change};
  const size_t bytes = sizeof(double) * (size_t)(nx + 2) * (ny + 2);
  {
    // Several sweeps in flight when the unknown and the coefficient arrays fit the 160 KB of LDS
    const PipePlan pl = pipe_plan(ctx, mode, nx, ny, maxit);
    if (pl.ok) {
      if (int rc = pipe_reserve(ctx, pl, nx, ny)) return rc;
      const bool pre = pl.pre;
#define NSK_PIPE(M_, P_, F_)                                                                                      \
  do {                                                                                                            \
    NSK_HIP(nsk_allow_lds(ctx, (const void *)k_sor_pipe<M_, P_, F_>, pl.pbytes));                                 \
    k_sor_pipe<M_, P_, F_><<<1, pl.nth, pl.pbytes, ctx->stream>>>(A, ctx->sor_snap, pl.W, pl.P, pl.M);            \
  } while (0)
      if (mode != 0) NSK_PIPE(1, false, false);
      else if (pl.fast) NSK_PIPE(0, true, true);
      else if (pre) NSK_PIPE(0, true, false);
      else NSK_PIPE(0, false, false);
#undef NSK_PIPE
      NSK_LAUNCH_CHECK();
      if (sweeps || change) {
        NSK_HIP(hipMemcpyAsync(ctx->red_host + 2, ctx->red_dev + 2, 16, hipMemcpyDeviceToHost, ctx->stream));
        NSK_HIP(hipStreamSynchronize(ctx->stream));
        if (sweeps) *sweeps = *reinterpret_cast<int *>(ctx->red_host + 2);
        if (change) *change = *reinterpret_cast<double *>(ctx->red_host + 3);
      }
      return 0;
    }
  }
  if (ctx->sor_grid) {
    // larger grids: tiles on the whole chip, several sweeps in flight (bubble_sor_grid.hip)
    if (int rc = nsk_sor_grid(ctx, mode, a, c1, c2, c3, beta, dx, dy, nx, ny, maxit, maxError, d_sweeps, d_change)) return rc;
    if (sweeps || change) {
      NSK_HIP(hipMemcpyAsync(ctx->red_host + 2, ctx->red_dev + 2, 16, hipMemcpyDeviceToHost, ctx->stream));
      NSK_HIP(hipStreamSynchronize(ctx->stream));
      const int n = *reinterpret_cast<int *>(ctx->red_host + 2);
      NSK_REQUIRE_RC(n >= 0 || dev_sweeps, NSK_ERR_HIP, "SOR: a wait inside the solve ran out (no progress on the device)");
      if (sweeps) *sweeps = n;
      if (change) *change = *reinterpret_cast<double *>(ctx->red_host + 3);
    }
    return 0;
  }
  const bool use_lds = bytes <= 128 * 1024;
  const size_t all_bytes = bytes * (mode == 0 ? 4 : 3);
  const bool coef_lds = all_bytes <= 144 * 1024;
  int nth = nx < ny ? nx : ny;
  nth = ((nth + 63) / 64) * 64;
  if (nth > 1024) nth = 1024;
  if (nth < 64) nth = 64;
#define NSK_SOR(M, U, CL, BYTES)                                                                         \
  do {                                                                                                   \
    if ((BYTES) > 0)                                                                                     \
      NSK_HIP(hipFuncSetAttribute((const void *)k_sor<M, U, CL>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)(BYTES)));                                                        \
    k_sor<M, U, CL><<<1, nth, (BYTES), ctx->stream>>>(A);                                                \
  } while (0)
  if (coef_lds) {
    if (mode == 0) NSK_SOR(0, true, true, all_bytes);
    else NSK_SOR(1, true, true, all_bytes);
  } else if (use_lds) {
    if (mode == 0) NSK_SOR(0, true, false, bytes);
    else NSK_SOR(1, true, false, bytes);
  } else {
    if (mode == 0) NSK_SOR(0, false, false, 0);
    else NSK_SOR(1, false, false, 0);
  }
#undef NSK_SOR
  NSK_LAUNCH_CHECK();
  if (sweeps || change) {
    NSK_HIP(hipMemcpyAsync(ctx->red_host + 2, ctx->red_dev + 2, 16, hipMemcpyDeviceToHost, ctx->stream));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    if (sweeps) *sweeps = *reinterpret_cast<int *>(ctx->red_host + 2);
    if (change) *change = *reinterpret_cast<double *>(ctx->red_host + 3);
  }
  return 0;
}

#define NSK_BGRID_OK(nx, ny) \
  NSK_REQUIRE((nx) >= 2 && (ny) >= 2, "grid too small: nx=%d ny=%d (need >= 2)", (nx), (ny))

}  // namespace

extern "C" int nsk_bub_set_density(nsk_ctx *ctx, double *r, double *m, double xc, double yc,
                                   const double *x, const double *y, double rad, double rho2,
                                   double m2, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(r, x, y);
  NSK_BGRID_OK(nx, ny);
  k_set_density<<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(r, m, xc, yc, x, y, rad, rho2, m2, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_set_tangential(nsk_ctx *ctx, double *u, double *v, double usouth,
                                      double unorth, double vwest, double veast, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(u, v);
  NSK_BGRID_OK(nx, ny);
  const int n = (nx > ny ? nx : ny) + 1;
  k_set_tangential<<<nsk_cdiv(n, 256), 256, 0, ctx->stream>>>(u, v, usouth, unorth, vwest, veast, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_temp_vel(nsk_ctx *ctx, double *ut, double *vt, const double *u,
                                const double *v, const double *r, double dt, double dx, double dy,
                                double gx, double gy, double m0, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(u, v, r);
  NSK_BGRID_OK(nx, ny);
  k_temp_vel<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(ut, vt, u, v, r, dt, dx, dy, gx, gy, m0, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_temp_vel_adv(nsk_ctx *ctx, double *ut, double *vt, const double *u,
                                    const double *v, const double *r, double dt, double dx,
                                    double dy, const double *fx, const double *fy, double gx,
                                    double gy, double rro, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(ut, vt, u, v, r, fx, fy);
  NSK_BGRID_OK(nx, ny);
  k_temp_vel_adv<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(ut, vt, u, v, r, dt, dx, dy, fx, fy, gx, gy, rro, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_temp_vel_diff(nsk_ctx *ctx, double *ut, double *vt, const double *u,
                                     const double *v, const double *r, double dt, double dx,
                                     double dy, const double *m, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(ut, vt, u, v, r, m);
  NSK_BGRID_OK(nx, ny);
  k_temp_vel_diff<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(ut, vt, u, v, r, dt, dx, dy, m, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_set_boundary(nsk_ctx *ctx, double *rt, const double *r, double lrg, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(rt, r);
  NSK_BGRID_OK(nx, ny);
  k_set_boundary<<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(rt, r, lrg, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_compute_sources(nsk_ctx *ctx, double *tmp1, double *tmp2, const double *rt,
                                       const double *ut, const double *vt, double dt, double dx,
                                       double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(tmp1, tmp2, rt, ut, vt);
  NSK_BGRID_OK(nx, ny);
  k_compute_sources<<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(tmp1, tmp2, rt, ut, vt, dt, dx, dy, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_solve_pressure(nsk_ctx *ctx, double *p, const double *tmp1,
                                      const double *tmp2, const double *rt, double beta, double dx,
                                      double dy, int nx, int ny, int maxit, double maxError,
                                      int *sweeps_out, double *change_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(p, tmp1, tmp2, rt);
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(maxit >= 1, "solve_pressure: maxit=%d", maxit);
  return launch_sor(ctx, 0, p, tmp1, tmp2, rt, beta, dx, dy, nx, ny, maxit, maxError, sweeps_out, change_out);
}

extern "C" int nsk_bub_construct_den(nsk_ctx *ctx, double *r, const double *fx, const double *fy,
                                     double beta, int maxit, double maxError, double dx, double dy,
                                     int nx, int ny, int *sweeps_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(r, fx, fy);
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(maxit >= 1, "construct_den: maxit=%d", maxit);
  return launch_sor(ctx, 1, r, fx, fy, nullptr, beta, dx, dy, nx, ny, maxit, maxError, sweeps_out, nullptr);
}

extern "C" int nsk_bub_correct_vel(nsk_ctx *ctx, double *u, double *v, const double *ut,
                                   const double *vt, const double *p, const double *r, double dt,
                                   double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(ut, vt, p, r);
  NSK_BGRID_OK(nx, ny);
  k_correct_vel<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(u, v, ut, vt, p, r, dt, dx, dy, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_advect_density(nsk_ctx *ctx, double *r, const double *ro, const double *u,
                                      const double *v, double dt, double dx, double dy, double m0,
                                      int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(r, ro, u, v);
  NSK_BGRID_OK(nx, ny);
  k_advect_density<<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(r, ro, u, v, dt, dx, dy, m0, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_velocity(nsk_ctx *ctx, double *uu, double *vv, const double *u,
                                const double *v, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(uu, vv, u, v);
  NSK_BGRID_OK(nx, ny);
  k_velocity<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(uu, vv, u, v, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_update_visco(nsk_ctx *ctx, const double *r, double *m, double m1, double m2,
                                    double rho1, double rho2, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(r, m);
  NSK_BGRID_OK(nx, ny);
  k_update_visco<<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(r, m, m1, m2, rho1, rho2, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_average(nsk_ctx *ctx, double *a, const double *b, size_t count) {
  NSK_CTX(ctx);
  NSK_NONNULL(a, b);
  if (count) {
    k_average<<<(unsigned)((count + 255) / 256), 256, 0, ctx->stream>>>(a, b, count);
    NSK_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int nsk_bub_surface_tension(nsk_ctx *ctx, double *fx, double *fy, double *tx, double *ty,
                                       const double *xf, const double *yf, double sigma, double dx,
                                       double dy, int nx, int ny, int Nf) {
  NSK_CTX(ctx);
  NSK_NONNULL(fx, fy, tx, ty, xf, yf);
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(Nf >= 1, "surface_tension: Nf=%d", Nf);
  k_front_tangents<<<nsk_cdiv(Nf + 2, 256), 256, 0, ctx->stream>>>(tx, ty, xf, yf, Nf);
  NSK_LAUNCH_CHECK();
  k_spread<0><<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(fx, fy, xf, yf, tx, ty, sigma, 0.0, dx, dy, nx, ny, Nf);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_distribute(nsk_ctx *ctx, double *fx, double *fy, const double *xf,
                                  const double *yf, double rho1, double rho2, double dx, double dy,
                                  int nx, int ny, int Nf) {
  NSK_CTX(ctx);
  NSK_NONNULL(fx, fy, xf, yf);
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(Nf >= 1, "distribute: Nf=%d", Nf);
  k_spread<1><<<grid2(nx + 2, ny + 2), block2(), 0, ctx->stream>>>(fx, fy, xf, yf, nullptr, nullptr, rho1, rho2, dx, dy, nx, ny, Nf);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_bub_advect_move_front(nsk_ctx *ctx, double *uf, double *vf, double *xf, double *yf,
                                         const double *u, const double *v, double dx, double dy,
                                         double dt, double Lx, double Ly, int nx, int ny, int Nf,
                                         int *out_of_bounds) {
  NSK_CTX(ctx);
  NSK_NONNULL(uf, vf, xf, yf, u, v);
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(Nf >= 1, "advect_move_front: Nf=%d", Nf);
  int *d_cnt = reinterpret_cast<int *>(ctx->red_dev + 4);
  if (out_of_bounds) NSK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), ctx->stream));
  k_advect_move_front<<<nsk_cdiv(Nf, 256), 256, 0, ctx->stream>>>(uf, vf, xf, yf, u, v, dx, dy, dt, Lx, Ly, nx, ny, Nf,
                                                                 out_of_bounds ? d_cnt : nullptr, false);
  NSK_LAUNCH_CHECK();
  k_close_front<<<1, 64, 0, ctx->stream>>>(xf, yf, Nf);
  NSK_LAUNCH_CHECK();
  if (out_of_bounds) {
    NSK_HIP(hipMemcpyAsync(ctx->red_host + 4, d_cnt, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    *out_of_bounds = *reinterpret_cast<int *>(ctx->red_host + 4);
  }
  return 0;
}

extern "C" int nsk_bub_regrid_front(nsk_ctx *ctx, double *xf, double *yf, double *xscratch,
                                    double *yscratch, double dx, double dy, int Nf, int cap,
                                    int *Nf_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(xf, yf, xscratch, yscratch);
  NSK_REQUIRE(Nf_out != nullptr && Nf >= 1 && cap >= Nf + 2, "regrid_front: bad arguments (Nf=%d cap=%d)", Nf, cap);
  int *d_nf = reinterpret_cast<int *>(ctx->red_dev + 5);
  *reinterpret_cast<int *>(ctx->red_host + 5) = Nf;
  NSK_HIP(hipMemcpyAsync(d_nf, ctx->red_host + 5, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  k_regrid_front<<<1, 64, 0, ctx->stream>>>(xf, yf, xscratch, yscratch, dx, dy, d_nf, cap);
  NSK_LAUNCH_CHECK();
  NSK_HIP(hipMemcpyAsync(ctx->red_host + 5, d_nf, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  *Nf_out = *reinterpret_cast<int *>(ctx->red_host + 5);
  NSK_REQUIRE_RC(*Nf_out >= 0, NSK_ERR_FRONT_CAPACITY, "regrid_front: the front outgrew its capacity of %d points", cap);
  return 0;
}

// ---- whole time steps --------------------------------------------------------------------------
// The reference's drivers call one kernel after the other and read the sweep count of every
// pressure solve back on the host (DNS.jl:83-93 even synchronises per sweep).  These two entry
// points queue a whole time step and wait once, at its end, for the sweep counts (and the new front
// size): same kernels, same order, same values as the kernel-by-kernel sequence.
namespace {
// Up to six (a, b, n) pairs in one launch: a = b (AVERAGE false) or a = 0.5*(a + b) (code3_fun.jl:445-450).
struct Pairs {
  double *a[6];
  const double *b[6];
  size_t n[6];
};
template <bool AVERAGE>
__global__ void k_pairs(Pairs P) {
  const int q = blockIdx.y;
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < P.n[q]) P.a[q][k] = AVERAGE ? 0.5 * (P.a[q][k] + P.b[q][k]) : P.b[q][k];
}
template <bool AVERAGE>
int launch_pairs(nsk_ctx *ctx, const Pairs &P, int count) {
  size_t nmax = 0;
  for (int q = 0; q < count; q++) nmax = P.n[q] > nmax ? P.n[q] : nmax;
  if (nmax == 0) return 0;
  k_pairs<AVERAGE><<<dim3((unsigned)((nmax + 255) / 256), count), 256, 0, ctx->stream>>>(P);
  NSK_LAUNCH_CHECK();
  return 0;
}
// words 8.. of the context's reduction scratch: sweep counts of the step's solves, then the front size
inline int *step_word(nsk_ctx *ctx, int k) { return reinterpret_cast<int *>(ctx->red_dev + 8 + k); }

// On grids the sweeps-in-flight solver takes (everything fits one workgroup's LDS) the element-wise
// kernels around a solve are phases of the same launch: the workgroup walks the grid once per phase
// with a barrier where the reference has a kernel boundary.  Same expressions per cell, same
// order of the phases, so the same values; a time step is a few launches instead of dozens.
template <class F>
__device__ __forceinline__ void for_cells(int n1, int n2, F f) {
  for (int k = threadIdx.x; k < n1 * n2; k += blockDim.x) {
    const int j = k / n1;
    f(k - j * n1 + 1, j + 1);
  }
}

// DNS.jl:71-101 in one launch
template <bool FAST>
__global__ void __launch_bounds__(1024) k_dns_step(nsk_bub_dns_state s, nsk_bub_params q, double *snap, int W, int P, int M,
                                                   int *sweeps_word, double *change_word) {
  const int nx = q.nx, ny = q.ny, n1 = nx + 2, n2 = ny + 2;
  const int nt = (nx > ny ? nx : ny) + 1;
  for (int t = threadIdx.x + 1; t <= nt; t += blockDim.x)
    c_set_tangential(t, s.u, s.v, q.usouth, q.unorth, q.vwest, q.veast, nx, ny);
  for_cells(n1, n2, [&](int i, int j) { c_set_boundary(i, j, s.rt, s.r, 1000.0, nx, ny); });
  __syncthreads();
  for_cells(n1, n2, [&](int i, int j) { c_temp_vel(i, j, s.ut, s.vt, s.u, s.v, s.r, q.dt, q.dx, q.dy, q.gx, q.gy, q.m1, nx, ny); });
  __syncthreads();
  for_cells(n1, n2, [&](int i, int j) { c_compute_sources(i, j, s.tmp1, s.tmp2, s.rt, s.ut, s.vt, q.dt, q.dx, q.dy, nx, ny); });
  __syncthreads();
  const SorArgs A{s.p, s.tmp1, s.tmp2, s.rt, q.beta, q.dx, q.dy, q.maxError, nx, ny, q.maxit, sweeps_word, change_word};
  sor_pipe<0, true, FAST>(A, snap, W, P, M);
  __syncthreads();
  for_cells(n1, n2, [&](int i, int j) {
    c_correct_vel(i, j, s.u, s.v, s.ut, s.vt, s.p, s.r, q.dt, q.dx, q.dy, nx, ny);
    B2(s.ro, n1, i, j) = B2(s.r, n1, i, j);
  });
  __syncthreads();
  for_cells(n1, n2, [&](int i, int j) {
    c_advect_density(i, j, s.r, s.ro, s.u, s.v, q.dt, q.dx, q.dy, q.m1, nx, ny);
    c_velocity(i, j, s.uu, s.vv, s.u, s.v, nx, ny);
  });
}

// A value that is the same in every lane, moved to scalar registers (the compiler cannot know it
// of a value read from LDS).
template <class T>
__device__ __forceinline__ T uniform(const T &v) {
  static_assert(sizeof(T) % 4 == 0, "whole words");
  union {
    T t;
    int w[sizeof(T) / 4];
  } x;
  x.t = v;
#pragma unroll
  for (unsigned k = 0; k < sizeof(T) / 4; k++) x.w[k] = __builtin_amdgcn_readfirstlane(x.w[k]);
  return x.t;
}

// code3_fun.jl:356-494 in one launch of one workgroup: the copies of the old state, the two
// substeps (surface tension spread to the grid; boundary velocities, predictor, pressure solve,
// projection, front moved with the new velocity; density jump spread to the grid, density,
// viscosity), the second-order averaging and the regridding of the front.  words[0..3] receive the
// sweeps of the four solves, words[4] the new front size (-1: it outgrew the vectors; -2: a front
// point left the box, where the reference's next access by that point is a BoundsError,
// code3_fun.jl:144-188 -- the step ends there and the state is not to be used).
// The argument blocks are parked in LDS and every phase reads the pointers and parameters it
// needs from there: held in registers for the whole launch they would crowd the solvers' loops
// out of the scalar registers.
// Dynamic LDS: the largest of the two solvers' arrays, the front records (10 x cap values) and the
// regridding's three vectors of Nf+2.
template <bool FAST>
__global__ void __launch_bounds__(1024) k_code3_step(nsk_bub_code3_state s_arg, nsk_bub_params q_arg, double *snap, int W, int P,
                                                     int M, unsigned long long *words, double *change_word, int *nf_dev) {
  extern __shared__ double lds[];
  __shared__ nsk_bub_code3_state ls;
  __shared__ nsk_bub_params lq;
  __shared__ int s_outside;  // front points that left the box in this step
  const int tid = threadIdx.x, nth = blockDim.x;
  if (tid == 0) {
    ls = s_arg;
    lq = q_arg;
    if (nf_dev) ls.Nf = *nf_dev;  // a run of steps keeps the front size on the device
    s_outside = 0;
  }
  __syncthreads();
  auto word = [&](int k) { return reinterpret_cast<int *>(words + k); };  // 8-byte words, an int in each
  if (uniform(ls.Nf) < 0) {  // an earlier step of the run ended in an error: hand its code on
    if (tid == 0) *word(4) = ls.Nf;
    return;
  }
  {
    // un, vn, rn, mn, xfn, yfn = u, v, r, m, xf, yf (:357-362)
    const nsk_bub_code3_state s = uniform(ls);
    const int nx = uniform(lq.nx), ny = uniform(lq.ny);
    const int nu = (nx + 1) * (ny + 2), nv = (nx + 2) * (ny + 1), nc = (nx + 2) * (ny + 2);
    for (int k = tid; k < nc; k += nth) {
      if (k < nu) s.un[k] = s.u[k];
      if (k < nv) s.vn[k] = s.v[k];
      s.rn[k] = s.r[k];
      s.mn[k] = s.m[k];
    }
    for (int k = tid; k < s.cap; k += nth) {
      s.xfn[k] = s.xf[k];
      s.yfn[k] = s.yf[k];
    }
  }
  for (int sub = 0; sub < 2; sub++) {
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      for (int l = tid + 1; l <= s.Nf + 2; l += nth) c_front_tangents(l, s.tx, s.ty, s.xf, s.yf, s.Nf);
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      spread_tiles<0>(lds, s.cap, s.fx, s.fy, s.xf, s.yf, s.tx, s.ty, q.sigma, 0.0, q.dx, q.dy, q.nx, q.ny, s.Nf);
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      const int nx = q.nx, ny = q.ny, nt = (nx > ny ? nx : ny) + 1;
      for (int t = tid + 1; t <= nt; t += nth) c_set_tangential(t, s.u, s.v, q.usouth, q.unorth, q.vwest, q.veast, nx, ny);
      for_cells(nx + 2, ny + 2, [&](int i, int j) { c_set_boundary(i, j, s.rt, s.r, 1000.0, nx, ny); });
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      const int nx = q.nx, ny = q.ny;
      for_cells(nx + 2, ny + 2, [&](int i, int j) {
        c_temp_vel_adv(i, j, s.ut, s.vt, s.u, s.v, s.r, q.dt, q.dx, q.dy, s.fx, s.fy, q.gx, q.gy, q.rro, nx, ny);
        c_temp_vel_diff(i, j, s.ut, s.vt, s.u, s.v, s.r, q.dt, q.dx, q.dy, s.m, nx, ny);  // its own cell of ut, vt only
      });
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      const int nx = q.nx, ny = q.ny;
      for_cells(nx + 2, ny + 2, [&](int i, int j) { c_compute_sources(i, j, s.tmp1, s.tmp2, s.rt, s.ut, s.vt, q.dt, q.dx, q.dy, nx, ny); });
    }
    __syncthreads();
    {
      const SorArgs A{uniform(ls.p), uniform(ls.tmp1), uniform(ls.tmp2), uniform(ls.rt), uniform(lq.beta), uniform(lq.dx), uniform(lq.dy),
                      uniform(lq.maxError), uniform(lq.nx), uniform(lq.ny), uniform(lq.maxit), word(2 * sub), change_word};
      sor_pipe<0, true, FAST>(A, snap, W, P, M);
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      const int nx = q.nx, ny = q.ny;
      for_cells(nx + 2, ny + 2, [&](int i, int j) { c_correct_vel(i, j, s.u, s.v, s.ut, s.vt, s.p, s.r, q.dt, q.dx, q.dy, nx, ny); });
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      for (int l = tid + 2; l <= s.Nf + 1; l += nth)
        c_advect_move_front(l, s.uf, s.vf, s.xf, s.yf, s.u, s.v, q.dx, q.dy, q.dt, q.Lx, q.Ly, q.nx, q.ny, s.Nf, &s_outside, true);
    }
    __syncthreads();
    if (uniform(s_outside) != 0) {  // the same in every thread; nobody writes it after the barrier
      if (tid == 0) {
        *word(4) = -2;
        if (nf_dev) *nf_dev = -2;
      }
      return;
    }
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      if (tid == 0) {
        s.xf[0] = s.xf[s.Nf];
        s.yf[0] = s.yf[s.Nf];
        s.xf[s.Nf + 1] = s.xf[1];
        s.yf[s.Nf + 1] = s.yf[1];
      }
      __syncthreads();
      spread_tiles<1>(lds, s.cap, s.fx, s.fy, s.xf, s.yf, nullptr, nullptr, q.rho1, q.rho2, q.dx, q.dy, q.nx, q.ny, s.Nf);
    }
    __syncthreads();
    {
      const SorArgs A{uniform(ls.r), uniform(ls.fx), uniform(ls.fy), nullptr, uniform(lq.beta), uniform(lq.dx), uniform(lq.dy),
                      uniform(lq.maxError), uniform(lq.nx), uniform(lq.ny), uniform(lq.maxit), word(2 * sub + 1), change_word};
      sor_pipe<1, false>(A, snap, W, P, M);
    }
    __syncthreads();
    {
      const nsk_bub_code3_state s = uniform(ls);
      const nsk_bub_params q = uniform(lq);
      const int nx = q.nx, ny = q.ny;
      for_cells(nx + 2, ny + 2, [&](int i, int j) { c_update_visco(i, j, s.r, s.m, q.m1, q.m2, q.rho1, q.rho2, nx, ny); });
    }
  }
  __syncthreads();
  const nsk_bub_code3_state s = uniform(ls);
  const nsk_bub_params q = uniform(lq);
  const int nx = q.nx, ny = q.ny, Nf = s.Nf;
  const int nu = (nx + 1) * (ny + 2), nv = (nx + 2) * (ny + 1), nc = (nx + 2) * (ny + 2);
  // second-order averaging, :445-450
  for (int k = tid; k < nc; k += nth) {
    if (k < nu) s.u[k] = 0.5 * (s.u[k] + s.un[k]);
    if (k < nv) s.v[k] = 0.5 * (s.v[k] + s.vn[k]);
    s.r[k] = 0.5 * (s.r[k] + s.rn[k]);
    s.m[k] = 0.5 * (s.m[k] + s.mn[k]);
  }
  // ... of the front, straight into the regridding's copies of the old front (:455-494)
  double *s_x = lds, *s_y = lds + (Nf + 2), *s_d = lds + 2 * (Nf + 2);
  for (int k = tid; k < Nf + 2; k += nth) {
    const double x = 0.5 * (s.xf[k] + s.xfn[k]), y = 0.5 * (s.yf[k] + s.yfn[k]);
    s.xf[k] = x;
    s.yf[k] = y;
    s.xs[k] = x;
    s.ys[k] = y;
    s_x[k] = x;
    s_y[k] = y;
  }
  __syncthreads();
  for (int l = 2 + tid; l <= Nf + 1; l += nth)
    s_d[l] = sqrt(SQ((s_x[l - 1] - s_x[l - 2]) / q.dx) + SQ((s_y[l - 1] - s_y[l - 2]) / q.dy));
  __syncthreads();
  if (tid == 0) {
    const int nn = regrid_chain<true>(s.xf, s.yf, s_x, s_y, s_d, q.dx, q.dy, Nf, s.cap);
    *word(4) = nn;
    if (nf_dev) *nf_dev = nn;
  }
}

// phases of a launch walk the grid with every thread of the workgroup: as many as there are cells,
// up to the 1024 a workgroup may have (wavefronts past the solver's only wait at its barriers)
inline int merged_threads(const PipePlan &pl, int nx, int ny) {
  int n = ((nx + 2) * (ny + 2) + 63) / 64 * 64;
  if (n > 1024) n = 1024;
  return pl.nth > n ? pl.nth : n;
}
inline int host_word(nsk_ctx *ctx, int k) { return *reinterpret_cast<int *>(ctx->red_host + 8 + k); }
}  // namespace

// DNS.jl:71-101 (in the form code1.m:33-96 intends) with the lexicographic SOR.
extern "C" int nsk_bub_dns_step(nsk_ctx *ctx, const nsk_bub_dns_state *s, const nsk_bub_params *q, int *sweeps_out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(s != nullptr && q != nullptr, "nsk_bub_dns_step: null argument");
  NSK_NONNULL(s->u, s->v, s->p, s->ut, s->vt, s->tmp1, s->tmp2, s->r, s->ro, s->rt, s->uu, s->vv);
  const int nx = q->nx, ny = q->ny;
  NSK_BGRID_OK(nx, ny);
  const PipePlan pl = pipe_plan(ctx, 0, nx, ny, q->maxit);
  if (ctx->bub_merge_launches && pl.ok && pl.pre) {
    if (int rc = pipe_reserve(ctx, pl, nx, ny)) return rc;
    const auto kernel = pl.fast ? k_dns_step<true> : k_dns_step<false>;
    NSK_HIP(nsk_allow_lds(ctx, (const void *)kernel, pl.pbytes));
    // the sweep count goes straight to the pinned host word (device-visible): no copy command to wait for
    int *word = sweeps_out ? reinterpret_cast<int *>(ctx->red_host + 8) : step_word(ctx, 0);
    kernel<<<1, merged_threads(pl, nx, ny), pl.pbytes, ctx->stream>>>(*s, *q, ctx->sor_snap, pl.W, pl.P, pl.M, word,
                                                                      reinterpret_cast<double *>(ctx->red_dev + 3));
    NSK_LAUNCH_CHECK();
    if (sweeps_out) {
      NSK_HIP(hipStreamSynchronize(ctx->stream));
      *sweeps_out = host_word(ctx, 0);
    }
    return 0;
  }
  if (int rc = nsk_bub_set_tangential(ctx, s->u, s->v, q->usouth, q->unorth, q->vwest, q->veast, nx, ny)) return rc;
  if (int rc = nsk_bub_temp_vel(ctx, s->ut, s->vt, s->u, s->v, s->r, q->dt, q->dx, q->dy, q->gx, q->gy, q->m1, nx, ny)) return rc;
  if (int rc = nsk_bub_set_boundary(ctx, s->rt, s->r, 1000.0, nx, ny)) return rc;
  if (int rc = nsk_bub_compute_sources(ctx, s->tmp1, s->tmp2, s->rt, s->ut, s->vt, q->dt, q->dx, q->dy, nx, ny)) return rc;
  if (int rc = launch_sor(ctx, 0, s->p, s->tmp1, s->tmp2, s->rt, q->beta, q->dx, q->dy, nx, ny, q->maxit, q->maxError,
                          nullptr, nullptr, step_word(ctx, 0)))
    return rc;
  if (int rc = nsk_bub_correct_vel(ctx, s->u, s->v, s->ut, s->vt, s->p, s->r, q->dt, q->dx, q->dy, nx, ny)) return rc;
  NSK_HIP(hipMemcpyAsync(s->ro, s->r, sizeof(double) * (size_t)(nx + 2) * (ny + 2), hipMemcpyDeviceToDevice, ctx->stream));
  if (int rc = nsk_bub_advect_density(ctx, s->r, s->ro, s->u, s->v, q->dt, q->dx, q->dy, q->m1, nx, ny)) return rc;
  if (int rc = nsk_bub_velocity(ctx, s->uu, s->vv, s->u, s->v, nx, ny)) return rc;
  if (sweeps_out) {
    NSK_HIP(hipMemcpyAsync(ctx->red_host + 8, ctx->red_dev + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    *sweeps_out = host_word(ctx, 0);
  }
  return 0;
}

// code3_fun.jl:356-494: two substeps, second-order averaging, regridding of the front.  s->Nf is
// updated; sweeps_out[0..3] = pressure and density sweeps of the two substeps.
extern "C" int nsk_bub_code3_step(nsk_ctx *ctx, nsk_bub_code3_state *s, const nsk_bub_params *q, int *sweeps_out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(s != nullptr && q != nullptr, "nsk_bub_code3_step: null argument");
  NSK_NONNULL(s->u, s->v, s->p, s->ut, s->vt, s->tmp1, s->tmp2, s->fx, s->fy, s->un, s->vn, s->r, s->m, s->rn, s->mn, s->rt);
  NSK_NONNULL(s->xf, s->yf, s->xfn, s->yfn, s->uf, s->vf, s->tx, s->ty, s->xs, s->ys);
  const int nx = q->nx, ny = q->ny, Nf = s->Nf;
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(Nf >= 1 && s->cap >= Nf + 2, "nsk_bub_code3_step: front of %d points in vectors of %d", Nf, s->cap);
  ctx->code3_steps_done = 0;
  const size_t nu = (size_t)(nx + 1) * (ny + 2), nv = (size_t)(nx + 2) * (ny + 1),
               nc = (size_t)(nx + 2) * (ny + 2), nf = (size_t)Nf + 2;
  {
    // One launch for the whole step when one workgroup holds the solvers' arrays and the front in LDS.
    // The two solvers share W, P, M (they depend on the grid and maxit only).
    const PipePlan plp = pipe_plan(ctx, 0, nx, ny, q->maxit), pld = pipe_plan(ctx, 1, nx, ny, q->maxit);
    size_t lds_bytes = plp.pbytes > pld.pbytes ? plp.pbytes : pld.pbytes;
    const size_t front_bytes = sizeof(double) * 10 * (size_t)s->cap;  // >= the regridding's 3 (Nf+2) values
    if (front_bytes > lds_bytes) lds_bytes = front_bytes;
    if (ctx->bub_merge_launches && plp.ok && plp.pre && pld.ok && lds_bytes <= 156 * 1024) {
      if (int rc = pipe_reserve(ctx, plp, nx, ny)) return rc;
      const auto kernel = plp.fast ? k_code3_step<true> : k_code3_step<false>;
      NSK_HIP(nsk_allow_lds(ctx, (const void *)kernel, lds_bytes));
      // sweep counts and front size go straight to the pinned host words (device-visible)
      kernel<<<1, merged_threads(plp, nx, ny), lds_bytes, ctx->stream>>>(*s, *q, ctx->sor_snap, plp.W, plp.P, plp.M, ctx->red_host + 8,
                                                                         reinterpret_cast<double *>(ctx->red_dev + 3), nullptr);
      NSK_LAUNCH_CHECK();
      NSK_HIP(hipStreamSynchronize(ctx->stream));
      const int nf_new = host_word(ctx, 4);
      if (sweeps_out)
        for (int k = 0; k < 4; k++) sweeps_out[k] = host_word(ctx, k);
      NSK_REQUIRE_RC(nf_new != -2, NSK_ERR_FRONT_OUTSIDE, "nsk_bub_code3_step: a front point left the box [0,%g] x [0,%g]; the state is not to be used", q->Lx, q->Ly);
      NSK_REQUIRE_RC(nf_new >= 0, NSK_ERR_FRONT_CAPACITY, "nsk_bub_code3_step: the front outgrew its capacity of %d points", s->cap);
      s->Nf = nf_new;
      ctx->code3_steps_done = 1;
      return 0;
    }
  }
  // kernel by kernel: front points that leave the box are counted in a device word read back with the others
  int *d_outside = step_word(ctx, 5);
  NSK_HIP(hipMemsetAsync(d_outside, 0, sizeof(int), ctx->stream));
  // un, vn, rn, mn, xfn, yfn = u, v, r, m, xf, yf (:357-362) in one launch
  const Pairs keep{{s->un, s->vn, s->rn, s->mn, s->xfn, s->yfn}, {s->u, s->v, s->r, s->m, s->xf, s->yf},
                   {nu, nv, nc, nc, (size_t)s->cap, (size_t)s->cap}};
  if (int rc = launch_pairs<false>(ctx, keep, 6)) return rc;
  for (int sub = 0; sub < 2; sub++) {
    if (int rc = nsk_bub_surface_tension(ctx, s->fx, s->fy, s->tx, s->ty, s->xf, s->yf, q->sigma, q->dx, q->dy, nx, ny, Nf)) return rc;
    if (int rc = nsk_bub_set_tangential(ctx, s->u, s->v, q->usouth, q->unorth, q->vwest, q->veast, nx, ny)) return rc;
    if (int rc = nsk_bub_temp_vel_adv(ctx, s->ut, s->vt, s->u, s->v, s->r, q->dt, q->dx, q->dy, s->fx, s->fy, q->gx, q->gy, q->rro, nx, ny)) return rc;
    if (int rc = nsk_bub_temp_vel_diff(ctx, s->ut, s->vt, s->u, s->v, s->r, q->dt, q->dx, q->dy, s->m, nx, ny)) return rc;
    if (int rc = nsk_bub_set_boundary(ctx, s->rt, s->r, 1000.0, nx, ny)) return rc;
    if (int rc = nsk_bub_compute_sources(ctx, s->tmp1, s->tmp2, s->rt, s->ut, s->vt, q->dt, q->dx, q->dy, nx, ny)) return rc;
    if (int rc = launch_sor(ctx, 0, s->p, s->tmp1, s->tmp2, s->rt, q->beta, q->dx, q->dy, nx, ny, q->maxit, q->maxError,
                            nullptr, nullptr, step_word(ctx, 2 * sub)))
      return rc;
    if (int rc = nsk_bub_correct_vel(ctx, s->u, s->v, s->ut, s->vt, s->p, s->r, q->dt, q->dx, q->dy, nx, ny)) return rc;
    k_advect_move_front<<<nsk_cdiv(Nf, 256), 256, 0, ctx->stream>>>(s->uf, s->vf, s->xf, s->yf, s->u, s->v, q->dx, q->dy, q->dt, q->Lx,
                                                                   q->Ly, nx, ny, Nf, d_outside, true);
    NSK_LAUNCH_CHECK();
    k_close_front<<<1, 64, 0, ctx->stream>>>(s->xf, s->yf, Nf);
    NSK_LAUNCH_CHECK();
    if (int rc = nsk_bub_distribute(ctx, s->fx, s->fy, s->xf, s->yf, q->rho1, q->rho2, q->dx, q->dy, nx, ny, Nf)) return rc;
    if (int rc = launch_sor(ctx, 1, s->r, s->fx, s->fy, nullptr, q->beta, q->dx, q->dy, nx, ny, q->maxit, q->maxError,
                            nullptr, nullptr, step_word(ctx, 2 * sub + 1)))
      return rc;
    if (int rc = nsk_bub_update_visco(ctx, s->r, s->m, q->m1, q->m2, q->rho1, q->rho2, nx, ny)) return rc;
  }
  // second-order averaging, code3_fun.jl:445-450
  const Pairs avg{{s->u, s->v, s->r, s->m, s->xf, s->yf}, {s->un, s->vn, s->rn, s->mn, s->xfn, s->yfn},
                  {nu, nv, nc, nc, nf, nf}};
  if (int rc = launch_pairs<true>(ctx, avg, 6)) return rc;
  // regridding, :455-494; the front size travels with the sweep counts
  int *d_nf = step_word(ctx, 4);
  *reinterpret_cast<int *>(ctx->red_host + 16) = Nf;
  NSK_HIP(hipMemcpyAsync(d_nf, ctx->red_host + 16, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  k_regrid_front<<<1, 64, 0, ctx->stream>>>(s->xf, s->yf, s->xs, s->ys, q->dx, q->dy, d_nf, s->cap);
  NSK_LAUNCH_CHECK();
  NSK_HIP(hipMemcpyAsync(ctx->red_host + 8, ctx->red_dev + 8, 6 * 8, hipMemcpyDeviceToHost, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  // (the clamped indices of the interpolation kept every access of the rest of the step inside the arrays)
  if (sweeps_out)
    for (int k = 0; k < 4; k++) sweeps_out[k] = host_word(ctx, k);
  NSK_REQUIRE_RC(host_word(ctx, 5) == 0, NSK_ERR_FRONT_OUTSIDE, "nsk_bub_code3_step: a front point left the box [0,%g] x [0,%g]; the state is not to be used", q->Lx, q->Ly);
  const int nf_new = host_word(ctx, 4);
  NSK_REQUIRE_RC(nf_new >= 0, NSK_ERR_FRONT_CAPACITY, "nsk_bub_code3_step: the front outgrew its capacity of %d points", s->cap);
  s->Nf = nf_new;
  ctx->code3_steps_done = 1;
  return 0;
}

// ---- time loops ----------------------------------------------------------------------------------
// The reference's drivers are loops over time steps (DNS.jl:70-102, code3_fun.jl:354-494).  On grids
// where a time step is one launch, a run of steps is queued back to back: nothing of a step is
// needed on the host before the next one starts (code3's front size stays on the device), so the
// host waits once per run instead of once per step.
namespace {
// eight words per step of a run: the step's sweep counts and, for code3, the front size after it
int run_words(nsk_ctx *ctx, int nsteps, unsigned long long **words) {
  const size_t need = (size_t)nsteps * 8;
  if (ctx->run_words_cap < need) {
    if (ctx->run_words) NSK_HIP(hipFree(ctx->run_words));
    ctx->run_words = nullptr;
    ctx->run_words_cap = 0;
    NSK_HIP(hipMalloc(&ctx->run_words, need * sizeof(unsigned long long)));
    ctx->run_words_cap = need;
  }
  *words = ctx->run_words;
  return 0;
}
}  // namespace

extern "C" int nsk_bub_dns_run(nsk_ctx *ctx, const nsk_bub_dns_state *s, const nsk_bub_params *q, int nsteps, int *sweeps_out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(s != nullptr && q != nullptr && nsteps >= 0, "nsk_bub_dns_run: bad arguments (nsteps=%d)", nsteps);
  NSK_NONNULL(s->u, s->v, s->p, s->ut, s->vt, s->tmp1, s->tmp2, s->r, s->ro, s->rt, s->uu, s->vv);
  const int nx = q->nx, ny = q->ny;
  NSK_BGRID_OK(nx, ny);
  const PipePlan pl = pipe_plan(ctx, 0, nx, ny, q->maxit);
  if (!(ctx->bub_merge_launches && pl.ok && pl.pre) || nsteps == 0) {
    for (int k = 0; k < nsteps; k++)
      if (int rc = nsk_bub_dns_step(ctx, s, q, sweeps_out ? sweeps_out + k : nullptr)) return rc;
    return 0;
  }
  unsigned long long *words = nullptr;
  if (int rc = run_words(ctx, nsteps, &words)) return rc;
  if (int rc = pipe_reserve(ctx, pl, nx, ny)) return rc;
  const auto kernel = pl.fast ? k_dns_step<true> : k_dns_step<false>;
  NSK_HIP(nsk_allow_lds(ctx, (const void *)kernel, pl.pbytes));
  for (int k = 0; k < nsteps; k++) {
    kernel<<<1, merged_threads(pl, nx, ny), pl.pbytes, ctx->stream>>>(*s, *q, ctx->sor_snap, pl.W, pl.P, pl.M,
                                                                      reinterpret_cast<int *>(words + 8 * (size_t)k),
                                                                      reinterpret_cast<double *>(ctx->red_dev + 3));
    NSK_LAUNCH_CHECK();
  }
  if (sweeps_out) {
    std::vector<unsigned long long> host((size_t)nsteps * 8);
    NSK_HIP(hipMemcpyAsync(host.data(), words, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < nsteps; k++) sweeps_out[k] = (int)(unsigned)host[8 * (size_t)k];
  }
  return 0;
}

extern "C" int nsk_bub_code3_run(nsk_ctx *ctx, nsk_bub_code3_state *s, const nsk_bub_params *q, int nsteps, int *sweeps_out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(s != nullptr && q != nullptr && nsteps >= 0, "nsk_bub_code3_run: bad arguments (nsteps=%d)", nsteps);
  NSK_NONNULL(s->u, s->v, s->p, s->ut, s->vt, s->tmp1, s->tmp2, s->fx, s->fy, s->un, s->vn, s->r, s->m, s->rn, s->mn, s->rt);
  NSK_NONNULL(s->xf, s->yf, s->xfn, s->yfn, s->uf, s->vf, s->tx, s->ty, s->xs, s->ys);
  const int nx = q->nx, ny = q->ny;
  NSK_BGRID_OK(nx, ny);
  NSK_REQUIRE(s->Nf >= 1 && s->cap >= s->Nf + 2, "nsk_bub_code3_run: front of %d points in vectors of %d", s->Nf, s->cap);
  const PipePlan plp = pipe_plan(ctx, 0, nx, ny, q->maxit), pld = pipe_plan(ctx, 1, nx, ny, q->maxit);
  size_t lds_bytes = plp.pbytes > pld.pbytes ? plp.pbytes : pld.pbytes;
  const size_t front_bytes = sizeof(double) * 10 * (size_t)s->cap;
  if (front_bytes > lds_bytes) lds_bytes = front_bytes;
  if (!(ctx->bub_merge_launches && plp.ok && plp.pre && pld.ok && lds_bytes <= 156 * 1024) || nsteps == 0) {
    int done = 0;
    for (int k = 0; k < nsteps; k++, done++)
      if (int rc = nsk_bub_code3_step(ctx, s, q, sweeps_out ? sweeps_out + 4 * k : nullptr)) {
        ctx->code3_steps_done = done;
        return rc;
      }
    ctx->code3_steps_done = done;
    return 0;
  }
  ctx->code3_steps_done = 0;
  unsigned long long *words = nullptr;
  if (int rc = run_words(ctx, nsteps, &words)) return rc;
  if (int rc = pipe_reserve(ctx, plp, nx, ny)) return rc;
  const auto kernel = plp.fast ? k_code3_step<true> : k_code3_step<false>;
  NSK_HIP(nsk_allow_lds(ctx, (const void *)kernel, lds_bytes));
  // the front size travels from step to step in a device word
  int *nf_dev = reinterpret_cast<int *>(ctx->red_dev + 20);
  *reinterpret_cast<int *>(ctx->red_host + 20) = s->Nf;
  NSK_HIP(hipMemcpyAsync(nf_dev, ctx->red_host + 20, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  for (int k = 0; k < nsteps; k++) {
    kernel<<<1, merged_threads(plp, nx, ny), lds_bytes, ctx->stream>>>(*s, *q, ctx->sor_snap, plp.W, plp.P, plp.M, words + 8 * (size_t)k,
                                                                       reinterpret_cast<double *>(ctx->red_dev + 3), nf_dev);
    NSK_LAUNCH_CHECK();
  }
  std::vector<unsigned long long> host((size_t)nsteps * 8);
  NSK_HIP(hipMemcpyAsync(host.data(), words, host.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < nsteps; k++) {
    const int nf = (int)(unsigned)host[8 * (size_t)k + 4];
    if (sweeps_out)
      for (int j = 0; j < 4; j++) sweeps_out[4 * k + j] = (int)(unsigned)host[8 * (size_t)k + j];
    NSK_REQUIRE_RC(nf != -2, NSK_ERR_FRONT_OUTSIDE, "nsk_bub_code3_run: a front point left the box [0,%g] x [0,%g] in step %d of the run; the state is not to be used", q->Lx, q->Ly, k + 1);
    NSK_REQUIRE_RC(nf >= 0, NSK_ERR_FRONT_CAPACITY, "nsk_bub_code3_run: the front outgrew its capacity of %d points in step %d of the run", s->cap, k + 1);
    s->Nf = nf;
    ctx->code3_steps_done = k + 1;
  }
  return 0;
}

extern "C" int nsk_bub_code3_run_status(nsk_ctx *ctx, int *steps_done) {
  NSK_REQUIRE(ctx != nullptr && steps_done != nullptr, "nsk_bub_code3_run_status: null argument");
  *steps_done = ctx->code3_steps_done;
  return 0;
}
