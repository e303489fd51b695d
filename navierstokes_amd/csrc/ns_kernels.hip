// Kernel-granular entry points: one HIP kernel per @parallel call site of the
// reference (scripts/NavierStokes.jl:115-250).  One thread per grid point,
// ix fastest (64 consecutive ix per wavefront row => coalesced 512-byte rows).
// Built with -ffp-contract=off: every expression keeps the reference's
// association order and no multiply-add is fused.
#include <cstring>

#include "nsk_internal.h"

namespace {

constexpr int BX = 64, BY = 4;

inline dim3 grid2(int n1, int n2) { return dim3(nsk_cdiv(n1, BX), nsk_cdiv(n2, BY)); }
inline dim3 block2() { return dim3(BX, BY); }

#define IXIY                                         \
  const int ix = blockIdx.x * BX + threadIdx.x + 1;  \
  const int iy = blockIdx.y * BY + threadIdx.y + 1

// NavierStokes.jl:116-121
__global__ void k_update_tau(double *__restrict__ txx, double *__restrict__ tyy,
                             double *__restrict__ txy, const double *__restrict__ Vx,
                             const double *__restrict__ Vy, double mu, double dx, double dy,
                             int nx, int ny) {
  IXIY;
  if (ix > nx || iy > ny) return;
  const double mu2 = 2 * mu;
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

// NavierStokes.jl:123-127
__global__ void k_predict_V(double *__restrict__ Vx, double *__restrict__ Vy,
                            const double *__restrict__ txx, const double *__restrict__ tyy,
                            const double *__restrict__ txy, double dtr, double rg, double dx,
                            double dy, int nx, int ny) {
  IXIY;
  if (ix > nx - 1 || iy > ny - 1) return;
  if (iy <= ny - 2) {
    double a = A2(txx, nx, ix + 1, iy + 1) - A2(txx, nx, ix, iy + 1);
    double b = A2(txy, nx - 1, ix, iy + 1) - A2(txy, nx - 1, ix, iy);
    A2(Vx, nx + 1, ix + 1, iy + 1) = A2(Vx, nx + 1, ix + 1, iy + 1) + dtr * (a / dx + b / dy);
  }
  if (ix <= nx - 2) {
    double a = A2(tyy, nx, ix + 1, iy + 1) - A2(tyy, nx, ix + 1, iy);
    double b = A2(txy, nx - 1, ix + 1, iy) - A2(txy, nx - 1, ix, iy);
    A2(Vy, nx, ix + 1, iy + 1) = A2(Vy, nx, ix + 1, iy + 1) + dtr * (a / dy + b / dx - rg);
  }
}

// NavierStokes.jl:129-132
__global__ void k_update_divV(double *__restrict__ divV, int ld, const double *__restrict__ Vx,
                              const double *__restrict__ Vy, double dx, double dy, int nx,
                              int ny) {
  IXIY;
  if (ix > nx || iy > ny) return;
  A2(divV, ld, ix, iy) = (A2(Vx, nx + 1, ix + 1, iy) - A2(Vx, nx + 1, ix, iy)) / dx +
                         (A2(Vy, nx, ix, iy + 1) - A2(Vy, nx, ix, iy)) / dy;
}

__device__ __forceinline__ double residual_at(const double *__restrict__ Pr,
                                              const double *__restrict__ divV, double rho_dt,
                                              double dx, double dy, int nx, int ix, int iy) {
  double c = A2(Pr, nx, ix + 1, iy + 1);
  double d2x = (A2(Pr, nx, ix + 2, iy + 1) - c) - (c - A2(Pr, nx, ix, iy + 1));
  double d2y = (A2(Pr, nx, ix + 1, iy + 2) - c) - (c - A2(Pr, nx, ix + 1, iy));
  return d2x / dx / dx + d2y / dy / dy - rho_dt * A2(divV, nx, ix + 1, iy + 1);
}

// NavierStokes.jl:134-137
__global__ void k_update_dPrdtau(const double *__restrict__ Pr, double *__restrict__ dP,
                                 const double *__restrict__ divV, double rho_dt, double dtau,
                                 double keep, double dx, double dy, int nx, int ny) {
  IXIY;
  if (ix > nx - 2 || iy > ny - 2) return;
  A2(dP, nx - 2, ix, iy) =
      A2(dP, nx - 2, ix, iy) * keep + dtau * residual_at(Pr, divV, rho_dt, dx, dy, nx, ix, iy);
}

// NavierStokes.jl:139-142
__global__ void k_update_Pr(double *__restrict__ Pr, const double *__restrict__ dP, double dtau,
                            int nx, int ny) {
  IXIY;
  if (ix > nx - 2 || iy > ny - 2) return;
  A2(Pr, nx, ix + 1, iy + 1) = A2(Pr, nx, ix + 1, iy + 1) + dtau * A2(dP, nx - 2, ix, iy);
}

// NavierStokes.jl:144-147
__global__ void k_compute_res(double *__restrict__ Rp, const double *__restrict__ Pr,
                              const double *__restrict__ divV, double rho_dt, double dx,
                              double dy, int nx, int ny) {
  IXIY;
  if (ix > nx - 2 || iy > ny - 2) return;
  A2(Rp, nx - 2, ix, iy) = residual_at(Pr, divV, rho_dt, dx, dy, nx, ix, iy);
}

// NavierStokes.jl:149-153
__global__ void k_correct_V(double *__restrict__ Vx, double *__restrict__ Vy,
                            const double *__restrict__ Pr, int ld, double dtr, double dx,
                            double dy, int nx, int ny) {
  IXIY;
  if (ix > nx - 1 || iy > ny - 1) return;
  if (iy <= ny - 2)
    A2(Vx, nx + 1, ix + 1, iy + 1) =
        A2(Vx, nx + 1, ix + 1, iy + 1) -
        dtr * (A2(Pr, ld, ix + 1, iy + 1) - A2(Pr, ld, ix, iy + 1)) / dx;
  if (ix <= nx - 2)
    A2(Vy, nx, ix + 1, iy + 1) =
        A2(Vy, nx, ix + 1, iy + 1) -
        dtr * (A2(Pr, ld, ix + 1, iy + 1) - A2(Pr, ld, ix + 1, iy)) / dy;
}

// NavierStokes.jl:155-159 (bc_x!): rows 1..nrows of an (n1 x n2) array.
__global__ void k_bc_x(double *__restrict__ A, int n1, int nrows) {
  const int iy = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (iy > nrows) return;
  A2(A, n1, 1, iy) = A2(A, n1, 2, iy);
  A2(A, n1, n1, iy) = A2(A, n1, n1 - 1, iy);
}

// NavierStokes.jl:161-177 (bc_y!, bc_yV!, bc_yval!) in one kernel:
// bottom: 0 none, 1 copy row 2, 2 take V[ix];  top: 0 none, 1 copy row n2-1, 2 set val.
__global__ void k_bc_y(double *__restrict__ A, const double *__restrict__ V, double val, int n1,
                       int n2, int bottom, int top) {
  const int ix = blockIdx.x * blockDim.x + threadIdx.x + 1;
  if (ix > n1) return;
  if (bottom == 1) A2(A, n1, ix, 1) = A2(A, n1, ix, 2);
  if (bottom == 2) A2(A, n1, ix, 1) = V[ix - 1];
  if (top == 1) A2(A, n1, ix, n2) = A2(A, n1, ix, n2 - 1);
  if (top == 2) A2(A, n1, ix, n2) = val;
}

__device__ __forceinline__ double lerp_ref(double a, double b, double t) {
  return b * t + a * (1 - t);  // NavierStokes.jl:204
}

__device__ __forceinline__ int clamp_floor(double f, int lo, int hi) {
  if (!(f >= (double)lo)) return lo;
  if (f >= (double)hi) return hi;
  return (int)f;
}

// NavierStokes.jl:192-202.  (ix, gy) are global 1-based indices; the source
// array window holds global rows iy0+1 .. iy0+n2w of an array with n2g rows.
__device__ __forceinline__ double backtrack(const double *__restrict__ A_o, int n1, int n2w,
                                            int n2g, int iy0, double vxc, double vyc, double dt,
                                            double dx, double dy, int ix, int gy) {
  double ddx = dt * vxc / dx, ddy = dt * vyc / dy;
  int ix1 = clamp_floor(floor((double)ix - ddx), 1, n1);
  int iy1 = clamp_floor(floor((double)gy - ddy), 1, n2g);
  int ix2 = min(ix1 + 1, n1);
  int iy2 = min(iy1 + 1, n2g);
  // to window rows; the extra clamp only acts when the window is too narrow
  iy1 = min(max(iy1 - iy0, 1), n2w);
  iy2 = min(max(iy2 - iy0, 1), n2w);
  ddx = (double)(ddx > 0) - fmod(ddx, 1.0);
  ddy = (double)(ddy > 0) - fmod(ddy, 1.0);
  double fx1 = lerp_ref(A2(A_o, n1, ix1, iy1), A2(A_o, n1, ix2, iy1), ddx);
  double fx2 = lerp_ref(A2(A_o, n1, ix1, iy2), A2(A_o, n1, ix2, iy2), ddx);
  return lerp_ref(fx1, fx2, ddy);
}

// NavierStokes.jl:206-223
__global__ void k_advect(double *__restrict__ Vx, const double *__restrict__ Vx_o,
                         double *__restrict__ Vy, const double *__restrict__ Vy_o,
                         double *__restrict__ C, const double *__restrict__ C_o, double dt,
                         double dx, double dy, int nx, int ny, int iy0, int nyg) {
  IXIY;
  if (ix > nx + 1 || iy > ny + 1) return;
  const int gy = iy + iy0;
  if (ix > 1 && ix < nx + 1 && iy <= ny) {
    double vxc = A2(Vx_o, nx + 1, ix, iy);
    double vyc = 0.25 * (A2(Vy_o, nx, ix - 1, iy) + A2(Vy_o, nx, ix - 1, iy + 1) +
                         A2(Vy_o, nx, ix, iy) + A2(Vy_o, nx, ix, iy + 1));
    A2(Vx, nx + 1, ix, iy) = backtrack(Vx_o, nx + 1, ny, nyg, iy0, vxc, vyc, dt, dx, dy, ix, gy);
  }
  if (gy > 1 && gy < nyg + 1 && iy > 1 && iy <= ny && ix <= nx) {
    double vxc = 0.25 * (A2(Vx_o, nx + 1, ix, iy - 1) + A2(Vx_o, nx + 1, ix + 1, iy - 1) +
                         A2(Vx_o, nx + 1, ix, iy) + A2(Vx_o, nx + 1, ix + 1, iy));
    double vyc = A2(Vy_o, nx, ix, iy);
    A2(Vy, nx, ix, iy) = backtrack(Vy_o, nx, ny + 1, nyg + 1, iy0, vxc, vyc, dt, dx, dy, ix, gy);
  }
  if (ix <= nx && iy <= ny) {
    double vxc = 0.5 * (A2(Vx_o, nx + 1, ix, iy) + A2(Vx_o, nx + 1, ix + 1, iy));
    double vyc = 0.5 * (A2(Vy_o, nx, ix, iy) + A2(Vy_o, nx, ix, iy + 1));
    A2(C, nx, ix, iy) = backtrack(C_o, nx, ny, nyg, iy0, vxc, vyc, dt, dx, dy, ix, gy);
  }
}

// NavierStokes.jl:225-250
__global__ void k_set_sphere(double *__restrict__ C, double *__restrict__ Vx,
                             double *__restrict__ Vy, double a2, double b2, double ox, double oy,
                             double sinb, double cosb, double lx, double ly, double dx, double dy,
                             int nx, int ny, int iy0) {
  IXIY;
  if (ix > nx + 1 || iy > ny + 1) return;
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

// maximum(abs.(A)) (NavierStokes.jl:90).  |x| has a clear sign bit, so the
// IEEE order of the values equals the unsigned order of their bit patterns and
// every NaN pattern sorts above +Inf: an integer max propagates NaN exactly as
// Julia's maximum does.  Wavefront shuffle, then LDS across the 4 waves, then
// one atomic per block.
__global__ void __launch_bounds__(256) k_maxabs(const double *__restrict__ a, size_t n,
                                                unsigned long long *__restrict__ out) {
  unsigned long long m = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long b = (unsigned long long)__double_as_longlong(fabs(a[i]));
    m = b > m ? b : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    unsigned long long o = __shfl_xor(m, off, 64);
    m = o > m ? o : m;
  }
  __shared__ unsigned long long s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) m = s[w] > m ? s[w] : m;
    atomicMax(out, m);
  }
}

}  // namespace

// Shared with the fused entry points.
int nsk_launch_maxabs(nsk_ctx *ctx, const double *A, size_t count) {
  NSK_HIP(hipMemsetAsync(ctx->red_dev, 0, sizeof(unsigned long long), ctx->stream));
  if (count > 0) {
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    k_maxabs<<<blocks, 256, 0, ctx->stream>>>(A, count, ctx->red_dev);
    NSK_LAUNCH_CHECK();
  }
  return 0;
}

int nsk_read_maxabs(nsk_ctx *ctx, double *out) {
  NSK_HIP(hipMemcpyAsync(ctx->red_host, ctx->red_dev, sizeof(unsigned long long),
                         hipMemcpyDeviceToHost, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  if (int rc = nsk_device_error_check(ctx, "residual")) return rc;
  unsigned long long b = *ctx->red_host;
  double d;
  memcpy(&d, &b, sizeof d);
  *out = d;
  return 0;
}

#define NSK_GRID_OK(nx, ny) \
  NSK_REQUIRE((nx) >= 3 && (ny) >= 3, "grid too small: nx=%d ny=%d (need >= 3)", (nx), (ny))

extern "C" int nsk_update_tau(nsk_ctx *ctx, double *txx, double *tyy, double *txy,
                              const double *Vx, const double *Vy, double mu, double dx, double dy,
                              int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(txx, tyy, txy, Vx, Vy);
  NSK_GRID_OK(nx, ny);
  k_update_tau<<<grid2(nx, ny), block2(), 0, ctx->stream>>>(txx, tyy, txy, Vx, Vy, mu, dx, dy, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_predict_V(nsk_ctx *ctx, double *Vx, double *Vy, const double *txx,
                             const double *tyy, const double *txy, double rho, double g, double dt,
                             double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(Vx, Vy, txx, tyy, txy);
  NSK_GRID_OK(nx, ny);
  k_predict_V<<<grid2(nx - 1, ny - 1), block2(), 0, ctx->stream>>>(Vx, Vy, txx, tyy, txy, dt / rho,
                                                                 rho * g, dx, dy, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_set_sphere(nsk_ctx *ctx, double *C, double *Vx, double *Vy, double a2, double b2,
                              double ox, double oy, double sinb, double cosb, double lx, double ly,
                              double dx, double dy, int nx, int ny, int iy0) {
  NSK_CTX(ctx);
  NSK_NONNULL(C, Vx, Vy);
  NSK_GRID_OK(nx, ny);
  k_set_sphere<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(C, Vx, Vy, a2, b2, ox, oy, sinb,
                                                                  cosb, lx, ly, dx, dy, nx, ny, iy0);
  NSK_LAUNCH_CHECK();
  return 0;
}

// Pitched variant (ld = leading dimension of divV).
extern "C" int nsk_update_divV_ld(nsk_ctx *ctx, double *divV, int ld, const double *Vx,
                                  const double *Vy, double dx, double dy, int nx, int ny);
int nsk_i_update_divV(nsk_ctx *ctx, double *divV, int ld, const double *Vx, const double *Vy,
                      double dx, double dy, int nx, int ny) {
  k_update_divV<<<grid2(nx, ny), block2(), 0, ctx->stream>>>(divV, ld, Vx, Vy, dx, dy, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_update_divV(nsk_ctx *ctx, double *divV, const double *Vx, const double *Vy,
                               double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(divV, Vx, Vy);
  NSK_GRID_OK(nx, ny);
  return nsk_i_update_divV(ctx, divV, nx, Vx, Vy, dx, dy, nx, ny);
}

extern "C" int nsk_update_dPrdtau(nsk_ctx *ctx, const double *Pr, double *dPrdtau,
                                  const double *divV, double rho, double dt, double dtau,
                                  double damp, double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, dPrdtau, divV);
  NSK_GRID_OK(nx, ny);
  k_update_dPrdtau<<<grid2(nx - 2, ny - 2), block2(), 0, ctx->stream>>>(
      Pr, dPrdtau, divV, rho / dt, dtau, 1.0 - damp, dx, dy, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_update_Pr(nsk_ctx *ctx, double *Pr, const double *dPrdtau, double dtau, int nx,
                             int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, dPrdtau);
  NSK_GRID_OK(nx, ny);
  k_update_Pr<<<grid2(nx - 2, ny - 2), block2(), 0, ctx->stream>>>(Pr, dPrdtau, dtau, nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

// Rows (window-local, 1-based) that bc_x! reaches for an array with n1 x n2g
// global extents, restricted to a window of n2w rows starting at global row iy0+1.
static int bc_x_rows(int n1, int n2g, int n2w, int iy0, int full_rows) {
  int lim = full_rows ? n2g : (n1 < n2g ? n1 : n2g);
  int rows = lim - iy0;
  if (rows < 0) rows = 0;
  if (rows > n2w) rows = n2w;
  return rows;
}

extern "C" int nsk_set_bc_Pr(nsk_ctx *ctx, double *Pr, double val, int nx, int ny, int full_rows,
                             int iy0, int nyg) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr);
  NSK_GRID_OK(nx, ny);
  NSK_REQUIRE(iy0 >= 0 && iy0 + ny <= nyg, "set_bc_Pr: window [%d,%d] outside 1..%d", iy0 + 1,
              iy0 + ny, nyg);
  int rows = bc_x_rows(nx, nyg, ny, iy0, full_rows);
  if (rows > 0) {
    k_bc_x<<<nsk_cdiv(rows, 256), 256, 0, ctx->stream>>>(Pr, nx, rows);
    NSK_LAUNCH_CHECK();
  }
  int bottom = iy0 == 0 ? 1 : 0, top = iy0 + ny == nyg ? 2 : 0;
  if (bottom || top) {
    k_bc_y<<<nsk_cdiv(nx, 256), 256, 0, ctx->stream>>>(Pr, nullptr, val, nx, ny, bottom, top);
    NSK_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int nsk_compute_res(nsk_ctx *ctx, double *Rp, const double *Pr, const double *divV,
                               double rho, double dt, double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(Rp, Pr, divV);
  NSK_GRID_OK(nx, ny);
  k_compute_res<<<grid2(nx - 2, ny - 2), block2(), 0, ctx->stream>>>(Rp, Pr, divV, rho / dt, dx, dy,
                                                                   nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_maxabs(nsk_ctx *ctx, const double *A, size_t count, double *out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(out != nullptr, "nsk_maxabs: out is null");
  if (count > 0) NSK_NONNULL(A);
  if (int rc = nsk_launch_maxabs(ctx, A, count)) return rc;
  return nsk_read_maxabs(ctx, out);
}

// Pitched variant (ld = leading dimension of Pr).
extern "C" int nsk_correct_V_ld(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr, int ld,
                                double dt, double rho, double dx, double dy, int nx, int ny);
int nsk_i_correct_V(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr, int ld, double dt,
                    double rho, double dx, double dy, int nx, int ny) {
  k_correct_V<<<grid2(nx - 1, ny - 1), block2(), 0, ctx->stream>>>(Vx, Vy, Pr, ld, dt / rho, dx, dy,
                                                                 nx, ny);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_correct_V(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr, double dt,
                             double rho, double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(Vx, Vy, Pr);
  NSK_GRID_OK(nx, ny);
  return nsk_i_correct_V(ctx, Vx, Vy, Pr, nx, dt, rho, dx, dy, nx, ny);
}

extern "C" int nsk_set_bc_Vel(nsk_ctx *ctx, double *Vx, double *Vy, const double *Vprof, int nx,
                              int ny, int full_rows, int iy0, int nyg) {
  NSK_CTX(ctx);
  NSK_NONNULL(Vx, Vy, Vprof);
  NSK_GRID_OK(nx, ny);
  NSK_REQUIRE(iy0 >= 0 && iy0 + ny <= nyg, "set_bc_Vel: window [%d,%d] outside 1..%d", iy0 + 1,
              iy0 + ny, nyg);
  const int bottom = iy0 == 0, top = iy0 + ny == nyg;
  if (bottom || top) {  // bc_y!(Vx)
    k_bc_y<<<nsk_cdiv(nx + 1, 256), 256, 0, ctx->stream>>>(Vx, nullptr, 0.0, nx + 1, ny, bottom ? 1 : 0,
                                                         top ? 1 : 0);
    NSK_LAUNCH_CHECK();
  }
  int rows = bc_x_rows(nx, nyg + 1, ny + 1, iy0, full_rows);  // bc_x!(Vy)
  if (rows > 0) {
    k_bc_x<<<nsk_cdiv(rows, 256), 256, 0, ctx->stream>>>(Vy, nx, rows);
    NSK_LAUNCH_CHECK();
  }
  if (bottom || top) {  // bc_yV!(Vy, Vprof)
    k_bc_y<<<nsk_cdiv(nx, 256), 256, 0, ctx->stream>>>(Vy, Vprof, 0.0, nx, ny + 1, bottom ? 2 : 0,
                                                     top ? 1 : 0);
    NSK_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int nsk_advect(nsk_ctx *ctx, double *Vx, const double *Vx_o, double *Vy,
                          const double *Vy_o, double *C, const double *C_o, double dt, double dx,
                          double dy, int nx, int ny, int iy0, int nyg) {
  NSK_CTX(ctx);
  NSK_NONNULL(Vx, Vx_o, Vy, Vy_o, C, C_o);
  NSK_GRID_OK(nx, ny);
  NSK_REQUIRE(iy0 >= 0 && iy0 + ny <= nyg, "advect: window [%d,%d] outside 1..%d", iy0 + 1,
              iy0 + ny, nyg);
  k_advect<<<grid2(nx + 1, ny + 1), block2(), 0, ctx->stream>>>(Vx, Vx_o, Vy, Vy_o, C, C_o, dt, dx,
                                                              dy, nx, ny, iy0, nyg);
  NSK_LAUNCH_CHECK();
  return 0;
}

extern "C" int nsk_update_divV_ld(nsk_ctx *ctx, double *divV, int ld, const double *Vx,
                                  const double *Vy, double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(divV, Vx, Vy);
  NSK_GRID_OK(nx, ny);
  NSK_REQUIRE(ld >= nx, "update_divV: ld=%d smaller than nx=%d", ld, nx);
  return nsk_i_update_divV(ctx, divV, ld, Vx, Vy, dx, dy, nx, ny);
}

extern "C" int nsk_correct_V_ld(nsk_ctx *ctx, double *Vx, double *Vy, const double *Pr, int ld,
                                double dt, double rho, double dx, double dy, int nx, int ny) {
  NSK_CTX(ctx);
  NSK_NONNULL(Vx, Vy, Pr);
  NSK_GRID_OK(nx, ny);
  NSK_REQUIRE(ld >= nx, "correct_V: ld=%d smaller than nx=%d", ld, nx);
  return nsk_i_correct_V(ctx, Vx, Vy, Pr, ld, dt, rho, dx, dy, nx, ny);
}
