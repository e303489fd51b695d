// Iterative pressure solvers for the bubble scripts that scale past one workgroup:
//   red/black SOR              code3_perf_gpu.jl:129-183, loop :609-617
//   pseudo-transient iteration code1_perf_gpu.jl:98-118, loop :228-242
// Both are sequences of whole-array maps, so every launch is an ordinary grid.  The stopping
// test (max|old-new| < maxError after every iteration) is evaluated on the device: iteration
// `it` leaves its maximum (bit pattern of a non-negative double, NaN sorts highest) in
// hist[it], and every kernel of a later iteration returns at once when the previous entry is
// below maxError.  The host enqueues iterations in batches and reads hist between batches, so
// there is one synchronisation per batch instead of one per iteration.
#include "nsk_internal.h"

namespace {

using dp = double *__restrict__;
using cdp = const double *__restrict__;
using u64 = unsigned long long;

#define B2(a, ld, i, j) (a)[(size_t)((j)-1) * (size_t)(ld) + (size_t)((i)-1)]
#define RT(i, j) B2(rt, ld, i, j)
#define P(i, j) B2(p, ld, i, j)

constexpr int BATCH = 8;

__device__ __forceinline__ bool already_converged(const u64 *hist, int it, double maxError) {
  return it > 0 && __longlong_as_double((long long)hist[it - 1]) < maxError;
}

// max over the wave of the bit patterns, one atomic per wave
__device__ __forceinline__ void wave_max_to(u64 *slot, u64 v) {
  for (int off = 32; off > 0; off >>= 1) {
    const u64 o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  if ((threadIdx.x & 63) == 0 && v != 0) atomicMax(slot, v);
}

// One colour of one SOR iteration: cells (i, j), i, j in 2..n+1, with (i + j) % 2 == colour.
// Thread x handles the x-th cell of that colour in row j.
__global__ void __launch_bounds__(256) k_rb_sweep(dp p, cdp tmp1, cdp tmp2, cdp rt, double beta, double dx,
                                                  double dy, int nx, int ny, int colour, u64 *hist, int it,
                                                  double maxError) {
  if (already_converged(hist, it, maxError)) return;
  const int ld = nx + 2;
  const int j = 2 + blockIdx.y;
  const int i = 2 + ((j + colour) & 1) + 2 * (int)(blockIdx.x * blockDim.x + threadIdx.x);
  u64 bits = 0;
  if (i <= nx + 1) {
    const double old = P(i, j);
    const double nw = (1.0 - beta) * old +
                      beta * B2(tmp2, ld, i, j) *
                          ((1.0 / dx) * (P(i + 1, j) / (dx * (RT(i + 1, j) + RT(i, j))) +
                                         P(i - 1, j) / (dx * (RT(i - 1, j) + RT(i, j)))) +
                           (1.0 / dy) * (P(i, j + 1) / (dy * (RT(i, j + 1) + RT(i, j))) +
                                         P(i, j - 1) / (dy * (RT(i, j - 1) + RT(i, j)))) -
                           B2(tmp1, ld, i, j));
    P(i, j) = nw;
    bits = (u64)__double_as_longlong(fabs(old - nw));
  }
  wave_max_to(hist + it, bits);
}

__global__ void __launch_bounds__(256) k_pt_dpdtau(cdp p, dp dPdtau, cdp tmp1, cdp rt, double dtau, double damp,
                                                   double dx, double dy, int nx, int ny, const u64 *hist, int it,
                                                   double maxError) {
  if (already_converged(hist, it, maxError)) return;
  const int ld = nx + 2;
  const int i = 2 + blockIdx.x * blockDim.x + threadIdx.x, j = 2 + blockIdx.y;
  if (i > nx + 1) return;
  B2(dPdtau, ld, i, j) =
      B2(dPdtau, ld, i, j) * (1.0 - damp) +
      dtau * ((1.0 / dx) * ((P(i + 1, j) - P(i, j)) / (dx * (RT(i + 1, j) + RT(i, j))) -
                            (P(i, j) - P(i - 1, j)) / (dx * (RT(i - 1, j) + RT(i, j)))) +
              (1.0 / dy) * ((P(i, j + 1) - P(i, j)) / (dy * (RT(i, j + 1) + RT(i, j))) -
                            (P(i, j) - P(i, j - 1)) / (dy * (RT(i, j - 1) + RT(i, j)))) -
              B2(tmp1, ld, i, j));
}

__global__ void __launch_bounds__(256) k_pt_update_p(dp p, cdp dPdtau, double dtau, int nx, int ny, u64 *hist,
                                                     int it, double maxError) {
  if (already_converged(hist, it, maxError)) return;
  const int ld = nx + 2;
  const int i = 2 + blockIdx.x * blockDim.x + threadIdx.x, j = 2 + blockIdx.y;
  u64 bits = 0;
  if (i <= nx + 1) {
    const double old = P(i, j);
    const double nw = old + dtau * B2(dPdtau, ld, i, j);
    P(i, j) = nw;
    bits = (u64)__double_as_longlong(fabs(old - nw));
  }
  wave_max_to(hist + it, bits);
}

int ensure_hist(nsk_ctx *ctx, int maxit) {
  if (ctx->hist_cap >= (size_t)maxit) return 0;
  if (ctx->hist_dev) NSK_HIP(hipFree(ctx->hist_dev));
  if (ctx->hist_host) NSK_HIP(hipHostFree(ctx->hist_host));
  ctx->hist_dev = ctx->hist_host = nullptr;
  ctx->hist_cap = 0;
  NSK_HIP(hipMalloc(&ctx->hist_dev, sizeof(u64) * (size_t)maxit));
  NSK_HIP(hipHostMalloc(&ctx->hist_host, sizeof(u64) * (size_t)maxit, hipHostMallocDefault));
  ctx->hist_cap = (size_t)maxit;
  return 0;
}

// Enqueue iterations in batches until the device-side history shows convergence.
template <class Enqueue>
int run_iterations(nsk_ctx *ctx, int maxit, double maxError, int *iters_out, double *change_out,
                   Enqueue enqueue) {
  if (int rc = ensure_hist(ctx, maxit)) return rc;
  u64 *hist = ctx->hist_dev;
  NSK_HIP(hipMemsetAsync(hist, 0, sizeof(u64) * (size_t)maxit, ctx->stream));
  int done = 0, iters = maxit;
  while (done < maxit) {
    const int n = maxit - done < BATCH ? maxit - done : BATCH;
    for (int k = 0; k < n; k++) enqueue(hist, done + k);
    NSK_LAUNCH_CHECK();
    NSK_HIP(hipMemcpyAsync(ctx->hist_host + done, hist + done, sizeof(u64) * (size_t)n, hipMemcpyDeviceToHost,
                           ctx->stream));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    bool stop = false;
    for (int k = 0; k < n && !stop; k++) {
      double ch;
      memcpy(&ch, ctx->hist_host + done + k, sizeof ch);
      if (ch < maxError) {
        iters = done + k + 1;
        stop = true;
      }
    }
    done += n;
    if (stop) break;
  }
  if (iters_out) *iters_out = iters;
  if (change_out) memcpy(change_out, ctx->hist_host + iters - 1, sizeof(double));
  return 0;
}

}  // namespace

extern "C" int nsk_bub_solve_pressure_rb(nsk_ctx *ctx, double *p, const double *tmp1, const double *tmp2,
                                         const double *rt, double beta, double dx, double dy, int nx, int ny,
                                         int maxit, double maxError, int *sweeps_out, double *change_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(p, tmp1, tmp2, rt);
  NSK_REQUIRE(nx >= 2 && ny >= 2, "solve_pressure_rb: grid too small: nx=%d ny=%d", nx, ny);
  NSK_REQUIRE(maxit >= 1, "solve_pressure_rb: maxit=%d", maxit);
  const dim3 block(256), grid(nsk_cdiv((unsigned)(nx + 1) / 2 + 1, 256), (unsigned)ny);
  hipStream_t st = ctx->stream;
  return run_iterations(ctx, maxit, maxError, sweeps_out, change_out, [&](u64 *hist, int it) {
    k_rb_sweep<<<grid, block, 0, st>>>(p, tmp1, tmp2, rt, beta, dx, dy, nx, ny, 0, hist, it, maxError);
    k_rb_sweep<<<grid, block, 0, st>>>(p, tmp1, tmp2, rt, beta, dx, dy, nx, ny, 1, hist, it, maxError);
  });
}

extern "C" int nsk_bub_pt_pressure(nsk_ctx *ctx, double *p, double *dPdtau, const double *tmp1,
                                   const double *rt, double dtau, double damp, double dx, double dy, int nx,
                                   int ny, int maxit, double maxError, int *iters_out, double *change_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(p, dPdtau, tmp1, rt);
  NSK_REQUIRE(nx >= 2 && ny >= 2, "pt_pressure: grid too small: nx=%d ny=%d", nx, ny);
  NSK_REQUIRE(maxit >= 1, "pt_pressure: maxit=%d", maxit);
  const dim3 block(256), grid(nsk_cdiv((unsigned)nx, 256), (unsigned)ny);
  hipStream_t st = ctx->stream;
  return run_iterations(ctx, maxit, maxError, iters_out, change_out, [&](u64 *hist, int it) {
    k_pt_dpdtau<<<grid, block, 0, st>>>(p, dPdtau, tmp1, rt, dtau, damp, dx, dy, nx, ny, hist, it, maxError);
    k_pt_update_p<<<grid, block, 0, st>>>(p, dPdtau, dtau, nx, ny, hist, it, maxError);
  });
}
