// Iterative pressure solvers for the bubble scripts that scale past one workgroup:
//   red/black SOR              code3_perf_gpu.jl:129-183, loop :609-617
//   pseudo-transient iteration code1_perf_gpu.jl:98-118, loop :228-242
// Both are sequences of whole-array maps; one iteration is one ordinary grid launch that reads p
// from one buffer and writes the other.  The stopping
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

// hist holds NSLOT partial maxima per iteration (workgroups spread their atomics over them: a
// single word would serialise one atomic per workgroup at the L2).
constexpr int NSLOT = 64;

__device__ __forceinline__ u64 wave_max(u64 v) {
  for (int off = 32; off > 0; off >>= 1) {
    const u64 o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ bool already_converged(const u64 *hist, int it, double maxError) {
  if (it == 0) return false;
  const u64 m = wave_max(hist[(size_t)(it - 1) * NSLOT + (threadIdx.x & 63)]);
  return __longlong_as_double((long long)m) < maxError;
}

// workgroup maximum of the bit patterns -> one atomic per workgroup (256 threads)
__device__ __forceinline__ void block_max_to(u64 *hist, int it, u64 v) {
  __shared__ u64 part[4];
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 m = part[0];
    for (int w = 1; w < 4; w++) m = part[w] > m ? part[w] : m;
    const unsigned b = blockIdx.x + blockIdx.y * gridDim.x;
    if (m != 0) atomicMax(hist + (size_t)it * NSLOT + (b % NSLOT), m);
  }
}

// One red/black SOR iteration per launch: red cells ((i + j) even), then black cells, for a tile
// of TX x TY cells held in LDS.  The black update of a tile needs the new red values one cell
// outside the tile, so the workgroup also updates that ring of red cells (its owner tile does the
// same and is the one that stores them); the ring's own inputs reach two cells out, hence the
// halo of 2.  Reading p_src and writing p_dst (ping-pong) keeps workgroups independent.  Every
// cell value comes from the reference's expression with the operands the colour order
// prescribes, so the result equals two whole-array colour sweeps bit for bit.
constexpr int TX = 64, TY = 16, HALO = 2, LX = TX + 2 * HALO + 1, LY = TY + 2 * HALO;

__device__ __forceinline__ double sor_value(const double (*P)[LX], const double (*R)[LX], int x, int y,
                                            double t1, double t2, double beta, double dx, double dy) {
  // x, y: LDS coordinates of the cell; (1.0-beta)*p + beta*tmp2*(...), code3_perf_gpu.jl:136-140
  return (1.0 - beta) * P[y][x] +
         beta * t2 *
             ((1.0 / dx) * (P[y][x + 1] / (dx * (R[y][x + 1] + R[y][x])) +
                            P[y][x - 1] / (dx * (R[y][x - 1] + R[y][x]))) +
              (1.0 / dy) * (P[y + 1][x] / (dy * (R[y + 1][x] + R[y][x])) +
                            P[y - 1][x] / (dy * (R[y - 1][x] + R[y][x]))) -
              t1);
}

__global__ void __launch_bounds__(256) k_rb_fused(cdp p_src, dp p_dst, cdp tmp1, cdp tmp2, cdp rt, double beta,
                                                  double dx, double dy, int nx, int ny, u64 *hist, int it,
                                                  double maxError) {
  if (already_converged(hist, it, maxError)) return;
  __shared__ double P[LY][LX], R[LY][LX];
  const int ld = nx + 2;
  // tile origin (1-based cell index of LDS coordinate (HALO, HALO)); interior cells are 2..n+1
  const int i0 = 2 + blockIdx.x * TX, j0 = 2 + blockIdx.y * TY;
  const int tid = threadIdx.x;
  for (int k = tid; k < LY * (TX + 2 * HALO); k += 256) {
    const int y = k / (TX + 2 * HALO), x = k % (TX + 2 * HALO);
    const int i = i0 - HALO + x, j = j0 - HALO + y;
    const bool in = i >= 1 && i <= nx + 2 && j >= 1 && j <= ny + 2;
    P[y][x] = in ? B2(p_src, ld, i, j) : 0.0;
    R[y][x] = in ? B2(rt, ld, i, j) : 1.0;
  }
  __syncthreads();
  u64 bits = 0;
  // red: tile plus a ring of one cell, interior cells of the domain only
  for (int k = tid; k < (TY + 2) * ((TX + 2) / 2); k += 256) {
    const int yy = k / ((TX + 2) / 2), xh = k % ((TX + 2) / 2);
    const int j = j0 - 1 + yy;
    // first red column of this row inside [i0-1, i0+TX]: (i + j) even
    const int ifirst = i0 - 1 + (((i0 - 1) + j) & 1);
    const int i = ifirst + 2 * xh;
    if (i <= i0 + TX && i >= 2 && i <= nx + 1 && j >= 2 && j <= ny + 1) {
      const int x = i - i0 + HALO, y = j - j0 + HALO;
      const double old = P[y][x];
      const double nw = sor_value(P, R, x, y, B2(tmp1, ld, i, j), B2(tmp2, ld, i, j), beta, dx, dy);
      // (the neighbours read are black, nobody writes them in this phase)
      P[y][x] = nw;
      const bool owned = i >= i0 && i < i0 + TX && j >= j0 && j < j0 + TY;
      if (owned) {
        const u64 b = (u64)__double_as_longlong(fabs(old - nw));
        bits = b > bits ? b : bits;
      }
    }
  }
  __syncthreads();
  // black: the tile's own cells
  for (int k = tid; k < TY * (TX / 2); k += 256) {
    const int yy = k / (TX / 2), xh = k % (TX / 2);
    const int j = j0 + yy;
    const int i = i0 + ((i0 + j + 1) & 1) + 2 * xh;  // (i + j) odd
    if (i <= nx + 1 && j <= ny + 1) {
      const int x = i - i0 + HALO, y = j - j0 + HALO;
      const double old = P[y][x];
      const double nw = sor_value(P, R, x, y, B2(tmp1, ld, i, j), B2(tmp2, ld, i, j), beta, dx, dy);
      P[y][x] = nw;  // (the neighbours read are red)
      const u64 b = (u64)__double_as_longlong(fabs(old - nw));
      bits = b > bits ? b : bits;
    }
  }
  __syncthreads();
  for (int k = tid; k < TY * TX; k += 256) {
    const int yy = k / TX, xx = k % TX;
    const int i = i0 + xx, j = j0 + yy;
    if (i <= nx + 1 && j <= ny + 1) B2(p_dst, ld, i, j) = P[yy + HALO][xx + HALO];
  }
  block_max_to(hist, it, bits);
}

// One pseudo-transient iteration per launch: update_dPdtau! then update_p! of the same cell
// (code1_perf_gpu.jl:98-118); p is read from p_src and written to p_dst so that neighbours still
// see the old values, which is what two whole-array launches give.
constexpr int PT_ROWS = 8;

__global__ void __launch_bounds__(256) k_pt_fused(cdp p, dp p_dst, dp dPdtau, cdp tmp1, cdp rt, double dtau,
                                                  double damp, double dx, double dy, int nx, int ny, u64 *hist,
                                                  int it, double maxError) {
  if (already_converged(hist, it, maxError)) return;
  const int ld = nx + 2;
  const int i = 2 + blockIdx.x * blockDim.x + threadIdx.x;
  u64 bits = 0;
  for (int j = 2 + blockIdx.y * PT_ROWS; j < 2 + (blockIdx.y + 1) * PT_ROWS && j <= ny + 1; j++) {
    if (i > nx + 1) break;
    const double d =
        B2(dPdtau, ld, i, j) * (1.0 - damp) +
        dtau * ((1.0 / dx) * ((P(i + 1, j) - P(i, j)) / (dx * (RT(i + 1, j) + RT(i, j))) -
                              (P(i, j) - P(i - 1, j)) / (dx * (RT(i - 1, j) + RT(i, j)))) +
                (1.0 / dy) * ((P(i, j + 1) - P(i, j)) / (dy * (RT(i, j + 1) + RT(i, j))) -
                              (P(i, j) - P(i, j - 1)) / (dy * (RT(i, j - 1) + RT(i, j)))) -
                B2(tmp1, ld, i, j));
    B2(dPdtau, ld, i, j) = d;
    const double old = P(i, j);
    const double nw = old + dtau * d;
    B2(p_dst, ld, i, j) = nw;
    const u64 b = (u64)__double_as_longlong(fabs(old - nw));
    bits = b > bits ? b : bits;
  }
  block_max_to(hist, it, bits);
}

int ensure_hist(nsk_ctx *ctx, int maxit) {
  if (ctx->hist_cap >= (size_t)maxit * NSLOT) return 0;
  if (ctx->hist_dev) NSK_HIP(hipFree(ctx->hist_dev));
  if (ctx->hist_host) NSK_HIP(hipHostFree(ctx->hist_host));
  ctx->hist_dev = ctx->hist_host = nullptr;
  ctx->hist_cap = 0;
  NSK_HIP(hipMalloc(&ctx->hist_dev, sizeof(u64) * (size_t)maxit * NSLOT));
  NSK_HIP(hipHostMalloc(&ctx->hist_host, sizeof(u64) * (size_t)maxit * NSLOT, hipHostMallocDefault));
  ctx->hist_cap = (size_t)maxit * NSLOT;
  return 0;
}

int ensure_partner(nsk_ctx *ctx, size_t elems) {
  if (ctx->iter_pp_elems >= elems) return 0;
  if (ctx->iter_pp) NSK_HIP(hipFree(ctx->iter_pp));
  ctx->iter_pp = nullptr;
  ctx->iter_pp_elems = 0;
  NSK_HIP(hipMalloc(&ctx->iter_pp, elems * sizeof(double)));
  ctx->iter_pp_elems = elems;
  return 0;
}

// Enqueue iterations in batches until the device-side history shows convergence.  Iteration i
// reads p from buf[i % 2] and writes buf[(i + 1) % 2] (buf[0] is the caller's array, buf[1] the
// context's partner, which starts as a copy so that the never-updated ghost cells agree); the
// result is brought home when it ends up in the partner.
template <class Enqueue>
int run_iterations(nsk_ctx *ctx, double *p, size_t elems, int maxit, double maxError, int *iters_out,
                   double *change_out, Enqueue enqueue) {
  if (int rc = ensure_hist(ctx, maxit)) return rc;
  if (int rc = ensure_partner(ctx, elems)) return rc;
  double *buf[2] = {p, ctx->iter_pp};
  NSK_HIP(hipMemcpyAsync(buf[1], buf[0], elems * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  u64 *hist = ctx->hist_dev;
  NSK_HIP(hipMemsetAsync(hist, 0, sizeof(u64) * (size_t)maxit * NSLOT, ctx->stream));
  int done = 0, iters = maxit;
  double last = 0.0;  // change of the last iteration looked at (= iteration `iters` when the loop ends)
  while (done < maxit) {
    const int n = maxit - done < BATCH ? maxit - done : BATCH;
    for (int k = 0; k < n; k++) enqueue(hist, done + k, buf[(done + k) & 1], buf[(done + k + 1) & 1]);
    NSK_LAUNCH_CHECK();
    NSK_HIP(hipMemcpyAsync(ctx->hist_host + (size_t)done * NSLOT, hist + (size_t)done * NSLOT,
                           sizeof(u64) * (size_t)n * NSLOT, hipMemcpyDeviceToHost, ctx->stream));
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    bool stop = false;
    for (int k = 0; k < n && !stop; k++) {
      const u64 *slots = ctx->hist_host + (size_t)(done + k) * NSLOT;
      u64 m = 0;
      for (int q = 0; q < NSLOT; q++) m = slots[q] > m ? slots[q] : m;
      double ch;
      memcpy(&ch, &m, sizeof ch);
      last = ch;
      if (ch < maxError) {
        iters = done + k + 1;
        stop = true;
      }
    }
    done += n;
    if (stop) break;
  }
  if (iters & 1)
    NSK_HIP(hipMemcpyAsync(p, buf[1], elems * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  if (iters_out) *iters_out = iters;
  if (change_out) *change_out = last;
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
  const dim3 block(256), grid(nsk_cdiv((unsigned)nx, TX), nsk_cdiv((unsigned)ny, TY));
  hipStream_t st = ctx->stream;
  return run_iterations(ctx, p, (size_t)(nx + 2) * (ny + 2), maxit, maxError, sweeps_out, change_out,
                        [&](u64 *hist, int it, const double *src, double *dst) {
                          k_rb_fused<<<grid, block, 0, st>>>(src, dst, tmp1, tmp2, rt, beta, dx, dy, nx, ny, hist,
                                                             it, maxError);
                        });
}

extern "C" int nsk_bub_pt_pressure(nsk_ctx *ctx, double *p, double *dPdtau, const double *tmp1,
                                   const double *rt, double dtau, double damp, double dx, double dy, int nx,
                                   int ny, int maxit, double maxError, int *iters_out, double *change_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(p, dPdtau, tmp1, rt);
  NSK_REQUIRE(nx >= 2 && ny >= 2, "pt_pressure: grid too small: nx=%d ny=%d", nx, ny);
  NSK_REQUIRE(maxit >= 1, "pt_pressure: maxit=%d", maxit);
  const dim3 block(256), grid(nsk_cdiv((unsigned)nx, 256), nsk_cdiv((unsigned)ny, PT_ROWS));
  hipStream_t st = ctx->stream;
  return run_iterations(ctx, p, (size_t)(nx + 2) * (ny + 2), maxit, maxError, iters_out, change_out,
                        [&](u64 *hist, int it, const double *src, double *dst) {
                          k_pt_fused<<<grid, block, 0, st>>>(src, dst, dPdtau, tmp1, rt, dtau, damp, dx, dy, nx, ny,
                                                             hist, it, maxError);
                        });
}
