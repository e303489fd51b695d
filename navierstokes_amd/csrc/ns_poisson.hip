// The hot path: the accelerated pseudo-transient pressure-Poisson iteration
//   update_dPrdτ!  scripts/NavierStokes.jl:134-137
//   update_Pr!     scripts/NavierStokes.jl:139-142
//   set_bc_Pr!     scripts/NavierStokes.jl:186-190 (bc_x! :155-159, bc_yval! :173-177)
// iterated by the loop at scripts/NavierStokes.jl:84-95.
//
// One pass of the three reference kernels is a pure map
//   (Pr, dPrdτ) -> (Pr', dPrdτ')
// with a one-cell stencil on Pr, so K consecutive passes can be evaluated from
// one read of the fields (temporal blocking): a cell of pass s needs the cells
// of pass s-1 within one cell of it.  Every cell value is produced by exactly
// the reference's expression (no FMA contraction, same association), so the
// result is bit-identical to K launches of the three kernels.
//
// Work decomposition: the grid is cut into column strips of 128 cells (64 lanes
// x 2 adjacent cells) and row chunks; ONE WAVEFRONT owns one (strip, chunk)
// task and streams through its rows bottom-to-top.  For every pass s = 1..K the
// wave keeps a three-row window of pass s-1 in registers, so the vertical
// neighbours are register reads and the horizontal neighbours of a lane's cell
// pair come from the adjacent lanes by a one-lane wavefront shift (DPP).  No
// LDS, no barriers: waves are independent.  The outer HK = K (rounded up to
// even) columns on each side of a strip and K rows at each end of a chunk are
// recomputed by the neighbouring task (halo); source and destination fields
// are distinct buffers (ping-pong), because a neighbouring task still reads
// pass-0 values that this task replaces.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "nsk_internal.h"

int nsk_launch_maxabs(nsk_ctx *ctx, const double *A, size_t count);
int nsk_read_maxabs(nsk_ctx *ctx, double *out);

namespace {

struct PoissonArgs {
  const double *pr_src;   // (ld x ny)
  const double *dp_src;   // embedded: cell (c,r) at dp[c + r*ldd]; else dp[(c-1) + (r-1)*ldd]
  const double *divv;     // (ld x ny)
  double *pr_dst;
  double *dp_dst;
  int nx, ny, ld, ldd;
  int dp_embedded;
  int nstrips, nchunks, rows_per_chunk;
  int xbc_rows;            // rows r < xbc_rows (0-based) receive the x-Neumann condition
  int phys_bottom, phys_top;
  double dx, dy, rho_dt, keep, dtau, topval;
};

// Value of x held by the previous / next lane of the wavefront (lane 0 resp.
// lane 63 receive 0).  v_mov_b32 with the DPP controls wave_shr:1 / wave_shl:1.
__device__ __forceinline__ double lane_prev(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// One cell of one pass: NavierStokes.jl:135 then :140, association as written
// there:  dPrdτ*(1.0-damp) + dτ*(d2x/dx/dx + d2y/dy/dy - ρ/dt*∇V).
__device__ __forceinline__ void cell_update(double l, double r, double dn, double up, double c,
                                            double dq, double g, const PoissonArgs &A,
                                            double &pnew, double &dnew) {
  double d2x = (r - c) - (c - l);
  double d2y = (up - c) - (c - dn);
  double L = d2x / A.dx / A.dx + d2y / A.dy / A.dy - A.rho_dt * g;
  dnew = dq * A.keep + A.dtau * L;
  pnew = c + A.dtau * dnew;
}

template <bool ALIGNED>
__device__ __forceinline__ void load_pair(const double *__restrict__ p, size_t off, bool m0,
                                          bool m1, double &v0, double &v1) {
  if (ALIGNED) {
    if (m0 || m1) {
      double2 t = *reinterpret_cast<const double2 *>(p + off);
      v0 = t.x;
      v1 = t.y;
    } else {
      v0 = 0.0;
      v1 = 0.0;
    }
  } else {
    v0 = m0 ? p[off] : 0.0;
    v1 = m1 ? p[off + 1] : 0.0;
  }
}

template <bool ALIGNED>
__device__ __forceinline__ void store_pair(double *__restrict__ p, size_t off, bool m0, bool m1,
                                           double v0, double v1) {
  if (ALIGNED) {
    if (m0 || m1) *reinterpret_cast<double2 *>(p + off) = make_double2(v0, v1);
  } else {
    if (m0) p[off] = v0;
    if (m1) p[off + 1] = v1;
  }
}

template <int K, bool ALIGNED, bool EDGE>
__device__ __forceinline__ void run_task(const PoissonArgs &A, int sx, int cy) {
  constexpr int HK = (K + 1) & ~1;       // column halo, even so that pairs stay aligned
  constexpr int WOUT = 128 - 2 * HK;     // columns a strip produces
  const int lane = threadIdx.x & 63;
  const int nx = A.nx, ny = A.ny;
  const int c0 = sx * WOUT - HK + 2 * lane;  // 0-based column of this lane's first cell
  // rows
  const int r0 = cy * A.rows_per_chunk;
  const int r1 = min(r0 + A.rows_per_chunk, ny);
  const int jstart = max(r0 - K, 0), jend = min(r1 - 1 + K, ny - 1);
  // per-cell column classes
  bool indom[2], inter[2], outcol[2];
#pragma unroll
  for (int v = 0; v < 2; v++) {
    int c = c0 + v;
    indom[v] = c >= 0 && c < nx;
    inter[v] = c >= 1 && c <= nx - 2;
    // Column nx-1 takes its value from column nx-2 of the same pass (bc_x!), so it
    // belongs to the strip that produces column nx-2.
    const int owner_c = c == nx - 1 ? nx - 2 : c;
    outcol[v] = indom[v] && owner_c >= sx * WOUT && owner_c < (sx + 1) * WOUT;
  }
  // ALIGNED pair accesses may touch the padding column nx (ld > nx when nx is odd)
  const bool ld0 = indom[0], ld1 = indom[1];
  const size_t ld = (size_t)A.ld, ldd = (size_t)A.ldd;
  const double *__restrict__ prs = A.pr_src;
  const double *__restrict__ dvs = A.divv;
  // dP addressed by Pr cell coordinates
  const double *__restrict__ dps = A.dp_embedded ? A.dp_src : A.dp_src - ldd - 1;
  double *__restrict__ dpd = A.dp_embedded ? A.dp_dst : A.dp_dst - ldd - 1;
  double *__restrict__ prd = A.pr_dst;

  double a[K + 1][2], b[K + 1][2], c[K + 1][2], dq[K + 1][2], dn[K + 1][2], g[K + 1][2];
#pragma unroll
  for (int s = 0; s <= K; s++)
#pragma unroll
    for (int v = 0; v < 2; v++) a[s][v] = b[s][v] = c[s][v] = dq[s][v] = dn[s][v] = g[s][v] = 0.0;

  double curP[2], curD[2], curG[2];
  auto load_row = [&](int j, double *P, double *D, double *G) {
    const bool rowint = j >= 1 && j <= ny - 2;
    load_pair<ALIGNED>(prs, (size_t)j * ld + c0, ld0, ld1, P[0], P[1]);
    if (ALIGNED && A.dp_embedded) {
      load_pair<true>(dps, (size_t)j * ldd + c0, rowint && inter[0], rowint && inter[1], D[0], D[1]);
    } else {
      load_pair<false>(dps, (size_t)j * ldd + c0, rowint && inter[0], rowint && inter[1], D[0], D[1]);
    }
    load_pair<ALIGNED>(dvs, (size_t)j * ld + c0, rowint && ld0, rowint && ld1, G[0], G[1]);
  };
  load_row(jstart, curP, curD, curG);

  for (int j = jstart; j <= jend + K; j++) {
    double nxtP[2] = {0.0, 0.0}, nxtD[2] = {0.0, 0.0}, nxtG[2] = {0.0, 0.0};
    if (j + 1 <= jend) load_row(j + 1, nxtP, nxtD, nxtG);
    if (j <= jend) {
#pragma unroll
      for (int v = 0; v < 2; v++) {
        c[1][v] = curP[v];
        dn[1][v] = curD[v];
        g[0][v] = curG[v];
      }
    }
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = j - s;  // row this pass works on now
      if (r >= jstart && r <= jend) {
        double pnew[2], dnew[2];
        if (r == jstart || r == jend) {
          // Edge row of the task: carried through unchanged.  It is either a
          // halo row (recomputed elsewhere) or a boundary row of the window.
#pragma unroll
          for (int v = 0; v < 2; v++) {
            pnew[v] = b[s][v];
            dnew[v] = dq[s][v];
          }
        } else {
          double lo0 = lane_prev(b[s][1]);  // left neighbour of cell 0
          double hi1 = lane_next(b[s][0]);  // right neighbour of cell 1
          double dn_[2], up_[2];
#pragma unroll
          for (int v = 0; v < 2; v++) {
            dn_[v] = a[s][v];
            up_[v] = c[s][v];
          }
          if (s >= 2) {
            // Inputs are pass >= 1 values: rows 0 / ny-1 of that pass hold the
            // y boundary conditions (bc_yval!), which the carried-through edge
            // rows do not.
            if (A.phys_bottom && r == 1) {
              dn_[0] = b[s][0];
              dn_[1] = b[s][1];
            }
            if (A.phys_top && r == ny - 2) {
              up_[0] = A.topval;
              up_[1] = A.topval;
            }
          }
          cell_update(lo0, b[s][1], dn_[0], up_[0], b[s][0], dq[s][0], g[s][0], A, pnew[0], dnew[0]);
          cell_update(b[s][0], hi1, dn_[1], up_[1], b[s][1], dq[s][1], g[s][1], A, pnew[1], dnew[1]);
          if (EDGE) {
#pragma unroll
            for (int v = 0; v < 2; v++) {
              if (!inter[v]) {  // boundary column or outside the domain: not updated
                pnew[v] = b[s][v];
                dnew[v] = dq[s][v];
              }
            }
            if (r < A.xbc_rows) {  // bc_x!: Pr[1,iy]=Pr[2,iy]; Pr[end,iy]=Pr[end-1,iy]
              double pl = lane_prev(pnew[1]);
              if (c0 == 0) pnew[0] = pnew[1];
              if (c0 + 1 == nx - 1) pnew[1] = pnew[0];
              if (c0 == nx - 1) pnew[0] = pl;
            }
          }
        }
        if (s < K) {
#pragma unroll
          for (int v = 0; v < 2; v++) {
            c[s + 1][v] = pnew[v];
            dn[s + 1][v] = dnew[v];
          }
        } else if (r >= r0 && r < r1) {
          const bool rowint = r >= 1 && r <= ny - 2;
          const bool first = r == 0, last = r == ny - 1;
          if (!(first && A.phys_bottom)) {
            double o0 = pnew[0], o1 = pnew[1];
            if (last && A.phys_top) o0 = o1 = A.topval;
            store_pair<ALIGNED>(prd, (size_t)r * ld + c0, outcol[0], outcol[1], o0, o1);
          }
          if (r == 1 && A.phys_bottom)  // bc_yval!: Pr[ix,1] = Pr[ix,2]
            store_pair<ALIGNED>(prd, (size_t)c0, outcol[0], outcol[1], pnew[0], pnew[1]);
          if (rowint) {
            if (ALIGNED && A.dp_embedded)
              store_pair<true>(dpd, (size_t)r * ldd + c0, outcol[0] && inter[0],
                               outcol[1] && inter[1], dnew[0], dnew[1]);
            else
              store_pair<false>(dpd, (size_t)r * ldd + c0, outcol[0] && inter[0],
                                outcol[1] && inter[1], dnew[0], dnew[1]);
          }
        }
      }
    }
    // advance the windows by one row
#pragma unroll
    for (int s = K; s >= 1; s--) {
#pragma unroll
      for (int v = 0; v < 2; v++) {
        a[s][v] = b[s][v];
        b[s][v] = c[s][v];
        dq[s][v] = dn[s][v];
        g[s][v] = g[s - 1][v];
      }
    }
#pragma unroll
    for (int v = 0; v < 2; v++) {
      curP[v] = nxtP[v];
      curD[v] = nxtD[v];
      curG[v] = nxtG[v];
    }
  }
}

template <int K, bool ALIGNED>
__global__ void __launch_bounds__(64) k_poisson(PoissonArgs A) {
  const int task = blockIdx.x;
  const int sx = task % A.nstrips, cy = task / A.nstrips;
  constexpr int HK = (K + 1) & ~1;
  constexpr int WOUT = 128 - 2 * HK;
  // A strip is an edge strip when any of its 128 columns is not an interior column.
  const int cfirst = sx * WOUT - HK, clast = cfirst + 127;
  if (cfirst < 1 || clast > A.nx - 2)
    run_task<K, ALIGNED, true>(A, sx, cy);
  else
    run_task<K, ALIGNED, false>(A, sx, cy);
}

template <int K>
int launch_k(nsk_ctx *ctx, PoissonArgs &A, bool aligned, int rows_per_chunk_hint) {
  constexpr int HK = (K + 1) & ~1;
  constexpr int WOUT = 128 - 2 * HK;
  A.nstrips = (A.nx - 1 + WOUT - 1) / WOUT;  // columns 0..nx-2; nx-1 rides with nx-2
  int rpc = rows_per_chunk_hint;
  if (rpc <= 0) {
    // Aim at about 2048 tasks (8 waves per CU on 256 CUs) but keep the K halo
    // rows at each chunk end a small fraction of the chunk.
    int want_chunks = 2048 / A.nstrips;
    if (want_chunks < 1) want_chunks = 1;
    rpc = (A.ny + want_chunks - 1) / want_chunks;
    if (rpc < 8 * K) rpc = 8 * K;
  }
  if (rpc < 2) rpc = 2;
  A.rows_per_chunk = rpc;
  A.nchunks = (A.ny + rpc - 1) / rpc;
  // the last chunk must hold at least 2 rows unless it is the only one (row ny-1's
  // store relies on nothing else, but row 0 is stored together with row 1)
  const unsigned tasks = (unsigned)A.nstrips * (unsigned)A.nchunks;
  if (aligned)
    k_poisson<K, true><<<tasks, 64, 0, ctx->stream>>>(A);
  else
    k_poisson<K, false><<<tasks, 64, 0, ctx->stream>>>(A);
  NSK_LAUNCH_CHECK();
  return 0;
}

int launch_poisson(nsk_ctx *ctx, PoissonArgs &A, int K, bool aligned, int rpc) {
  switch (K) {
    case 1: return launch_k<1>(ctx, A, aligned, rpc);
    case 2: return launch_k<2>(ctx, A, aligned, rpc);
    case 3: return launch_k<3>(ctx, A, aligned, rpc);
    case 4: return launch_k<4>(ctx, A, aligned, rpc);
    case 5: return launch_k<5>(ctx, A, aligned, rpc);
    case 6: return launch_k<6>(ctx, A, aligned, rpc);
    case 7: return launch_k<7>(ctx, A, aligned, rpc);
    case 8: return launch_k<8>(ctx, A, aligned, rpc);
  }
  nsk_set_error("poisson: unsupported passes per launch %d", K);
  return 2;
}

// Residual (NavierStokes.jl:145) with max|.| over the owned rows; pitched Pr/divV.
__global__ void __launch_bounds__(256) k_residual_max(double *__restrict__ Rp,
                                                      const double *__restrict__ Pr,
                                                      const double *__restrict__ divV, int ld,
                                                      double rho_dt, double dx, double dy, int nx,
                                                      int ny, int own_lo, int own_hi,
                                                      unsigned long long *__restrict__ out) {
  // thread -> interior cell (ix,iy) of dPrdτ/Rp indexing (1-based, nx-2 x ny-2)
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  const int iy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  unsigned long long m = 0;
  if (ix <= nx - 2 && iy <= ny - 2) {
    double c = A2(Pr, ld, ix + 1, iy + 1);
    double d2x = (A2(Pr, ld, ix + 2, iy + 1) - c) - (c - A2(Pr, ld, ix, iy + 1));
    double d2y = (A2(Pr, ld, ix + 1, iy + 2) - c) - (c - A2(Pr, ld, ix + 1, iy));
    double res = d2x / dx / dx + d2y / dy / dy - rho_dt * A2(divV, ld, ix + 1, iy + 1);
    if (Rp) A2(Rp, nx - 2, ix, iy) = res;
    if (iy + 1 >= own_lo && iy + 1 <= own_hi)
      m = (unsigned long long)__double_as_longlong(fabs(res));
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
    if (m) atomicMax(out, m);
  }
}

int check_params(const nsk_ns_params *p) {
  NSK_REQUIRE(p != nullptr, "null parameter block");
  NSK_REQUIRE(p->nx >= 3 && p->ny >= 3, "grid too small: nx=%d ny=%d (need >= 3)", p->nx, p->ny);
  NSK_REQUIRE(p->iy0 >= 0 && p->iy0 + p->ny <= p->nyg, "window [%d,%d] outside 1..%d", p->iy0 + 1,
              p->iy0 + p->ny, p->nyg);
  NSK_REQUIRE(p->ld_pr == 0 || p->ld_pr >= p->nx, "ld_pr=%d smaller than nx=%d", p->ld_pr, p->nx);
  NSK_REQUIRE(p->dp_embedded || p->ld_dp == 0 || p->ld_dp >= p->nx - 2,
              "ld_dp=%d smaller than nx-2=%d", p->ld_dp, p->nx - 2);
  return 0;
}

inline int ld_pr_of(const nsk_ns_params *p) { return p->ld_pr ? p->ld_pr : p->nx; }
inline int ld_dp_of(const nsk_ns_params *p) {
  if (p->dp_embedded) return ld_pr_of(p);
  return p->ld_dp ? p->ld_dp : p->nx - 2;
}

int ensure_pingpong(nsk_ctx *ctx, size_t elems) {
  if (ctx->pp_elems >= elems) return 0;
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->pp_pr) NSK_HIP(hipFree(ctx->pp_pr));
  if (ctx->pp_dp) NSK_HIP(hipFree(ctx->pp_dp));
  ctx->pp_pr = ctx->pp_dp = nullptr;
  ctx->pp_elems = 0;
  NSK_HIP(hipMalloc(&ctx->pp_pr, elems * sizeof(double)));
  NSK_HIP(hipMalloc(&ctx->pp_dp, elems * sizeof(double)));
  // columns/rows the kernels never write must not hold NaN patterns by accident
  NSK_HIP(hipMemsetAsync(ctx->pp_pr, 0, elems * sizeof(double), ctx->stream));
  NSK_HIP(hipMemsetAsync(ctx->pp_dp, 0, elems * sizeof(double), ctx->stream));
  ctx->pp_elems = elems;
  return 0;
}

}  // namespace

extern "C" int nsk_ns_poisson_iters(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV,
                                    const nsk_ns_params *p, int count) {
  NSK_CTX(ctx);
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(count >= 0, "negative pass count %d", count);
  if (count == 0) return 0;
  const int nx = p->nx, ny = p->ny, ld = ld_pr_of(p), ldd = ld_dp_of(p);
  const bool aligned = p->dp_embedded && (ld % 2 == 0) && ld >= ((nx + 1) & ~1) &&
                       ((uintptr_t)Pr % 16 == 0) &&
                       ((uintptr_t)dPrdtau % 16 == 0) && ((uintptr_t)divV % 16 == 0);
  // ping-pong partners use the layout of the caller's arrays
  const size_t pr_elems = (size_t)ld * ny;
  const size_t dp_elems = p->dp_embedded ? pr_elems : (size_t)ldd * (ny - 2);
  if (int rc = ensure_pingpong(ctx, pr_elems > dp_elems ? pr_elems : dp_elems)) return rc;

  PoissonArgs A;
  memset(&A, 0, sizeof A);
  A.divv = divV;
  A.nx = nx;
  A.ny = ny;
  A.ld = ld;
  A.ldd = ldd;
  A.dp_embedded = p->dp_embedded;
  {
    int lim = p->full_rows ? p->nyg : (nx < p->nyg ? nx : p->nyg);
    int rows = lim - p->iy0;
    A.xbc_rows = rows < 0 ? 0 : (rows > ny ? ny : rows);
  }
  A.phys_bottom = p->iy0 == 0;
  A.phys_top = p->iy0 + ny == p->nyg;
  A.dx = p->dx;
  A.dy = p->dy;
  A.rho_dt = p->rho / p->dt;
  A.keep = 1.0 - p->damp;
  A.dtau = p->dtau;
  A.topval = 0.0;  // set_bc_Pr!(Pr, 0.0), NavierStokes.jl:87

  int kmax = ctx->poisson_k > 0 ? ctx->poisson_k : 4;
  if (kmax > 8) kmax = 8;
  const char *rpc_env = getenv("NSK_POISSON_ROWS");
  const int rpc = rpc_env ? atoi(rpc_env) : 0;

  // Split `count` into launches of at most kmax passes; make the number of
  // launches even when possible so the result lands in the caller's arrays.
  int nl = (count + kmax - 1) / kmax;
  if ((nl & 1) && count >= nl + 1) nl++;
  double *src_pr = Pr, *src_dp = dPrdtau, *dst_pr = ctx->pp_pr, *dst_dp = ctx->pp_dp;
  int left = count;
  for (int i = 0; i < nl; i++) {
    int k = (left + (nl - i) - 1) / (nl - i);
    A.pr_src = src_pr;
    A.dp_src = src_dp;
    A.pr_dst = dst_pr;
    A.dp_dst = dst_dp;
    if (int rc = launch_poisson(ctx, A, k, aligned, rpc)) return rc;
    left -= k;
    double *t = src_pr; src_pr = dst_pr; dst_pr = t;
    t = src_dp; src_dp = dst_dp; dst_dp = t;
  }
  if (src_pr != Pr) {  // odd number of launches: bring the result home
    NSK_HIP(hipMemcpyAsync(Pr, src_pr, pr_elems * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    NSK_HIP(hipMemcpyAsync(dPrdtau, src_dp, dp_elems * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  return 0;
}

extern "C" int nsk_ns_residual(nsk_ctx *ctx, double *Rp, const double *Pr, const double *divV,
                               const nsk_ns_params *p, int own_lo, int own_hi, double *maxabs_out) {
  NSK_CTX(ctx);
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(maxabs_out != nullptr, "nsk_ns_residual: maxabs_out is null");
  const int nx = p->nx, ny = p->ny;
  NSK_HIP(hipMemsetAsync(ctx->red_dev, 0, sizeof(unsigned long long), ctx->stream));
  dim3 grid(nsk_cdiv(nx - 2, 64), nsk_cdiv(ny - 2, 4));
  k_residual_max<<<grid, 256, 0, ctx->stream>>>(Rp, Pr, divV, ld_pr_of(p), p->rho / p->dt, p->dx,
                                               p->dy, nx, ny, own_lo, own_hi, ctx->red_dev);
  NSK_LAUNCH_CHECK();
  return nsk_read_maxabs(ctx, maxabs_out);
}

extern "C" int nsk_ns_poisson(nsk_ctx *ctx, double *Pr, double *dPrdtau, double *Rp,
                              const double *divV, const nsk_ns_params *p, int *iters_out,
                              double *err_hist, int *iter_hist, int hist_cap, int *nhist_out) {
  NSK_CTX(ctx);
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(p->iy0 == 0 && p->ny == p->nyg, "nsk_ns_poisson works on a whole domain");
  const int niter = p->niter, nchk = p->nchk;
  int done = 0, nh = 0;
  while (done < niter) {
    int chunk = (nchk > 0) ? nchk - (done % nchk) : niter - done;
    if (chunk > niter - done) chunk = niter - done;
    if (int rc = nsk_ns_poisson_iters(ctx, Pr, dPrdtau, divV, p, chunk)) return rc;
    done += chunk;
    if (nchk > 0 && done % nchk == 0) {
      double m = 0.0;
      if (int rc = nsk_ns_residual(ctx, Rp, Pr, divV, p, 2, p->ny - 1, &m)) return rc;
      double err = m * p->ly * p->ly / p->psc;  // NavierStokes.jl:90
      if (nh < hist_cap) {
        if (err_hist) err_hist[nh] = err;
        if (iter_hist) iter_hist[nh] = done;
      }
      nh++;
      if (err < p->eps_it || !std::isfinite(err)) break;  // :93
    }
  }
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  if (iters_out) *iters_out = done;
  if (nhist_out) *nhist_out = nh;
  return 0;
}
