// Multi-pass Poisson kernel (templates).  Included by ns_poisson.hip (host logic) and by
// ns_poisson_inst.hip, which is compiled once per passes-per-launch value K.
#pragma once
#include "nsk_internal.h"

namespace nsk_poisson {

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
  double rdx_h, rdx_l, rdy_h, rdy_l;  // 1/dx, 1/dy as unevaluated sums hi + lo
};

// How x/dx and x/dy are evaluated.  All three give the correctly rounded IEEE
// quotient, i.e. the same bits as the reference's `/`.
//   DIV_IEEE  the compiler's division expansion (v_div_scale / v_rcp / Newton / v_div_fixup)
//   DIV_FMA3  q0 = RN(x*rh); r = x - q0*d (exact, FMA); q = RN(q0 + r*rh), rh = RN(1/d).
//             Markstein's theorem: q = RN(x/d) when rh is the correctly rounded
//             reciprocal and q0 is a faithful rounding of x/d; q0 is faithful for every
//             x when |1 - d*rh| <= 2^-54, which the host checks for the two divisors.
//   DIV_FMA4  as FMA3 with q0 = RN(x*rh + RN(x*rl)), rl = RN(1/d - rh): q0 is within
//             (1/2 + 2^-50) ulp of x/d for every divisor, hence faithful.
// The FMA forms assume no intermediate under/overflow: they are used only while
// 2^-900 <= |x| <= 2^900 (checked per wavefront, see SAFE_LO/SAFE_HI); otherwise
// the wavefront falls back to DIV_IEEE for that row.
enum { DIV_IEEE = 0, DIV_FMA3 = 3, DIV_FMA4 = 4 };

template <int DIV>
__device__ __forceinline__ double div_const(double x, double d, double rh, double rl) {
  if (DIV == DIV_IEEE) return x / d;
  const double q0 = DIV == DIV_FMA4 ? __builtin_fma(x, rh, x * rl) : x * rh;
  const double r = __builtin_fma(-q0, d, x);
  return __builtin_fma(r, rh, q0);
}

// Exponent window in which the FMA forms are exact: the biased exponent field of x must lie
// in [1023-900, 1023+900).  unsafe_key() maps a double to an unsigned that is < SAFE_SPAN
// exactly when it does (zero, subnormal, huge, Inf and NaN all map outside).
constexpr unsigned SAFE_LO = (1023u - 900u) << 20, SAFE_SPAN = 1800u << 20;
__device__ __forceinline__ unsigned unsafe_key(double x) {
  return ((unsigned)__double2hiint(x) & 0x7ff00000u) - SAFE_LO;
}

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

// One (strip, chunk) task, executed by one wavefront.
//
// Register state per pass s = 1..K (v = the lane's two adjacent cells), for the row r the
// pass works on in the current step:
//   P[s][ph][v]    pass s-1 pressure of row r, P[s][ph^1][v] of row r+1 (two slots that swap
//                  roles every step, so no register is ever moved)
//   E[s][v]        P_{s-1}[r] - P_{s-1}[r-1]: the lower half of the y second difference, which
//                  is the upper half of the previous step's -- row r-1 itself is not kept
//   D[s][ph][v]    pass s-1 dPrdτ of row r, D[s][ph^1][v] of row r+1
// ρ/dt*∇V of the last K+1 rows sits in an LDS ring private to the wavefront.
#ifndef NSK_PREFETCH_ROWS
#define NSK_PREFETCH_ROWS 1
#endif

template <int K, bool ALIGNED, int DIV, bool EDGE>
struct Task {
  static constexpr int HK = (K + 1) & ~1;    // column halo, even so that pairs stay aligned
  static constexpr int WOUT = 128 - 2 * HK;  // columns a strip produces
  static constexpr int RING = K + 1;
  static constexpr int PF = NSK_PREFETCH_ROWS;  // rows requested from memory ahead of use
  static constexpr int NSLOT = PF + 1;
  static constexpr int UNROLL = NSLOT % 2 == 0 ? NSLOT : 2 * NSLOT;  // lcm(2, NSLOT)

  const PoissonArgs &A;
  double2 *gring;  // [RING][64]
  int lane, c0, r0, r1, jstart, jend;
  bool indom[2], inter[2], outcol[2];
  const double *__restrict__ prs;
  const double *__restrict__ dvs;
  const double *__restrict__ dps;
  double *__restrict__ dpd;
  double *__restrict__ prd;
  size_t ld, ldd;

  double P[K + 1][2][2], E[K + 1][2], D[K + 1][2][2];
  double LP[NSLOT][2], LD[NSLOT][2], LG[NSLOT][2];  // rows in flight from memory

  __device__ __forceinline__ Task(const PoissonArgs &a, double2 *ring, int sx, int cy)
      : A(a), gring(ring) {
    lane = threadIdx.x & 63;
    c0 = sx * WOUT - HK + 2 * lane;
    r0 = cy * A.rows_per_chunk;
    r1 = min(r0 + A.rows_per_chunk, A.ny);
    jstart = max(r0 - K, 0);
    jend = min(r1 - 1 + K, A.ny - 1);
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const int c = c0 + v;
      indom[v] = c >= 0 && c < A.nx;
      inter[v] = c >= 1 && c <= A.nx - 2;
      // Column nx-1 takes its value from column nx-2 of the same pass (bc_x!), so it
      // belongs to the strip that produces column nx-2.
      const int owner_c = c == A.nx - 1 ? A.nx - 2 : c;
      outcol[v] = indom[v] && owner_c >= sx * WOUT && owner_c < (sx + 1) * WOUT;
    }
    ld = (size_t)A.ld;
    ldd = (size_t)A.ldd;
    prs = A.pr_src;
    dvs = A.divv;
    dps = A.dp_embedded ? A.dp_src : A.dp_src - ldd - 1;  // addressed by Pr cell coordinates
    dpd = A.dp_embedded ? A.dp_dst : A.dp_dst - ldd - 1;
    prd = A.pr_dst;
#pragma unroll
    for (int s = 0; s <= K; s++)
#pragma unroll
      for (int v = 0; v < 2; v++) {
        E[s][v] = 0.0;
#pragma unroll
        for (int h = 0; h < 2; h++) P[s][h][v] = D[s][h][v] = 0.0;
      }
  }

  __device__ __forceinline__ void load_row(int j, int slot) {
    const bool rowint = j >= 1 && j <= A.ny - 2;
    load_pair<ALIGNED>(prs, (size_t)j * ld + c0, indom[0], indom[1], LP[slot][0], LP[slot][1]);
    if (ALIGNED && A.dp_embedded)
      load_pair<true>(dps, (size_t)j * ldd + c0, rowint && inter[0], rowint && inter[1],
                      LD[slot][0], LD[slot][1]);
    else
      load_pair<false>(dps, (size_t)j * ldd + c0, rowint && inter[0], rowint && inter[1],
                       LD[slot][0], LD[slot][1]);
    load_pair<ALIGNED>(dvs, (size_t)j * ld + c0, rowint && indom[0], rowint && indom[1],
                       LG[slot][0], LG[slot][1]);
  }

  // One pass of one row: NavierStokes.jl:135 then :140, association as written there:
  //   dPrdτ*(1.0-damp) + dτ*(d2x/dx/dx + d2y/dy/dy - ρ/dt*∇V);  Pr + dτ*dPrdτ
  template <int QD>
  __device__ __forceinline__ void update_pair(const double d2x[2], const double d2y[2],
                                              const double dq[2], double2 g, const double cen[2],
                                              double pnew[2], double dnew[2]) const {
    const double gg[2] = {g.x, g.y};
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const double qx = div_const<QD>(div_const<QD>(d2x[v], A.dx, A.rdx_h, A.rdx_l), A.dx, A.rdx_h, A.rdx_l);
      const double qy = div_const<QD>(div_const<QD>(d2y[v], A.dy, A.rdy_h, A.rdy_l), A.dy, A.rdy_h, A.rdy_l);
      const double L = qx + qy - gg[v];
      dnew[v] = dq[v] * A.keep + A.dtau * L;
      pnew[v] = cen[v] + A.dtau * dnew[v];
    }
  }

  // One step: row j arrives, pass s advances to row j-s.  STEADY promises that every pass
  // works on a row strictly inside the task's row range, away from the y boundaries, that
  // row j+1 exists and that row j-K is one of the task's output rows; all row tests then
  // vanish and the K passes form one straight-line block the scheduler can interleave.
  template <int PH, int SL, bool STEADY>
  __device__ __forceinline__ void step(int j, int ring_j) {
    // row j is row r+1 of pass 1's window (r = j-1)
    if (STEADY || j <= jend) {
#pragma unroll
      for (int v = 0; v < 2; v++) {
        P[1][PH ^ 1][v] = LP[SL][v];
        D[1][PH ^ 1][v] = LD[SL][v];
      }
      // ρ/dt*∇V (NavierStokes.jl:135) is the same number in every pass: form it once
      gring[ring_j * 64 + lane] = make_double2(A.rho_dt * LG[SL][0], A.rho_dt * LG[SL][1]);
    }
    if (STEADY || j + PF <= jend) load_row(j + PF, (SL + PF) % NSLOT);
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = j - s;  // row this pass works on now
      if (STEADY || (r >= jstart && r <= jend)) {
        double pnew[2], dnew[2];
        const double *cen = P[s][PH], *up = P[s][PH ^ 1], *dq = D[s][PH];
        if (!STEADY && (r == jstart || r == jend)) {
          // Edge row of the task: carried through unchanged.  It is either a halo row
          // (recomputed by the neighbouring task) or a boundary row of the window.
#pragma unroll
          for (int v = 0; v < 2; v++) {
            pnew[v] = cen[v];
            dnew[v] = dq[v];
            E[s][v] = up[v] - cen[v];
          }
        } else {
          int rs = ring_j - s;
          if (rs < 0) rs += RING;
          const double2 g = gring[rs * 64 + lane];
          const double lft = lane_prev(cen[1]);  // left neighbour of cell 0
          const double rgt = lane_next(cen[0]);  // right neighbour of cell 1
          // x: (r-c)-(c-l); the middle difference serves both cells of the pair
          const double tl = cen[0] - lft, tm = cen[1] - cen[0], tr = rgt - cen[1];
          const double d2x[2] = {tm - tl, tr - tm};
          // y: (u-c)-(c-d); c-d is the previous step's u-c
          double upd[2] = {up[0] - cen[0], up[1] - cen[1]};
          double low[2] = {E[s][0], E[s][1]};
          E[s][0] = upd[0];
          E[s][1] = upd[1];
          if (!STEADY && s >= 2) {
            // Inputs are pass >= 1 values: rows 0 / ny-1 of that pass hold the y boundary
            // conditions (bc_yval!), which the carried-through edge rows do not.
            if (A.phys_bottom && r == 1) low[0] = low[1] = 0.0;  // Pr[ix,1] = Pr[ix,2]: c - d = +0
            if (A.phys_top && r == A.ny - 2) {
              upd[0] = A.topval - cen[0];
              upd[1] = A.topval - cen[1];
            }
          }
          const double d2y[2] = {upd[0] - low[0], upd[1] - low[1]};
          if (DIV == DIV_IEEE) {
            update_pair<DIV_IEEE>(d2x, d2y, dq, g, cen, pnew, dnew);
          } else {
            unsigned k0 = max(unsafe_key(d2x[0]), unsafe_key(d2y[0]));
            unsigned k1 = max(unsafe_key(d2x[1]), unsafe_key(d2y[1]));
            if (EDGE) {
              k0 = inter[0] ? k0 : 0u;
              k1 = inter[1] ? k1 : 0u;
            }
            if (__builtin_amdgcn_ballot_w64(max(k0, k1) >= SAFE_SPAN) == 0)
              update_pair<DIV>(d2x, d2y, dq, g, cen, pnew, dnew);
            else
              update_pair<DIV_IEEE>(d2x, d2y, dq, g, cen, pnew, dnew);
          }
          if (EDGE) {
#pragma unroll
            for (int v = 0; v < 2; v++) {
              if (!inter[v]) {  // boundary column or outside the domain: not updated
                pnew[v] = cen[v];
                dnew[v] = dq[v];
              }
            }
            if (r < A.xbc_rows) {  // bc_x!: Pr[1,iy]=Pr[2,iy]; Pr[end,iy]=Pr[end-1,iy]
              const double pl = lane_prev(pnew[1]);
              if (c0 == 0) pnew[0] = pnew[1];
              if (c0 + 1 == A.nx - 1) pnew[1] = pnew[0];
              if (c0 == A.nx - 1) pnew[0] = pl;
            }
          }
        }
        if (s < K) {
#pragma unroll
          for (int v = 0; v < 2; v++) {
            P[s + 1][PH ^ 1][v] = pnew[v];
            D[s + 1][PH ^ 1][v] = dnew[v];
          }
        } else if (STEADY) {
          store_pair<ALIGNED>(prd, (size_t)r * ld + c0, outcol[0], outcol[1], pnew[0], pnew[1]);
          if (ALIGNED && A.dp_embedded)
            store_pair<true>(dpd, (size_t)r * ldd + c0, outcol[0] && inter[0],
                             outcol[1] && inter[1], dnew[0], dnew[1]);
          else
            store_pair<false>(dpd, (size_t)r * ldd + c0, outcol[0] && inter[0],
                              outcol[1] && inter[1], dnew[0], dnew[1]);
        } else if (r >= r0 && r < r1) {
          const bool rowint = r >= 1 && r <= A.ny - 2;
          const bool first = r == 0, last = r == A.ny - 1;
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
  }

  __device__ __forceinline__ bool steady(int j) const {
    // passes 1..K work on rows j-1 .. j-K
    return j - K > jstart && j + PF <= jend && j - K >= r0 &&
           (!A.phys_bottom || j - K >= 2) && (!A.phys_top || j <= A.ny - 2);
  }

  template <int I, bool STEADY>
  __device__ __forceinline__ void steps(int j, int &ring_j, int jlast) {
    if constexpr (I < UNROLL) {
      if (STEADY || j + I <= jlast) step<I % 2, I % NSLOT, STEADY>(j + I, ring_j);
      ring_j = ring_j + 1 == RING ? 0 : ring_j + 1;
      steps<I + 1, STEADY>(j, ring_j, jlast);
    }
  }

  __device__ __forceinline__ void run() {
#pragma unroll
    for (int i = 0; i < PF; i++)
      if (jstart + i <= jend) load_row(jstart + i, i);
    int ring_j = 0;
    const int jlast = jend + K;
    for (int j = jstart; j <= jlast; j += UNROLL) {
      if (steady(j) && steady(j + UNROLL - 1))
        steps<0, true>(j, ring_j, jlast);
      else
        steps<0, false>(j, ring_j, jlast);
    }
  }
};

template <int K, bool ALIGNED, int DIV>
__global__ void __launch_bounds__(64) k_poisson(PoissonArgs A) {
  __shared__ double2 gring[(K + 1) * 64];
  const int task = blockIdx.x;
  const int sx = task % A.nstrips, cy = task / A.nstrips;
  constexpr int HK = (K + 1) & ~1;
  constexpr int WOUT = 128 - 2 * HK;
  // A strip is an edge strip when any of its 128 columns is not an interior column.
  const int cfirst = sx * WOUT - HK, clast = cfirst + 127;
  if (cfirst < 1 || clast > A.nx - 2) {
    Task<K, ALIGNED, DIV, true> t(A, gring, sx, cy);
    t.run();
  } else {
    Task<K, ALIGNED, DIV, false> t(A, gring, sx, cy);
    t.run();
  }
}

template <int K>
int launch_k(nsk_ctx *ctx, PoissonArgs &A, bool aligned, int div, int rows_per_chunk_hint) {
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
  const unsigned tasks = (unsigned)A.nstrips * (unsigned)A.nchunks;
#define NSK_LAUNCH(AL, DV) k_poisson<K, AL, DV><<<tasks, 64, 0, ctx->stream>>>(A)
  if (aligned) {
    if (div == DIV_FMA3) NSK_LAUNCH(true, DIV_FMA3);
    else if (div == DIV_FMA4) NSK_LAUNCH(true, DIV_FMA4);
    else NSK_LAUNCH(true, DIV_IEEE);
  } else {
    if (div == DIV_FMA3) NSK_LAUNCH(false, DIV_FMA3);
    else if (div == DIV_FMA4) NSK_LAUNCH(false, DIV_FMA4);
    else NSK_LAUNCH(false, DIV_IEEE);
  }
#undef NSK_LAUNCH
  NSK_LAUNCH_CHECK();
  return 0;
}


}  // namespace nsk_poisson
