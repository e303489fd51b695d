// Multi-pass Poisson kernel (templates).  Included by ns_poisson.hip (host logic) and by
// ns_poisson_inst.hip, which is compiled once per passes-per-launch value K.
#pragma once
#include <algorithm>
#include <utility>
#include <vector>

#include "nsk_internal.h"

// Build-time shape of the staged input path: rows requested ahead of the row in use, and
// wavefronts resident per SIMD (bounds the register budget: 512 / NSK_WAVES per lane).
#ifndef NSK_PFD
#define NSK_PFD 3
#endif
#ifndef NSK_WAVES
#define NSK_WAVES 2
#endif
#define NSK_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(NSK_WAVES, NSK_WAVES)))

namespace nsk_poisson {
#ifdef NSK_POISSON_TRACE
// Debug build only: [0] tasks run, [1] of them run again with IEEE division.
static __device__ unsigned long long nsk_trace_cnt[4];
#endif

struct PoissonArgs {
  const double *pr_src;   // (ld x ny)
  const double *dp_src;   // embedded: cell (c,r) at dp[c + r*ldd]; else dp[(c-1) + (r-1)*ldd]
  const double *divv;     // (ld x ny)
  double *pr_dst;
  double *dp_dst;
  int nx, ny, ld, ldd;
  int dp_embedded;
  // Rows this launch produces: chunks are cut from [row_begin, row_end) and, after those, from a
  // second range [row2_begin, row2_end) (empty unless the launch serves both edges of a window).
  int row_begin, row_end, row2_begin, row2_end;
  int nchunks1, nchunks1_edge;            // chunks of the first range (interior strips / edge strips)
  int nstrips, nchunks, rows_per_chunk;   // interior strips: chunks and rows per chunk
  int nchunks_edge, rows_per_chunk_edge;  // the first and the last strip do more work per row
  int n_interior_tasks;
  // Unequal task heights (0 = all chunks alike): the chunks of the interior strips come in pairs of a tall
  // one (rows_per_chunk + tall_extra rows) and, above it, a short one (rows_per_chunk - tall_extra); the
  // tall chunks take the first half of the workgroup numbers, see k_poisson.
  int tall_extra, pair_tasks, pair_tasks_pad;
  int xcd_bands;  // nonzero: XCD-aware task numbering
  int xbc_rows;            // rows r < xbc_rows (0-based) receive the x-Neumann condition
  int phys_bottom, phys_top;
  double dx, dy, rho_dt, keep, dtau, topval;
  double rdx_h, rdx_l, rdy_h, rdy_l;  // 1/dx, 1/dy as unevaluated sums hi + lo
};

// How x/dx and x/dy are evaluated.  All forms give the correctly rounded IEEE
// quotient, i.e. the same bits as the reference's `/`.
//   DIV_IEEE  the compiler's division expansion (v_div_scale / v_rcp / Newton / v_div_fixup)
//   DIV_FMA2  q = RN(x*rh + RN(x*rl)) with rh + rl = 1/d to 106 bits.  The argument of the final
//             rounding is x/d (1 + e), |e| < 2^-104, so q = RN(x/d) unless x/d lies within 2^-104
//             (relative) of a midpoint between two doubles.  For a given divisor only a handful of
//             significands x come that close (solutions of X 2^s - O D = N for small N); the host
//             enumerates them for a 16 times wider window and evaluates the form on each
//             (reciprocal_for in ns_poisson.hip).  Used only for divisors that pass.
//   DIV_FMA3  q0 = RN(x*rh); r = x - q0*d (exact, FMA); q = RN(q0 + r*rh), rh = RN(1/d).
//             Markstein's theorem: q = RN(x/d) when rh is the correctly rounded
//             reciprocal and q0 is a faithful rounding of x/d; q0 is faithful for every
//             x when |1 - d*rh| <= 2^-54, which the host checks for the two divisors.
//   DIV_FMA4  as FMA3 with q0 = RN(x*rh + RN(x*rl)), rl = RN(1/d - rh): q0 is within
//             (1/2 + 2^-50) ulp of x/d for every divisor, hence faithful.
// The FMA forms assume no intermediate underflow, i.e. every dividend is 0 or at least 2^-900 in
// magnitude, and finite operands (see note_values / E_MIN_OK).  A task that cannot guarantee this
// is run again from its inputs with DIV_IEEE (Task::run_task).
enum { DIV_IEEE = 0, DIV_FMA2 = 2, DIV_FMA3 = 3, DIV_FMA4 = 4 };

template <int DIV>
__device__ __forceinline__ double div_const(double x, double d, double rh, double rl) {
  if (DIV == DIV_IEEE) return x / d;
  if (DIV == DIV_FMA2) return __builtin_fma(x, rh, x * rl);
  const double q0 = DIV == DIV_FMA4 ? __builtin_fma(x, rh, x * rl) : x * rh;
  const double r = __builtin_fma(-q0, d, x);
  return __builtin_fma(r, rh, q0);
}

// When are the FMA forms safe?  Every dividend is a second difference (c-b)-(b-a) of pressure
// values.  If every NONZERO pressure value that enters a stencil has magnitude >= 2^m, all three
// are multiples of 2^(m-52), and so are the rounded differences: a dividend is then exactly 0
// (harmless: every form returns 0) or at least 2^(m-52).  With m = -848 the dividends of the first
// division are 0 or >= 2^-900 and, for divisors up to 2^30, those of the second >= 2^-930, where
// all remainders and correction terms are still normal numbers.  So a task only tracks the smallest
// frexp exponent of the pressure values it consumes (inputs and the intermediate passes), with 0
// neutral: v_frexp_exp_i32_f64 returns 0 for zero, Inf and NaN and the true exponent for
// subnormals.  |x| >= 2^-848  <=>  frexp exponent >= -847.
// The upper side needs no tracking: huge finite operands are handled exactly (nothing overflows
// before the quotient itself does, and then the quotient is the same infinity), and an Inf or NaN
// operand makes every value that depends on it Inf or NaN -- in either form of division -- so a
// task whose stored output contains a non-finite value is run again with IEEE division.
constexpr int E_MIN_OK = -847;
__device__ __forceinline__ bool non_finite(double x) { return __builtin_amdgcn_class(x, 0x207); }

// Value of x held by the previous / next lane of the wavefront (lane 0 resp.
// lane 63 receive 0).  v_mov_b32 with the DPP controls wave_shr:1 / wave_shl:1.
__device__ __forceinline__ double lane_prev(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <bool ALIGNED>
__device__ __forceinline__ void load_pair(const double *__restrict__ p, ptrdiff_t off, bool m0,
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
__device__ __forceinline__ void store_pair(double *__restrict__ p, ptrdiff_t off, bool m0, bool m1,
                                           double v0, double v1) {
  if (ALIGNED) {
    if (m0 || m1) *reinterpret_cast<double2 *>(p + off) = make_double2(v0, v1);
  } else {
    if (m0) p[off] = v0;
    if (m1) p[off + 1] = v1;
  }
}

// ---- LDS helpers for the staged (LDS-DMA) input path ---------------------------------------
// The staging rings are written by global_load_lds (asynchronous, counted in vmcnt) and read by
// ds_read_b128 issued from inline asm, with the s_waitcnt placed by hand: if the compiler saw
// ordinary LDS reads of a ring that LDS-DMA writes it would insert s_waitcnt vmcnt(0) before
// each of them, which would drain the whole prefetch queue every step.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;
typedef int v4i_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i_t lds_read16_nowait(unsigned byte_addr) {
  v4i_t v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_write16(unsigned byte_addr, v4i_t v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(byte_addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// Re-defines x behind the wait so that no use of it can be scheduled ahead of the wait.
__device__ __forceinline__ void pin_after_wait(v4i_t &x) { asm volatile("" : "+v"(x)); }
template <int N>
__device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ double2 as_double2(v4i_t v) {
  return make_double2(__hiloint2double(v.y, v.x), __hiloint2double(v.w, v.z));
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
//
// Input rows reach the wave in one of two ways:
//   DMA (the vectorised storage): global_load_lds_dwordx4 streams the rows of Pr, dPrdτ and ∇V
//       PFD rows ahead of their use into per-wave LDS rings, costing no registers while in
//       flight; ∇V rows stay in their ring for K more steps, where the K passes read them.
//   registers (any other storage): 8-byte loads one row ahead into registers, ρ/dt*∇V of the
//       last K+1 rows in a small LDS ring.
#ifndef NSK_PREFETCH_ROWS
#define NSK_PREFETCH_ROWS 1
#endif

template <int K>
struct DmaGeometry {
  static constexpr int PFD = NSK_PFD;                 // rows requested ahead of the row in use
  static constexpr int NST = PFD + 1;             // ring slots of Pr and of dPrdτ
  static constexpr int NSG = PFD + K + 1;         // ring slots of ∇V (rows j-K .. j+PFD)
  static constexpr int LDS_BYTES = (2 * NST + NSG) * 1024;
};

// PAIR: the task is one of two that start at a common row and walk away from it, one up and one down the
// array (the second with negative row strides: in its own row numbers it walks up as well).  Neither needs
// halo rows at the common end: where a pass first works on the task's first row, the row "below" it is the
// partner's first row of the same pass, handed over through LDS (one exchange per pass, in the first K
// steps), instead of K rows that both would recompute.
template <int K, bool ALIGNED, int DIV, bool EDGE, bool STREAM = false, bool PAIR = false>
struct Task {
  static constexpr int HK = (K + 1) & ~1;    // column halo, even so that pairs stay aligned
  static constexpr int WOUT = 128 - 2 * HK;  // columns a strip produces
  static constexpr bool DMA = ALIGNED;
  static constexpr int RING = K + 1;
  static constexpr int PF = NSK_PREFETCH_ROWS;  // register path: rows requested ahead of use
  static constexpr int NSLOT = PF + 1;
  static constexpr int UNROLL = DMA ? 2 : (NSLOT % 2 == 0 ? NSLOT : 2 * NSLOT);  // lcm(2, NSLOT)
  static constexpr int PFD = DmaGeometry<K>::PFD, NST = DmaGeometry<K>::NST, NSG = DmaGeometry<K>::NSG;

  const PoissonArgs &A;
  double2 *lds;  // register path: ∇V ring [RING][64]; DMA path: the three staging rings
  int lane, c0, r0, r1, jstart, jend;
  int jtop;  // PAIR: last row that arrives in a steady step (below the window's top rows, if the task reaches them)
  bool indom[2], inter[2], outcol[2];
  const double *__restrict__ prs;
  const double *__restrict__ dvs;
  const double *__restrict__ dps;
  double *__restrict__ dpd;
  double *__restrict__ prd;
  ptrdiff_t ld, ldd;  // row strides, negative for a task that walks down the array
  int nyt;            // rows of the window in the task's row numbers (no limit for a task that walks down)
  int xb_lo;          // PAIR: bc_x! reaches task rows >= xb_lo (a task that walks down) or < A.xbc_rows (xb_lo < 0)
  bool pb, pt;        // row 0 / row nyt-1 of the task's numbering is the window's physical bottom / top
  unsigned xch;       // PAIR: LDS byte address of the exchange area, and which of the two this wavefront is
  int wave;
  size_t cc;          // DMA path: this lane's column, clamped into the row
  unsigned lds_lane;  // DMA path: LDS byte address of this lane's 16 bytes in slot 0

  double P[K + 1][2][2], E[K + 1][2], D[K + 1][2][2];
  int emin;       // smallest frexp exponent of the nonzero pressure values consumed so far
  bool bad_out;   // a stored output was Inf or NaN
  double LP[NSLOT][2], LD[NSLOT][2], LG[NSLOT][2];  // register path: rows in flight

  // Rows [row0, row1) of the window.  mirror >= 0 (= row1 - 1): the task walks DOWN from window row `mirror`
  // to row0; in its own numbering window row r is row 2*mirror + 1 - r, so its first row has the number
  // mirror + 1 -- the number of the first row of its partner, which walks up from there.
  __device__ __forceinline__ Task(const PoissonArgs &a, double2 *shared, int sx, int row0, int row1, int mirror = -1,
                                  unsigned exchange = 0, int which = 0)
      : A(a), lds(shared) {
    lane = threadIdx.x & 63;
    c0 = sx * WOUT - HK + 2 * lane;
    xch = exchange;
    wave = which;
    if (mirror >= 0) {  // window row r <-> task row 2*mirror + 1 - r
      r0 = mirror + 1;
      r1 = 2 * mirror + 2 - row0;
      nyt = 1 << 30;
      pb = pt = false;
      xb_lo = max(2 * mirror + 2 - A.xbc_rows, 0);  // window row 2*mirror + 1 - r < xbc_rows
    } else {
      r0 = row0;
      r1 = row1;
      nyt = A.ny;
      pb = A.phys_bottom != 0;
      pt = A.phys_top != 0;
      xb_lo = -1;
    }
    jstart = PAIR ? r0 : max(r0 - K, 0);
    jend = min(r1 - 1 + K, nyt - 1);
    jtop = pt ? nyt - 2 : 0x7fffffff;
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
    ld = (ptrdiff_t)A.ld;
    ldd = (ptrdiff_t)A.ldd;
    prs = A.pr_src;
    dvs = A.divv;
    dps = A.dp_embedded ? A.dp_src : A.dp_src - ldd - 1;  // addressed by Pr cell coordinates
    dpd = A.dp_embedded ? A.dp_dst : A.dp_dst - ldd - 1;
    prd = A.pr_dst;
    if (mirror >= 0) {  // task row j is window row 2*mirror + 1 - j: bases at that row 0, strides negated
      const ptrdiff_t top = 2 * (ptrdiff_t)mirror + 1;
      prs += top * ld, dvs += top * ld, prd += top * ld;
      dps += top * ldd, dpd += top * ldd;
      ld = -ld, ldd = -ldd;
    }
    // Lanes outside the domain fetch some pair inside the row: their values are never used
    // (such cells are carried through and not stored), the clamp only keeps the access legal.
    cc = (size_t)min(max(c0, 0), A.ld - 2);
    lds_lane = (unsigned)(uintptr_t)(lds_void_t *)shared + 16u * (unsigned)lane;
    reset();
  }

  // Pipeline state of a task that has not started.
  __device__ __forceinline__ void reset() {
    emin = 0;
    bad_out = false;
#pragma unroll
    for (int s = 0; s <= K; s++)
#pragma unroll
      for (int v = 0; v < 2; v++) {
        E[s][v] = 0.0;
#pragma unroll
        for (int h = 0; h < 2; h++) P[s][h][v] = D[s][h][v] = 0.0;
      }
  }

  // ---- register path ---------------------------------------------------------------------
  __device__ __forceinline__ void load_row(int j, int slot) {
    const bool rowint = j >= 1 && j <= nyt - 2;
    load_pair<ALIGNED>(prs, (ptrdiff_t)j * ld + c0, indom[0], indom[1], LP[slot][0], LP[slot][1]);
    load_pair<false>(dps, (ptrdiff_t)j * ldd + c0, rowint && inter[0], rowint && inter[1], LD[slot][0],
                     LD[slot][1]);
    load_pair<ALIGNED>(dvs, (ptrdiff_t)j * ld + c0, rowint && indom[0], rowint && indom[1],
                       LG[slot][0], LG[slot][1]);
  }

  // ---- DMA path ----------------------------------------------------------------------------
  // Request row j: 1 KiB of each field straight into LDS, slot sp of the Pr and dPrdτ rings and
  // slot sg of the ∇V ring (64 lanes x 16 bytes, lane-major).
  // STREAM: Pr and dPrdτ are requested with the non-temporal hint (cache-policy bit `nt` of the load;
  // ∇V, which every launch reads again, without).  Values do not depend on it; when it pays is the
  // host's decision (stream_inputs).
  __device__ __forceinline__ void dma_row(int j, int sp, int sg) {
    const ptrdiff_t off = (ptrdiff_t)j * ld + (ptrdiff_t)cc;
    char *base = reinterpret_cast<char *>(lds);
    // (the rows around the far end are read by the task beyond it as well, at about the same time -- its halo
    //  rows and ours: those without the hint, so that the second reader finds them in L2: 7 % fewer bytes
    //  from the fabric at 2559 x 4095, the same time)
    if (STREAM && j < r1 - 2 * K) {
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(prs + off), (lds_void_t *)(base + sp * 1024), 16, 0, 2);
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(dps + off), (lds_void_t *)(base + (NST + sp) * 1024), 16, 0, 2);
    } else {
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(prs + off), (lds_void_t *)(base + sp * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(dps + off), (lds_void_t *)(base + (NST + sp) * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(dvs + off), (lds_void_t *)(base + (2 * NST + sg) * 1024), 16, 0, 0);
  }

  // Row j of Pr and dPrdτ (slot sp) and ρ/dt*∇V of rows j-1 .. j-K (slots behind sg).
  // The ∇V row that has just arrived (row j, slot sg) is replaced in its slot by ρ/dt*∇V
  // (NavierStokes.jl:135), so the K passes that use it read the product instead of forming it K
  // times; LDS operations of a wave execute in order, so the reads of the following steps see it.
  __device__ __forceinline__ void dma_fetch(int sp, int sg, bool want_row, double inP[2], double inD[2],
                                            double2 g[K + 1]) {
    v4i_t vp = {0, 0, 0, 0}, vd = {0, 0, 0, 0}, vg[K + 1], vraw = {0, 0, 0, 0};
    if (want_row) {
      vp = lds_read16_nowait(lds_lane + (unsigned)sp * 1024u);
      vd = lds_read16_nowait(lds_lane + (unsigned)(NST + sp) * 1024u);
      vraw = lds_read16_nowait(lds_lane + (unsigned)(2 * NST + sg) * 1024u);
    }
#pragma unroll
    for (int s = 1; s <= K; s++) {
      int q = sg - s;
      if (q < 0) q += NSG;
      vg[s] = lds_read16_nowait(lds_lane + (unsigned)(2 * NST + q) * 1024u);
    }
    lds_wait_all();
    pin_after_wait(vp);
    pin_after_wait(vd);
    pin_after_wait(vraw);
#pragma unroll
    for (int s = 1; s <= K; s++) pin_after_wait(vg[s]);
    if (want_row) {
      const double2 r = as_double2(vraw);
      const double g0 = A.rho_dt * r.x, g1 = A.rho_dt * r.y;
      const v4i_t w = {__double2loint(g0), __double2hiint(g0), __double2loint(g1), __double2hiint(g1)};
      lds_write16(lds_lane + (unsigned)(2 * NST + sg) * 1024u, w);
    }
    const double2 p = as_double2(vp), d = as_double2(vd);
    inP[0] = p.x;
    inP[1] = p.y;
    inD[0] = d.x;
    inD[1] = d.y;
#pragma unroll
    for (int s = 1; s <= K; s++) g[s] = as_double2(vg[s]);
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

  // Columns 0 and nx-1 are not updated by the iteration itself.  In rows that bc_x! reaches
  // they take the new value of their interior neighbour (Pr[1,iy]=Pr[2,iy];
  // Pr[end,iy]=Pr[end-1,iy]); in the other rows they keep their value.  Cells outside the
  // domain hold don't-care values: nothing reads them and they are never stored.
  __device__ __forceinline__ void edge_fixups(int r, const double cen[2], const double dq[2],
                                              double pnew[2], double dnew[2]) const {
    if (PAIR && xb_lo >= 0 ? r >= xb_lo : r < A.xbc_rows) {
      const double pl = lane_prev(pnew[1]);
      if (c0 == 0) pnew[0] = pnew[1];
      if (c0 + 1 == A.nx - 1) pnew[1] = pnew[0];
      if (c0 == A.nx - 1) pnew[0] = pl;
    } else {
      if (c0 == 0 || c0 == A.nx - 1) pnew[0] = cen[0];
      if (c0 + 1 == A.nx - 1) pnew[1] = cen[1];
    }
  }

  __device__ __forceinline__ void store_dp(int r, const double dnew[2]) {
    if (ALIGNED)
      store_pair<true>(dpd, (ptrdiff_t)r * ldd + c0, outcol[0] && inter[0], outcol[1] && inter[1],
                       dnew[0], dnew[1]);
    else
      store_pair<false>(dpd, (ptrdiff_t)r * ldd + c0, outcol[0] && inter[0], outcol[1] && inter[1],
                        dnew[0], dnew[1]);
  }

  // General step: row j (inP, inD; ignored when j > jend) arrives, pass s advances to row j-s
  // if that row is inside the task's range; every row test is evaluated.
  template <int PH, int QD>
  __device__ __forceinline__ void general_step(int j, const double inP[2], const double inD[2],
                                               const double2 g[K + 1]) {
    if (j <= jend) {
#pragma unroll
      for (int v = 0; v < 2; v++) {
        P[1][PH ^ 1][v] = inP[v];
        D[1][PH ^ 1][v] = inD[v];
      }
      note_values<QD>(inP);
    }
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = j - s;  // row this pass works on now
      if (r >= jstart && r <= jend) {
        double pnew[2], dnew[2];
        const double *cen = P[s][PH], *up = P[s][PH ^ 1], *dq = D[s][PH];
        if ((r == jstart && !PAIR) || r == jend) {
          // Edge row of the task: carried through unchanged.  It is either a halo row
          // (recomputed by the neighbouring task) or a boundary row of the window.
#pragma unroll
          for (int v = 0; v < 2; v++) {
            pnew[v] = cen[v];
            dnew[v] = dq[v];
            E[s][v] = up[v] - cen[v];
          }
        } else {
          const double lft = lane_prev(cen[1]);  // left neighbour of cell 0
          const double rgt = lane_next(cen[0]);  // right neighbour of cell 1
          // x: (r-c)-(c-l); the middle difference serves both cells of the pair
          const double tl = cen[0] - lft, tm = cen[1] - cen[0], tr = rgt - cen[1];
          const double d2x[2] = {tm - tl, tr - tm};
          // y: (u-c)-(c-d); c-d is the previous step's u-c
          double upd[2] = {up[0] - cen[0], up[1] - cen[1]};
          double low[2] = {E[s][0], E[s][1]};
          if (PAIR && r == jstart) {
            // The row below the task's first row is the partner's first row, at the same pass and in the same
            // step: both hand theirs over.  (LDS write, wait for it, barrier of the two wavefronts, read; the
            // slots alternate with the pass, so one barrier per exchange is enough.)
            const unsigned slot = xch + (unsigned)(s & 1) * 2048u + 16u * (unsigned)lane;
            const v4i_t mine = {__double2loint(cen[0]), __double2hiint(cen[0]), __double2loint(cen[1]), __double2hiint(cen[1])};
            lds_write16(slot + (unsigned)wave * 1024u, mine);
            lds_wait_all();
            __builtin_amdgcn_s_barrier();
            v4i_t theirs = lds_read16_nowait(slot + (unsigned)(wave ^ 1) * 1024u);
            lds_wait_all();
            pin_after_wait(theirs);
            const double2 pc = as_double2(theirs);
            const double below[2] = {pc.x, pc.y};
            note_values<QD>(below);
            low[0] = cen[0] - below[0];
            low[1] = cen[1] - below[1];
          }
          E[s][0] = upd[0];
          E[s][1] = upd[1];
          if (s >= 2) {
            // Inputs are pass >= 1 values: rows 0 / ny-1 of that pass hold the y boundary
            // conditions (bc_yval!), which the carried-through edge rows do not.
            if (pb && r == 1) low[0] = low[1] = 0.0;  // Pr[ix,1] = Pr[ix,2]: c - d = +0
            if (pt && r == nyt - 2) {
              upd[0] = A.topval - cen[0];
              upd[1] = A.topval - cen[1];
            }
          }
          const double d2y[2] = {upd[0] - low[0], upd[1] - low[1]};
          update_pair<QD>(d2x, d2y, dq, g[s], cen, pnew, dnew);
          if (EDGE) edge_fixups(r, cen, dq, pnew, dnew);
          if (s < K) note_values<QD>(pnew);
        }
        if (s < K) {
#pragma unroll
          for (int v = 0; v < 2; v++) {
            P[s + 1][PH ^ 1][v] = pnew[v];
            D[s + 1][PH ^ 1][v] = dnew[v];
          }
        } else if (r >= r0 && r < r1) {
          note_stored<QD>(pnew);
          const bool rowint = r >= 1 && r <= nyt - 2;
          const bool first = r == 0, last = r == nyt - 1;
          if (!(first && pb)) {
            double o0 = pnew[0], o1 = pnew[1];
            if (last && pt) o0 = o1 = A.topval;
            store_pair<ALIGNED>(prd, (ptrdiff_t)r * ld + c0, outcol[0], outcol[1], o0, o1);
          }
          if (r == 1 && pb)  // bc_yval!: Pr[ix,1] = Pr[ix,2]
            store_pair<ALIGNED>(prd, (ptrdiff_t)c0, outcol[0], outcol[1], pnew[0], pnew[1]);
          if (rowint) store_dp(r, dnew);
        }
      }
    }
  }

  // The FMA forms of the division need pressure values that are 0 or >= 2^-848 (see E_MIN_OK).
  // A task runs optimistically: it keeps the smallest exponent of the pressure values it has
  // consumed and whether a stored value was non-finite, the driver looks at both between steps,
  // and a task that cannot vouch for its quotients is run again from its inputs with IEEE
  // division (the inputs are intact: a launch reads one buffer pair and writes the other).  In
  // whole time steps from the reference's initial condition this happens only in the first step,
  // along the front where the pressure arrives through subnormal values.
  template <int QD>
  __device__ __forceinline__ void note_values(const double x[2]) {
    if (QD != DIV_IEEE) {
      int e0 = __builtin_amdgcn_frexp_exp(x[0]), e1 = __builtin_amdgcn_frexp_exp(x[1]);
      if (EDGE) {  // cells outside the domain hold don't-care values
        e0 = indom[0] ? e0 : 0;
        e1 = indom[1] ? e1 : 0;
      }
      emin = min(emin, min(e0, e1));
    }
  }
  template <int QD>
  __device__ __forceinline__ void note_stored(const double x[2]) {
    if (QD != DIV_IEEE) bad_out |= (outcol[0] && non_finite(x[0])) || (outcol[1] && non_finite(x[1]));
  }
  __device__ __forceinline__ bool left_safe_window() const {
    return __builtin_amdgcn_ballot_w64(emin < E_MIN_OK || bad_out) != 0;
  }

  // Steady step: every pass works on a row strictly inside the task's row range, away from the
  // y boundaries, and row j-K is one of the task's output rows, so no row test is needed and the
  // K passes form one basic block.  Pass s reads P[s][PH] (row r), E[s], D[s][PH] and, as the
  // row above, what pass s-1 has just written to P[s][PH^1] (the newly arrived row for s = 1).
  template <int PH, int QD>
  __device__ __forceinline__ void steady_step(int j, const double inP[2], const double inD[2],
                                              const double2 g[K + 1]) {
#pragma unroll
    for (int v = 0; v < 2; v++) {
      P[1][PH ^ 1][v] = inP[v];
      D[1][PH ^ 1][v] = inD[v];
    }
    note_values<QD>(inP);
#pragma unroll
    for (int s = 1; s <= K; s++) {
      const int r = j - s;
      const double *cen = P[s][PH], *up = P[s][PH ^ 1], *dq = D[s][PH];
      const double lft = lane_prev(cen[1]), rgt = lane_next(cen[0]);
      const double tl = cen[0] - lft, tm = cen[1] - cen[0], tr = rgt - cen[1];
      const double d2x[2] = {tm - tl, tr - tm};
      const double upd[2] = {up[0] - cen[0], up[1] - cen[1]};
      const double d2y[2] = {upd[0] - E[s][0], upd[1] - E[s][1]};
      E[s][0] = upd[0];
      E[s][1] = upd[1];
      double pnew[2], dnew[2];
      update_pair<QD>(d2x, d2y, dq, g[s], cen, pnew, dnew);
      if (EDGE) edge_fixups(r, cen, dq, pnew, dnew);
      if (s < K) {
        note_values<QD>(pnew);
#pragma unroll
        for (int v = 0; v < 2; v++) {
          P[s + 1][PH ^ 1][v] = pnew[v];
          D[s + 1][PH ^ 1][v] = dnew[v];
        }
      } else if (!DMA || r >= r0) {  // (the register path enters steady steps only once r >= r0)
        note_stored<QD>(pnew);
        store_pair<ALIGNED>(prd, (ptrdiff_t)r * ld + c0, outcol[0], outcol[1], pnew[0], pnew[1]);
        store_dp(r, dnew);
      }
    }
  }

  __device__ __forceinline__ bool steady(int j) const {
    // passes 1..K work on rows j-1 .. j-K; the row requested ahead must exist
    return j - K > jstart && j + (DMA ? 0 : PF) <= jend && (DMA || j - K >= r0) &&
           (PAIR ? j <= jtop : ((!A.phys_bottom || j - K >= 2) && (!A.phys_top || j <= A.ny - 2)));
    // (a task of a pair never starts at the window's bottom; without pairs pb, pt, nyt are A's own values --
    //  written out here because the compiler then keeps 26 registers fewer)
  }

  // ---- register-path driver ----------------------------------------------------------------
  template <int I, bool STEADY, int QD>
  __device__ __forceinline__ void reg_steps(int j, int &ring_j, int jlast) {
    if constexpr (I < UNROLL) {
      constexpr int PH = I % 2, SL = I % NSLOT;
      const int jj = j + I;
      if (STEADY || jj <= jlast) {
        if (STEADY || jj <= jend)
          lds[ring_j * 64 + lane] = make_double2(A.rho_dt * LG[SL][0], A.rho_dt * LG[SL][1]);
        double2 g[K + 1];
#pragma unroll
        for (int s = 1; s <= K; s++) {
          int rs = ring_j - s;
          if (rs < 0) rs += RING;
          g[s] = lds[rs * 64 + lane];
        }
        const double inP[2] = {LP[SL][0], LP[SL][1]}, inD[2] = {LD[SL][0], LD[SL][1]};
        if (STEADY || jj + PF <= jend) load_row(jj + PF, (SL + PF) % NSLOT);
        if constexpr (STEADY)
          steady_step<PH, QD>(jj, inP, inD, g);
        else
          general_step<PH, QD>(jj, inP, inD, g);
      }
      ring_j = ring_j + 1 == RING ? 0 : ring_j + 1;
      reg_steps<I + 1, STEADY, QD>(j, ring_j, jlast);
    }
  }

  // ---- DMA-path driver -----------------------------------------------------------------------
  template <int PH, bool STEADY, int QD>
  __device__ __forceinline__ void dma_step(int j, int &sp, int &sg, bool &drained) {
    // Row j was requested PFD steps ago.  vmcnt counts row requests and stores together in
    // issue order; while rows are still being requested, the PFD-1 steps since have each put
    // 3 younger requests behind row j's (their stores only add to that), so waiting for all
    // but 3*(PFD-1) operations guarantees that row j has landed while about half of the
    // requested rows stay in flight.  Once the last row has been requested one full drain
    // covers every remaining step.
    if (j + PFD - 1 <= jend) {
      vm_wait<3 * (PFD - 1)>();
    } else if (!drained) {
      vm_wait<0>();
      drained = true;
    }
    double inP[2], inD[2];
    double2 g[K + 1];
    dma_fetch(sp, sg, STEADY || j <= jend, inP, inD, g);
    if (j + PFD <= jend) {
      int np = sp + PFD, ng = sg + PFD;
      if (np >= NST) np -= NST;
      if (ng >= NSG) ng -= NSG;
      dma_row(j + PFD, np, ng);
    }
    if constexpr (STEADY)
      steady_step<PH, QD>(j, inP, inD, g);
    else
      general_step<PH, QD>(j, inP, inD, g);
    sp = sp + 1 == NST ? 0 : sp + 1;
    sg = sg + 1 == NSG ? 0 : sg + 1;
  }

  // The whole task with division form QD.  Returns false when an operand left the window in
  // which that form is exact: the task's output is then to be produced again with IEEE division.
  template <int QD>
  __device__ __forceinline__ bool run() {
    // the last output row r1-1 leaves pass K at step r1-1+K; later steps would only advance halo
    // rows that nobody stores
    const int jlast = min(jend + K, r1 - 1 + K);
    if constexpr (DMA) {
      // A row request writes LDS from the memory side, unordered with the LDS instructions this wavefront has
      // issued before.  Within a run every step's reads are waited for and the rho/dt * grad V row written in
      // step j is complete by the wait of step j+1, long before its slot is requested again.  A task that is
      // run a SECOND time requests its first rows into the slots the first run used last: the scaled row
      // written in the first run's last step could still be in the LDS queue and land after the freshly
      // requested row -- which the second run then scaled again.  Seen on a 127 x 1574 grid of subnormal
      // values in one-pass launches (1574 tasks of two rows, every one run twice): the first output row of a
      // few tasks per launch wrong, in 22 of 100 calls; none in 400 calls with this wait.
      lds_wait_all();
#pragma unroll
      for (int i = 0; i < PFD; i++)
        if (jstart + i <= jend) dma_row(jstart + i, i % NST, i % NSG);
      int sp = 0, sg = 0;
      bool drained = false;
      // Two wavefronts share a SIMD, and the arbiter serves the older one first: it ran at nearly the speed it
      // would have alone and the younger one finished a quarter later, alone on the SIMD.  They take turns at
      // the higher priority instead, two row steps each (the parity of the hardware wavefront slot tells the
      // two apart; should both slots of a SIMD have the same parity nothing changes): 1.5-2 % per launch.
      const int slot_parity = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1u);  // HW_ID, wave_id bit 0
      for (int j = jstart; j <= jlast; j += 2) {
        if ((((j - jstart) >> 1) & 1) == slot_parity) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        if (steady(j) && steady(j + 1)) {
          dma_step<0, true, QD>(j, sp, sg, drained);
          dma_step<1, true, QD>(j + 1, sp, sg, drained);
        } else {
          dma_step<0, false, QD>(j, sp, sg, drained);
          if (j + 1 <= jlast) dma_step<1, false, QD>(j + 1, sp, sg, drained);
        }
        // (a pair runs to the end whatever happens: its two wavefronts meet at barriers)
        if (!PAIR && QD != DIV_IEEE && left_safe_window()) {
          vm_wait<0>();  // rows still on their way into the rings must not land in the next run
          return false;
        }
      }
      if (PAIR) return !(QD != DIV_IEEE && left_safe_window());
    } else {
#pragma unroll
      for (int i = 0; i < PF; i++)
        if (jstart + i <= jend) load_row(jstart + i, i);
      int ring_j = 0;
      for (int j = jstart; j <= jlast; j += UNROLL) {
        if (steady(j) && steady(j + UNROLL - 1))
          reg_steps<0, true, QD>(j, ring_j, jlast);
        else
          reg_steps<0, false, QD>(j, ring_j, jlast);
        if (QD != DIV_IEEE && left_safe_window()) return false;
      }
    }
    return true;
  }

  // Returns false when the task has to be produced again with IEEE division and cannot do that itself: the
  // task of a pair, whose partner may be long gone -- the caller runs it again as a plain task.
  __device__ __forceinline__ bool run_task() {
    const bool ok = run<DIV>();
    if (PAIR) return ok;
#ifdef NSK_POISSON_TRACE
    if (lane == 0) {
      atomicAdd(&nsk_trace_cnt[0], 1ull);
      if (!ok) atomicAdd(&nsk_trace_cnt[1], 1ull);
    }
#endif
    if (!ok) {
      reset();
      run<DIV_IEEE>();
    }
    return true;
  }

  // The task once more with IEEE division, by a wavefront that has just run its paired form on the same rings.
  __device__ __forceinline__ void run_again() { run<DIV_IEEE>(); }
};

#ifdef NSK_POISSON_TRACE
// Debug build only: per-task start/end clock (100 MHz) and hardware placement.
static __device__ unsigned long long nsk_trace_buf[NSK_POISSON_TRACE * 4];
#endif

template <int K, bool ALIGNED, int DIV, bool STREAM = false>
__global__ void __launch_bounds__(64) NSK_WAVES_ATTR k_poisson(PoissonArgs A) {
  constexpr int LDS_VEC = ALIGNED ? DmaGeometry<K>::LDS_BYTES / 16 : (K + 1) * 64;
  __shared__ double2 shared[LDS_VEC];
#ifdef NSK_POISSON_TRACE
  const unsigned long long trace_t0 = wall_clock64();
#endif
  // Tasks: first the interior strips (1 .. nstrips-2) x their chunks, then the two edge strips
  // (0 and nstrips-1) with their own, shorter chunks.
  int task = blockIdx.x, sx, r0, r1;
  if (A.tall_extra > 0 && task < A.n_interior_tasks) {
    // The two wavefronts that share a SIMD do not share it evenly: the one that arrived first is served
    // first and runs at nearly the speed it would have alone, the other fills the gaps and then finishes
    // alone.  Workgroups are handed out in the order of their numbers, so the first half of the numbers
    // are the first arrivals: they get the tall chunk of a pair, the second half the short one, and the
    // two finish together.  XCD-aware as below: workgroup i runs on XCD i % 8, and XCD x works on a
    // contiguous band of pairs (both chunks of each).
    const int nb = A.pair_tasks_pad >> 3;  // pairs per band
    int pt, tall;
    if (A.xcd_bands) {
      const int x = task & 7, k = task >> 3;
      tall = k < nb;
      pt = x * nb + (tall ? k : k - nb);
    } else {
      tall = task < A.pair_tasks_pad;
      pt = tall ? task : task - A.pair_tasks_pad;
    }
    if (pt >= A.pair_tasks) return;  // padding of the pair count to a multiple of 8
    const int nint = A.nstrips - 2;
    sx = 1 + pt % nint;
    const int base = A.row_begin + (pt / nint) * 2 * A.rows_per_chunk;
    if (tall) {
      r0 = base;
      r1 = min(r0 + A.rows_per_chunk + A.tall_extra, A.row_end);
    } else {
      r0 = base + A.rows_per_chunk + A.tall_extra;
      r1 = min(base + 2 * A.rows_per_chunk, A.row_end);
    }
    if (r0 >= r1) return;
  } else if (task < A.n_interior_tasks) {
    if (A.xcd_bands) {
      // Workgroup i runs on XCD i % 8, each with its own L2.  Give every XCD a contiguous band
      // of chunks over all strips, so that tasks that read the same halo columns (and rows)
      // share an L2: task t is the (t / 8)-th task of band t % 8.
      const int base = A.n_interior_tasks >> 3, rem = A.n_interior_tasks & 7;
      const int x = task & 7;
      task = x * base + min(x, rem) + (task >> 3);
    }
    const int nint = A.nstrips - 2;
    sx = 1 + task % nint;
    int c = task / nint;
    if (c < A.nchunks1) {
      r0 = A.row_begin + c * A.rows_per_chunk;
      r1 = min(r0 + A.rows_per_chunk, A.row_end);
    } else {
      r0 = A.row2_begin + (c - A.nchunks1) * A.rows_per_chunk;
      r1 = min(r0 + A.rows_per_chunk, A.row2_end);
    }
  } else {
    task -= A.n_interior_tasks;
    const int nedge = A.nstrips >= 2 ? 2 : 1;
    sx = (task % nedge) == 0 ? 0 : A.nstrips - 1;
    int c = task / nedge;
    if (c < A.nchunks1_edge) {
      r0 = A.row_begin + c * A.rows_per_chunk_edge;
      r1 = min(r0 + A.rows_per_chunk_edge, A.row_end);
    } else {
      r0 = A.row2_begin + (c - A.nchunks1_edge) * A.rows_per_chunk_edge;
      r1 = min(r0 + A.rows_per_chunk_edge, A.row2_end);
    }
  }
  constexpr int HK = (K + 1) & ~1;
  constexpr int WOUT = 128 - 2 * HK;
  // A strip is an edge strip when any of its 128 columns is not an interior column.
  const int cfirst = sx * WOUT - HK, clast = cfirst + 127;
  if (cfirst < 1 || clast > A.nx - 2) {
    Task<K, ALIGNED, DIV, true, STREAM> t(A, shared, sx, r0, r1);
    t.run_task();
  } else {
    Task<K, ALIGNED, DIV, false, STREAM> t(A, shared, sx, r0, r1);
    t.run_task();
  }
#ifdef NSK_POISSON_TRACE
  if (threadIdx.x == 0 && blockIdx.x < NSK_POISSON_TRACE) {
    nsk_trace_buf[blockIdx.x * 4 + 0] = trace_t0;
    nsk_trace_buf[blockIdx.x * 4 + 1] = wall_clock64();
    nsk_trace_buf[blockIdx.x * 4 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    nsk_trace_buf[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
}

// The same tasks, two per workgroup (a wavefront each), so that two of them can be a PAIR (see Task): the
// chunks of a column strip are, from the bottom, one plain chunk, then pairs of a chunk that walks down and the
// chunk above it that walks up, and a plain chunk on top when one is left over.  Every task then has halo rows
// at one end only (the plain ones because their other end is the window's).  The host lists the tasks
// (launch_pairs): entry 2*workgroup + wavefront.
struct PairTask {
  int sx, r0, r1, kind;  // kind: 0 = none, 1 = plain, 2 = walks down from r1-1 (partner: the other wavefront), 3 = walks up from r0
};

template <int K, int DIV, bool STREAM>
__global__ void __launch_bounds__(128) NSK_WAVES_ATTR k_poisson_pairs(PoissonArgs A, const PairTask *__restrict__ table) {
  constexpr int LDS_VEC = DmaGeometry<K>::LDS_BYTES / 16;
  __shared__ double2 shared[2 * LDS_VEC + 4096 / 16];  // the two wavefronts' staging rings, the exchange slots
#ifdef NSK_POISSON_TRACE
  const unsigned long long trace_t0 = wall_clock64();
#endif
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const PairTask *e = table + (2 * blockIdx.x + wave);
  const int kind = __builtin_amdgcn_readfirstlane(e->kind), sx = __builtin_amdgcn_readfirstlane(e->sx);
  const int r0 = __builtin_amdgcn_readfirstlane(e->r0), r1 = __builtin_amdgcn_readfirstlane(e->r1);
  if (kind == 0) return;
  double2 *rings = shared + wave * LDS_VEC;
  constexpr int HK = (K + 1) & ~1;
  constexpr int WOUT = 128 - 2 * HK;
  const int cfirst = sx * WOUT - HK, clast = cfirst + 127;
  if (kind >= 2) {
    const unsigned xch = (unsigned)(uintptr_t)(lds_void_t *)(shared + 2 * LDS_VEC);
    // A pressure value left the range of the fast division: the task is produced again with IEEE division, as
    // a plain task with halo rows at both ends, from the launch's inputs -- nothing is needed from the
    // partner, which decides for itself.
    if (cfirst < 1 || clast > A.nx - 2) {  // (a strip with boundary-column work)
      bool ok;
      {
        Task<K, true, DIV, true, STREAM, true> t(A, rings, sx, r0, r1, kind == 2 ? r1 - 1 : -1, xch, wave);
        ok = t.run_task();
      }
      if (DIV != DIV_IEEE && !ok) {
        Task<K, true, DIV_IEEE, true, STREAM> again(A, rings, sx, r0, r1);
        again.run_again();
      }
    } else {
      bool ok;
      {
        Task<K, true, DIV, false, STREAM, true> t(A, rings, sx, r0, r1, kind == 2 ? r1 - 1 : -1, xch, wave);
        ok = t.run_task();
      }
      if (DIV != DIV_IEEE && !ok) {
        Task<K, true, DIV_IEEE, false, STREAM> again(A, rings, sx, r0, r1);
        again.run_again();
      }
    }
  } else if (cfirst < 1 || clast > A.nx - 2) {
    Task<K, true, DIV, true, STREAM> t(A, rings, sx, r0, r1);
    t.run_task();
  } else {
    Task<K, true, DIV, false, STREAM> t(A, rings, sx, r0, r1);
    t.run_task();
  }
#ifdef NSK_POISSON_TRACE
  const unsigned slot = 2 * blockIdx.x + wave;
  if ((threadIdx.x & 63) == 0 && slot < NSK_POISSON_TRACE) {
    nsk_trace_buf[slot * 4 + 0] = trace_t0;
    nsk_trace_buf[slot * 4 + 1] = wall_clock64();
    // (hardware ids in the low halves; the task in the high halves: kind, strip / first row)
    nsk_trace_buf[slot * 4 + 2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)kind << 32) |
                                  ((unsigned long long)sx << 40);
    nsk_trace_buf[slot * 4 + 3] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((unsigned long long)r0 << 32);
  }
#endif
}

// ---- host side: how a launch is cut into tasks -------------------------------------------------

// Whole runs of passes in one launch on grids whose fields fit on chip (ns_poisson_small.hip): tiles of
// tile_cols x tile_rows cells, gx x gy of them, a frame of `halo` cells around each, `slots` cells per thread.
struct SmallPlan {
  int tile_cols, tile_rows, halo, gx, gy, slots;
};
struct PoissonArgs;
bool small_grid_applies(const nsk_ctx *ctx, const nsk_ns_params *p, int count, SmallPlan *plan);
int small_grid_iters(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV, const nsk_ns_params *p, int count,
                     const PoissonArgs &P, int div, double *part_pr, double *part_dp, const SmallPlan &plan);

// Geometry of one launch: strips of WOUT output columns x chunks of rows.  The first and the last
// strip carry the boundary-column work on top, so their chunks are shorter.
struct TaskPlan {
  int K, nstrips, rows, rows_edge, nchunks, nchunks_edge;
  long tasks;
  double est_us_per_pass;
  int nchunks1 = -1, nchunks1_edge = -1;  // chunks that belong to the first of two row ranges (-1: all)
  bool pairs = false;     // the tasks of every strip in pairs (k_poisson_pairs, two tasks per workgroup)
  long slots = 0;         // wavefront slots the launch takes: the tasks, or two per workgroup
};

// Can a launch of K passes with this plan run its tasks in pairs?  (The two wavefronts of a workgroup and the
// exchange slots must fit a quarter of a compute unit's LDS; a chunk must outlast the K exchanges.)
inline bool pairs_fit(int K, int nstrips, int rows, int rows_edge) {
  return 2 * DmaGeometry<1>::LDS_BYTES + 2 * 1024 * (K - 1) + 4096 <= 40960 && nstrips >= 3 && rows >= K + 2 && rows_edge >= 2;
}

#ifndef NSK_EDGE_PERCENT_PAIRS
#define NSK_EDGE_PERCENT_PAIRS 62
#endif
inline TaskPlan plan_tasks(int K, int nx, int ny, int rows, int edge_percent, bool pairs = false) {
  TaskPlan t;
  const int HK = (K + 1) & ~1, WOUT = 128 - 2 * HK;
  t.K = K;
  t.nstrips = (nx - 1 + WOUT - 1) / WOUT;  // columns 0..nx-2; nx-1 rides with nx-2
  if (rows < 2) rows = 2;
  if (rows > ny) rows = ny;
  int re = (int)((long)rows * (edge_percent > 0 ? edge_percent : (pairs ? NSK_EDGE_PERCENT_PAIRS : 75)) / 100);
  if (re < 2) re = 2;
  if (re > rows) re = rows;
  t.rows = rows;
  t.rows_edge = re;
  t.nchunks = (ny + rows - 1) / rows;
  t.nchunks_edge = (ny + re - 1) / re;
  const int nint = t.nstrips > 2 ? t.nstrips - 2 : 0, nedge = t.nstrips >= 2 ? 2 : 1;
  t.tasks = (long)nint * t.nchunks + (long)nedge * t.nchunks_edge;
  t.est_us_per_pass = 0.0;
  t.slots = t.tasks;
  if (pairs && pairs_fit(K, t.nstrips, rows, re)) {
    // as launch_pairs lists them: per strip one plain chunk, then pairs, a plain one when a chunk is left over
    // (chunks of the edge height on strips that reach the rim columns); plain tasks two per workgroup
    long npairs = 0, plain = 0;
    for (int sx = 0; sx < t.nstrips; sx++) {
      const bool edge = sx * WOUT - HK < 1 || sx * WOUT - HK + 127 > nx - 2;
      const int h = edge ? re : rows, n = edge ? t.nchunks_edge : t.nchunks, last_rows = ny - (n - 1) * h;
      if (h < K + 2) {  // (too short for the K exchanges)
        plain += n;
        continue;
      }
      int m = n - 1, p = m / 2, left = m % 2;
      if (p > 0 && left == 0 && last_rows < 2) p -= 1, left = 2;  // (a last chunk of one row is no partner)
      npairs += p;
      plain += 1 + left;
    }
    t.pairs = true;
    t.slots = 2 * (npairs + (plain + 1) / 2);
  }
  return t;
}

// One launch over two row ranges of n1 and n2 rows (either may be empty), chunks of `rows` rows
// (a range shorter than a chunk is one short chunk).
inline TaskPlan plan_two_ranges(int K, int nx, int n1, int n2, int rows, int edge_percent) {
  TaskPlan t = plan_tasks(K, nx, n1 + n2 > 0 ? n1 + n2 : 1, rows, edge_percent);  // strips, clamped chunk heights
  auto chunks = [](int n, int r) { return n > 0 ? (n + r - 1) / r : 0; };
  t.nchunks1 = chunks(n1, t.rows);
  t.nchunks1_edge = chunks(n1, t.rows_edge);
  t.nchunks = t.nchunks1 + chunks(n2, t.rows);
  t.nchunks_edge = t.nchunks1_edge + chunks(n2, t.rows_edge);
  const int nint = t.nstrips > 2 ? t.nstrips - 2 : 0, nedge = t.nstrips >= 2 ? 2 : 1;
  t.tasks = (long)nint * t.nchunks + (long)nedge * t.nchunks_edge;
  return t;
}

// Estimated time of a launch with this plan, per pass.  One wavefront per task, at most
// NSK_WAVES resident per SIMD; a task takes rows + 2K row steps, and a row step costs a fixed part
// (row requests, LDS traffic, stores, index arithmetic) plus a part per pass.  Fitted on MI355X
// (tools/tune_poisson.py, grids of 159 x 255 up to 2559 x 8191): 0.30 + 0.23 K us per row step with
// two wavefronts on a SIMD, 0.18 + 0.14 K us with one; the memory side moves the three inputs (with
// their halo re-reads) and the two outputs at about 5.2 TB/s from HBM, about twice that when the
// five fields fit the 256 MB last-level cache; about 5 us per launch on top.
inline double plan_cost(const TaskPlan &t, int nx, int ny, int num_simds) {
  const int K = t.K, HK = (K + 1) & ~1, WOUT = 128 - 2 * HK;
  const long slots = (long)num_simds * NSK_WAVES;
  const double rounds = (double)((t.slots + slots - 1) / slots);
  // (the two wavefronts of a workgroup of a paired launch are placed without regard to the wavefronts a SIMD
  //  already holds: only with no more workgroups than compute units does every wavefront have a SIMD to itself)
  const bool shared_simds = t.pairs ? t.slots > num_simds / 2 : t.tasks > num_simds;
  const double per_step = shared_simds ? 0.30 + 0.23 * K : 0.18 + 0.14 * K;
  const int halo_steps = t.pairs ? K + 1 : 2 * K;  // (a task of a pair has halo rows at one end only)
  const double alu = rounds * (t.rows + halo_steps) * per_step;
  const double field_bytes = 8.0 * nx * ny;
  const double rate = 5.0 * field_bytes < 200e6 ? 10.4e6 : 5.2e6;  // bytes per us
  const double mem = field_bytes * (3.0 * (1.0 + (double)halo_steps / t.rows) * (128.0 / WOUT) + 2.0) / rate;
  return ((alu > mem ? alu : mem) + 5.0) / K;
}

// Passes per launch and chunk height.  k_forced / rows_forced > 0 pin the respective choice;
// k_limit caps the automatic K (no more passes than are asked for).
inline TaskPlan choose_plan(int nx, int ny, int num_simds, bool staged, int k_forced, int k_limit,
                            int rows_forced, int edge_percent, int pairs = 0) {
  // automatic K stops where the kernel no longer fits two wavefronts per SIMD: 6 passes with the
  // LDS-staged input path, 5 when the input rows travel through registers
  int klo = 1, khi = staged ? 6 : 5;
  if (k_forced > 0) klo = khi = k_forced;
  if (khi > k_limit) khi = k_limit;
  if (klo > khi) klo = khi;
  const long slots = (long)num_simds * NSK_WAVES;
  TaskPlan best{};
  bool have = false;
  for (int K = klo; K <= khi; K++)
    for (int form = pairs == 2 ? 1 : 0; form <= (pairs ? 1 : 0); form++) {  // plain tasks; tasks in pairs where the launch can have them
                                                                               // (pairs: 0 never, 1 the cheaper form by the estimate, 2 wherever they fit)
      const bool pr = form == 1;
      const long alone = pr ? num_simds / 2 : num_simds;
      int rlo = rows_forced, rhi = rows_forced;
      if (rows_forced <= 0) {
        // from the smallest height that fits one round to the one that leaves every wavefront a SIMD to itself
        rlo = 2;
        while (rlo < ny && plan_tasks(K, nx, ny, rlo, edge_percent, pr).slots > slots) rlo++;
        rhi = rlo;
        while (rhi < ny && plan_tasks(K, nx, ny, rhi, edge_percent, pr).slots > alone) rhi++;
        if (rhi > rlo + 256) rhi = rlo + 256;
      }
      for (int r = rlo; r <= rhi; r++) {
        TaskPlan t = plan_tasks(K, nx, ny, r, edge_percent, pr);
        if (pr && !t.pairs && pairs != 2) continue;  // (this height cannot be paired: the plain form has it already)
        t.est_us_per_pass = plan_cost(t, nx, ny, num_simds);
        if (!have || t.est_us_per_pass < best.est_us_per_pass) {
          best = t;
          have = true;
        }
      }
    }
  return best;
}

// Should a launch on an nx x ny window request Pr and dPrdτ with the non-temporal hint?  Measured on
// MI355X (256 MB last-level cache) over grids of 159 x 255 .. 3839 x 6143, five fields of 8 nx ny
// bytes per launch: 3-5 % faster where the five fields take 1.4-2 times the cache (2399 x 3839 ..
// 2879 x 4607: what the previous launch wrote is partly still there, and input rows that are not
// kept leave it there), even at 1.2 and 2.4 times, 2-4 % slower on grids the cache holds
// (they are served from it either way, the hint only shortens their stay) and 6 % slower at 3.5
// times, where the halo rows a neighbouring task reads again are then fetched from memory again.
inline bool stream_inputs(int nx, int ny) {
  const double bytes = 5.0 * 8.0 * nx * ny, llc = 256.0 * 1024 * 1024;
  return bytes >= 1.3 * llc && bytes <= 2.3 * llc;
}

// `plan` cuts the rows [A.row_begin, A.row_end) (plan_tasks was given their number).
// One launch of k_poisson_pairs: the task list (cached per shape of the launch on the context), two tasks per
// workgroup.  XCD-aware like k_poisson: workgroup i runs on XCD i % 8, and XCD x is given a contiguous band
// of rows over all strips.
template <int K>
int launch_pairs(nsk_ctx *ctx, hipStream_t stream, const PoissonArgs &A, int div, const TaskPlan &plan, bool hint) {
  const std::array<int, 8> key = {K, plan.nstrips, A.row_begin, A.row_end, plan.rows, plan.rows_edge, ctx->poisson_xcd, ctx->poisson_pairs};
  auto it = ctx->pair_tables.find(key);
  if (it == ctx->pair_tables.end()) {
    if (ctx->pair_tables.size() >= 64) {  // (never in practice: a few shapes per grid)
      NSK_HIP(hipStreamSynchronize(stream));
      for (auto &kv : ctx->pair_tables) (void)hipFree(kv.second.first);
      ctx->pair_tables.clear();
    }
    struct Group {
      PairTask t[2];
    };
    std::vector<Group> groups;
    std::vector<PairTask> plain;
    auto cut = [&](int rows, std::vector<std::pair<int, int>> &out) {
      out.clear();
      for (int r = A.row_begin; r < A.row_end; r += rows) out.push_back({r, r + rows < A.row_end ? r + rows : A.row_end});
    };
    std::vector<std::pair<int, int>> ch, che;
    cut(plan.rows, ch);
    cut(plan.rows_edge, che);
    constexpr int HK = (K + 1) & ~1, WOUT = 128 - 2 * HK;
    for (int sx = 0; sx < plan.nstrips; sx++) {
      // (a strip whose 128 columns reach the rim columns does the boundary-column work on top: shorter chunks)
      const bool edge = sx * WOUT - HK < 1 || sx * WOUT - HK + 127 > A.nx - 2;
      const std::vector<std::pair<int, int>> &c_ = edge ? che : ch;
      const int n = (int)c_.size();
      plain.push_back(PairTask{sx, c_[0].first, c_[0].second, 1});
      for (int c = 1; c < n;) {
        // (2: the same launch, no pairs: for timing; the walker down needs K + 2 rows, its partner 2)
        if (ctx->poisson_pairs != 2 && c + 1 < n && c_[c].second - c_[c].first >= K + 2 && c_[c + 1].second - c_[c + 1].first >= 2) {
          groups.push_back(Group{{PairTask{sx, c_[c].first, c_[c].second, 2}, PairTask{sx, c_[c + 1].first, c_[c + 1].second, 3}}});
          c += 2;
        } else {
          plain.push_back(PairTask{sx, c_[c].first, c_[c].second, 1});
          c += 1;
        }
      }
    }
    // plain tasks two by two, neighbours in the grid together
    std::sort(plain.begin(), plain.end(), [](const PairTask &a, const PairTask &b) { return a.r0 != b.r0 ? a.r0 < b.r0 : a.sx < b.sx; });
    for (size_t i = 0; i < plain.size(); i += 2)
      groups.push_back(Group{{plain[i], i + 1 < plain.size() ? plain[i + 1] : PairTask{0, 0, 0, 0}}});
    std::stable_sort(groups.begin(), groups.end(), [](const Group &a, const Group &b) {
      return a.t[0].r0 != b.t[0].r0 ? a.t[0].r0 < b.t[0].r0 : a.t[0].sx < b.t[0].sx;
    });
    const size_t ng = groups.size(), per_band = (ng + 7) / 8, nblocks = ctx->poisson_xcd ? 8 * per_band : ng;
    std::vector<PairTask> table(2 * nblocks, PairTask{0, 0, 0, 0});
    for (size_t b = 0; b < nblocks; b++) {
      const size_t src = ctx->poisson_xcd ? (b & 7) * per_band + (b >> 3) : b;
      if (src < ng && (!ctx->poisson_xcd || (b >> 3) < per_band)) {
        table[2 * b] = groups[src].t[0];
        table[2 * b + 1] = groups[src].t[1];
      }
    }
    PairTask *dev = nullptr;
    NSK_HIP(hipMalloc(&dev, table.size() * sizeof(PairTask)));
    NSK_HIP(hipMemcpy(dev, table.data(), table.size() * sizeof(PairTask), hipMemcpyHostToDevice));
    it = ctx->pair_tables.emplace(key, std::make_pair((void *)dev, (unsigned)nblocks)).first;
  }
  const PairTask *table = static_cast<const PairTask *>(it->second.first);
  const unsigned blocks = it->second.second;
  if (div == DIV_FMA2 && hint) k_poisson_pairs<K, DIV_FMA2, true><<<blocks, 128, 0, stream>>>(A, table);
  else if (div == DIV_FMA2) k_poisson_pairs<K, DIV_FMA2, false><<<blocks, 128, 0, stream>>>(A, table);
  else k_poisson_pairs<K, DIV_IEEE, false><<<blocks, 128, 0, stream>>>(A, table);
  NSK_LAUNCH_CHECK();
  return 0;
}

template <int K>
int launch_k(nsk_ctx *ctx, hipStream_t stream, PoissonArgs &A, bool aligned, int div, const TaskPlan &plan) {
  A.nstrips = plan.nstrips;
  A.rows_per_chunk = plan.rows;
  A.nchunks = plan.nchunks;
  A.rows_per_chunk_edge = plan.rows_edge;
  A.nchunks_edge = plan.nchunks_edge;
  A.nchunks1 = plan.nchunks1 >= 0 ? plan.nchunks1 : plan.nchunks;
  A.nchunks1_edge = plan.nchunks1_edge >= 0 ? plan.nchunks1_edge : plan.nchunks_edge;
  A.n_interior_tasks = (plan.nstrips > 2 ? plan.nstrips - 2 : 0) * plan.nchunks;
  A.xcd_bands = ctx->poisson_xcd;
  unsigned tasks = (unsigned)plan.tasks;
  A.tall_extra = A.pair_tasks = A.pair_tasks_pad = 0;
  // pairs of a tall and a short chunk: whole-window launches of two wavefronts per SIMD only (with one
  // wavefront per SIMD nobody shares, and the two-range launches of the edges-first schedule are short anyway)
  const int extra = plan.rows * ctx->poisson_tall_percent / 100;
  if (extra > 0 && extra < plan.rows && plan.nchunks1 < 0 && plan.nstrips > 2 && plan.tasks > ctx->num_simds) {
    const int nint = plan.nstrips - 2, nrows = A.row_end - A.row_begin;
    const int npairs = (nrows + 2 * plan.rows - 1) / (2 * plan.rows);
    A.tall_extra = extra;
    A.pair_tasks = nint * npairs;
    A.pair_tasks_pad = (A.pair_tasks + 7) / 8 * 8;
    tasks = tasks - (unsigned)A.n_interior_tasks + 2u * (unsigned)A.pair_tasks_pad;
    A.n_interior_tasks = 2 * A.pair_tasks_pad;
  }
#define NSK_LAUNCH(AL, DV) k_poisson<K, AL, DV><<<tasks, 64, 0, stream>>>(A)
  // the hint is built for the form every shipped grid takes; the other forms do without
  const bool hint = ctx->poisson_stream < 0 ? stream_inputs(A.nx, A.ny) : ctx->poisson_stream != 0;
  // Tasks in pairs that share no halo rows: pitched storage, one row range, chunks tall enough for the K
  // exchanges, and a caller that may allocate (the task list lives in device memory).
  if (aligned && ctx->poisson_pairs && !ctx->no_alloc_call && (div == DIV_FMA2 || div == DIV_IEEE) && plan.nchunks1 < 0 &&
      A.row2_end <= A.row2_begin && plan.pairs && pairs_fit(K, plan.nstrips, plan.rows, plan.rows_edge) && A.tall_extra == 0) {
    ctx->last_launch_pairs = 1;
    return launch_pairs<K>(ctx, stream, A, div, plan, hint);
  }
  ctx->last_launch_pairs = 0;
  if (aligned && div == DIV_FMA2 && hint) {
    k_poisson<K, true, DIV_FMA2, true><<<tasks, 64, 0, stream>>>(A);
  } else if (aligned) {
    if (div == DIV_FMA2) NSK_LAUNCH(true, DIV_FMA2);
    else if (div == DIV_FMA3) NSK_LAUNCH(true, DIV_FMA3);
    else if (div == DIV_FMA4) NSK_LAUNCH(true, DIV_FMA4);
    else NSK_LAUNCH(true, DIV_IEEE);
  } else {
    if (div == DIV_FMA2) NSK_LAUNCH(false, DIV_FMA2);
    else if (div == DIV_FMA3) NSK_LAUNCH(false, DIV_FMA3);
    else if (div == DIV_FMA4) NSK_LAUNCH(false, DIV_FMA4);
    else NSK_LAUNCH(false, DIV_IEEE);
  }
#undef NSK_LAUNCH
  NSK_LAUNCH_CHECK();
  return 0;
}

}  // namespace nsk_poisson
