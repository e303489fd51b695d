// Lexicographic SOR (pressure: scripts/code1.m:68-78 / scripts/code3_fun.jl:104-124; density:
// scripts/code3_fun.jl:224-240) on grids that one workgroup's LDS does not hold, on the whole chip and
// still bit-identical to the serial i-outer / j-inner loop.
//
// The grid is cut into tiles of 64 x BJ cells.  Task (I, J, k) = tile (I, J) in sweep k.  A cell needs
// THIS sweep's values to its left and below and the PREVIOUS sweep's values to its right and above, so
// task (I, J, k) may run once (I-1, J, k), (I, J-1, k), (I+1, J, k-1) and (I, J+1, k-1) are complete:
// any order that respects this produces the operands of the serial sweep, hence its bits.  Tasks on
// I + J + 2k = const are independent of each other -- with enough sweeps in flight that is half of
// all tiles at once.
//
// Storage: a ring of R copies of the unknown; sweep k reads its own tile and the right / upper halo
// from copy (k-1) mod R (sweep 0 = the caller's array), the left / lower halo from copy k mod R and
// writes its tile to copy k mod R.  Nothing is updated in place, so the sweeps that run ahead of the
// stopping test (max|old - new| < maxError belongs to whole sweeps) are speculation that is simply not
// copied out: the finishing kernel copies the final sweep's copy into the caller's array.  Copy
// k mod R is taken again by sweep k + R only after sweep k has been judged.
//
// Execution: one wavefront per task (64 lanes = 64 columns of the tile; a lane walks up its column, the
// columns one step apart: the anti-diagonal order inside the tile; the left neighbour's new value
// arrives by a one-lane wave shift, the lower one is the lane's own previous value).  Workgroups of
// one wavefront take task numbers from a counter.  Numbering: rounds of all tiles, tiles with I + J
// even first; tile (I, J) of round rho works on sweep rho - (I + J) / 2.  Every task a task waits for
// has a smaller number, so whoever holds it is running: no co-residency is needed and nothing can
// deadlock.  Completion travels through one progress word per tile (last sweep completed): the tile's
// values are stored write-through, drained, then the word is stored; a consumer polls relaxed, runs
// one agent-scope acquire and loads.  Every spin is bounded and ends the solve with an error.
#include <vector>

#include "nsk_internal.h"

namespace {

typedef unsigned long long u64;
constexpr int TW = 64;        // tile width = lanes
constexpr int LDT = TW + 2;   // row pitch of a tile with its halo columns in LDS
enum { C_NEXT = 0, C_JUDGED = 1, C_FINAL = 2, C_ERR = 3, C_WORDS = 16 };
constexpr unsigned FINAL_ERROR = 0xffffffffu;
constexpr unsigned SPIN_LIMIT = 1u << 22;  // polls of one wait (about a microsecond each)

struct GridSor {
  double *a;                   // the unknown, (nx+2) x (ny+2): sweep 0 and, at the end, the result
  const double *c1, *c2, *c3;  // pressure: tmp1, tmp2, rt; density: fx, fy, unused
  double beta, dx, dy, maxError;
  int nx, ny, maxit;
  double *ring;                // R copies of the unknown
  int R;
  unsigned *ctrl;              // C_WORDS control words
  unsigned *prog;              // per tile: last sweep completed
  u64 *smax;                   // per ring copy: max|old-new| of its sweep (bit pattern), then [R] = the final one
  const int *tiles;            // task order within a round: I | J << 16
  int TI, TJ, BJ, NT, total;
  int P;                       // sweeps start in batches of R, a batch every P >= R rounds (see nsk_sor_grid)
};

// Debug builds (-DNSK_SOR_GRID_TRACE): every workgroup sums the time its tasks spend in each phase (100 MHz
// clock) into the words behind the tile order; tools/trace_sor_grid.py prints the averages.
#ifdef NSK_SOR_GRID_TRACE
#define TRACE_MARK(i)                                  \
  do {                                                 \
    const u64 now_ = wall_clock64();                   \
    if (lane == 0) tr[i] += now_ - tlast;              \
    tlast = now_;                                      \
  } while (0)
#else
#define TRACE_MARK(i) \
  do {                \
  } while (0)
#endif

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned ld_word(unsigned *p) { return __hip_atomic_load(p, RLX, AGENT); }
__device__ __forceinline__ void st_word(unsigned *p, unsigned v) { __hip_atomic_store(p, v, RLX, AGENT); }

// value of the lane below (lane 0 receives 0): v_mov_b32 with the DPP control wave_shr:1
__device__ __forceinline__ double from_lane_below(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

template <int MODE>
__global__ void __launch_bounds__(64) k_sor_grid(GridSor G) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x;
  const int ldg = G.nx + 2;
  const size_t ncell = (size_t)ldg * (G.ny + 2);
  const int tile_elems = (G.BJ + 2) * LDT;
  double *a = lds, *k1 = a + tile_elems, *k2 = k1 + tile_elems, *rt = k2 + tile_elems;
  const double beta = G.beta, dx = G.dx, dy = G.dy;
#ifdef NSK_SOR_GRID_TRACE
  u64 tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = wall_clock64();
  struct Flush {
    u64 *tr, *out;
    int lane;
    __device__ ~Flush() {
      if (lane == 0)
        for (int i = 0; i < 8; i++) out[i] = tr[i];
    }
  } flush{tr, reinterpret_cast<u64 *>(const_cast<int *>(G.tiles) + G.NT + (G.NT & 1)) + 8 * blockIdx.x, lane};
#endif
  for (;;) {
    TRACE_MARK(0);  // between tasks: the counter, skipped task numbers
    unsigned n = 0;
    if (lane == 0) n = __hip_atomic_fetch_add(G.ctrl + C_NEXT, 1u, RLX, AGENT);
    n = __builtin_amdgcn_readfirstlane(n);
    if (n >= (unsigned)G.total) return;
    if (ld_word(G.ctrl + C_FINAL) != 0) return;
    const int rho = (int)(n / (unsigned)G.NT) + 1, idx = (int)(n % (unsigned)G.NT);
    const int packed = G.tiles[idx], I = packed & 0xffff, J = packed >> 16;
    const int x = rho - (I + J) / 2;
    if (x < 1) continue;
    const int batch = (x - 1) / G.P, m = (x - 1) - batch * G.P;
    if (m >= G.R) continue;  // the gap between two batches of sweeps
    const int k = batch * G.R + 1 + m;
    if (k > G.maxit) continue;
    const int w = min(TW, G.nx - TW * I), h = min(G.BJ, G.ny - G.BJ * J);
    // 0-based global index of tile element (c, rr), c = 0..w+1, rr = 0..h+1 (halo included)
    const size_t g00 = (size_t)(G.BJ * J) * ldg + (size_t)(TW * I);
    constexpr int CH = 12;  // rows requested together, so that their latencies overlap
    // ---- the coefficients of the tile (nobody writes them): requested before the waits ----------
    // Straight-line code: rows past the tile's last row and lanes past its last column repeat the last
    // one (the same value goes to the same place again).
    const int cl = min(lane, w + 1);
    for (int rr0 = 0; rr0 <= h + 1; rr0 += CH) {
      double v1[CH], v2[CH], v3[CH];
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const size_t g = g00 + (size_t)min(rr0 + u, h + 1) * ldg + cl;
        v1[u] = G.c1[g];
        v2[u] = G.c2[g];
        v3[u] = MODE == 0 ? G.c3[g] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < CH; u++) {
        const int t = min(rr0 + u, h + 1) * LDT + cl;
        if (MODE == 0) {
          k1[t] = v1[u];
          k2[t] = beta * v2[u];
          rt[t] = v3[u];
        } else {
          k1[t] = dx * v1[u];
          k2[t] = dy * v2[u];
        }
      }
    }
    if (w + 1 >= TW) {  // columns 64 and 65 of a full-width tile: a lane per row, two rounds of rows
      const int c = TW + (lane & 1);
      for (int rr0 = 0; rr0 <= h + 1; rr0 += TW / 2) {
        const int rr = min(rr0 + (lane >> 1), h + 1);
        const size_t g = g00 + (size_t)rr * ldg + c;
        const int t = rr * LDT + c;
        const double v1 = G.c1[g], v2 = G.c2[g], v3 = MODE == 0 ? G.c3[g] : 0.0;
        if (MODE == 0) {
          k1[t] = v1;
          k2[t] = beta * v2;
          rt[t] = v3;
        } else {
          k1[t] = dx * v1;
          k2[t] = dy * v2;
        }
      }
    }
    TRACE_MARK(1);  // coefficients
    // ---- what this task waits for ---------------------------------------------------------------
    // One lane per word, all of them asked for together: lanes 0..5 hold the (word, value) pairs of this
    // task, lane 6 watches the word that ends the solve.
    const unsigned uk = (unsigned)k;
    unsigned *word = G.ctrl + C_FINAL;
    unsigned need = 0;  // lanes without a condition: satisfied by any value
    if (lane == 0 && I == 0 && J == 0 && k > G.R) word = G.ctrl + C_JUDGED, need = uk - (unsigned)G.R;
    if (lane == 1 && I > 0) word = G.prog + (J * G.TI + I - 1), need = uk;
    if (lane == 2 && J > 0) word = G.prog + ((J - 1) * G.TI + I), need = uk;
    if (lane == 3 && I + 1 < G.TI) word = G.prog + (J * G.TI + I + 1), need = uk - 1;
    if (lane == 4 && J + 1 < G.TJ) word = G.prog + ((J + 1) * G.TI + I), need = uk - 1;
    if (lane == 5 && I + 1 == G.TI && J + 1 == G.TJ) word = G.prog + (J * G.TI + I), need = uk - 1;
    bool go = false;
    for (unsigned spins = 0;; spins++) {
      const unsigned x = ld_word(word);
      if (__any(lane == 6 && x != 0)) break;  // a final sweep was found, or an error
      if (__all(x >= need)) {
        go = true;
        break;
      }
      if (spins > SPIN_LIMIT) {
        if (lane == 0) {
          st_word(G.ctrl + C_ERR, (unsigned)n);
          st_word(G.ctrl + C_FINAL, FINAL_ERROR);
        }
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (!go) return;
    TRACE_MARK(2);  // waits
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the unknown on the tile and its halo ---------------------------------------------------
    const double *oldc = k == 1 ? G.a : G.ring + (size_t)((k - 1) % G.R) * ncell;
    double *newc = G.ring + (size_t)(k % G.R) * ncell;
    // Element (c, rr): ghost cells never change, the caller's array holds them; left / lower neighbours
    // come from this sweep's copy, everything else from the previous sweep's.
    auto fetch = [&](int c, int rr) -> double {
      const size_t g = g00 + (size_t)rr * ldg + c;
      const bool side = c == 0 || c == w + 1, ends = rr == 0 || rr == h + 1;
      const double *west_src = I == 0 ? G.a : newc, *east_src = I + 1 == G.TI ? G.a : oldc;
      const double *south_src = J == 0 ? G.a : newc, *north_src = J + 1 == G.TJ ? G.a : oldc;
      const double *src = c == 0 ? west_src : c == w + 1 ? east_src : rr == 0 ? south_src : rr == h + 1 ? north_src : oldc;
      const double v = src[g];
      return (side && ends) ? 0.0 : v;  // the corners are not part of the stencil
    };
    constexpr int CHA = 34;  // a tile of 32 rows and its halo in one go
    for (int rr0 = 0; rr0 <= h + 1; rr0 += CHA) {
      double va[CHA];
#pragma unroll
      for (int u = 0; u < CHA; u++) va[u] = fetch(cl, min(rr0 + u, h + 1));
#pragma unroll
      for (int u = 0; u < CHA; u++) a[min(rr0 + u, h + 1) * LDT + cl] = va[u];
    }
    if (w + 1 >= TW) {
      const int c = TW + (lane & 1);
      for (int rr0 = 0; rr0 <= h + 1; rr0 += TW / 2) {
        const int rr = min(rr0 + (lane >> 1), h + 1);
        a[rr * LDT + c] = fetch(c, rr);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    TRACE_MARK(3);  // acquire + the unknown
    // ---- the sweep over the tile: lane = column, columns one step apart -------------------------
    // What a cell needs from the previous sweep (its own value, the values to the right and above) and
    // its coefficients are read one step ahead; the values of this sweep arrive in registers (the left
    // one from the lane below, the lower one is the lane's own previous result), so nothing that a step
    // waits for goes through memory.  A lane overwrites a cell one or two steps after its neighbours
    // have read the old value (cell (c, r) is written at step c + r; its old value is read at steps
    // c + r - 2 and c + r - 1).
    // The step is straight-line code (a lane outside its column's rows works on a clamped cell and keeps
    // nothing), so that the reads and quotients prepared for the next step fill the waits of this step's
    // chain of dependent operations.
    const int ci = lane;
    struct Cell {
      int q;
      double old, e, n, dw, ds, s1, s2, halo_w, halo_s;
    };
    auto prepare = [&](int s) -> Cell {
      Cell c;
      const int r = min(max(s - ci, 0), h - 1);
      const int q = (r + 1) * LDT + min(ci, w - 1) + 1;
      c.q = q;
      c.old = a[q];
      const double east = a[q + 1], north = a[q + LDT];
      c.halo_w = a[q - 1];     // used by lane 0 only
      c.halo_s = a[q - LDT];   // used in row 0 only
      if (MODE == 0) {
        const double rc = rt[q], re = rt[q + 1], rw = rt[q - 1], rn = rt[q + LDT], rs = rt[q - LDT];
        c.e = east / (dx * (re + rc));
        c.n = north / (dy * (rn + rc));
        c.dw = dx * (rw + rc);
        c.ds = dy * (rs + rc);
        c.s1 = k1[q];
        c.s2 = k2[q];
      } else {
        c.e = east;
        c.n = north;
        c.dw = k1[q - 1];
        c.ds = k2[q - LDT];
        c.s1 = k1[q];
        c.s2 = k2[q];
      }
      return c;
    };
    double prev = 0.0;  // the value this lane produced in the previous step
    u64 mymax = 0;
    const int steps = w + h - 1;
    Cell nxt = prepare(0);
    for (int s = 0; s < steps; s++) {
      const Cell c = nxt;
      const double from_west = from_lane_below(prev);
      nxt = prepare(s + 1);
      const int r = s - ci;
      const bool act = ci < w && r >= 0 && r < h;
      const double west = ci == 0 ? c.halo_w : from_west;
      const double south = r == 0 ? c.halo_s : prev;
      double nw;
      if (MODE == 0) {
        const double ex = c.e + west / c.dw;
        const double ey = c.n + south / c.ds;
        nw = (1.0 - beta) * c.old + c.s2 * ((1.0 / dx) * ex + (1.0 / dy) * ey - c.s1);
      } else {
        nw = (1.0 - beta) * c.old + beta * (0.25 * (c.e + west + c.n + south + c.dw - c.s1 + c.ds - c.s2));
      }
      if (act) a[c.q] = nw;
      prev = act ? nw : prev;
      const u64 b = (u64)__double_as_longlong(fabs(c.old - nw));
      mymax = (act && b > mymax) ? b : mymax;
      // (for the compiler: the LDS accesses of a step stay in that step -- lanes read what other lanes write)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    TRACE_MARK(4);  // the sweep
#ifdef NSK_SOR_GRID_TRACE
    if (lane == 0) tr[7] += 1;
#endif
    // ---- the tile's new values, write-through, then its progress word ---------------------------
    for (int rr = 1; rr <= h; rr++)
      if (lane < w) {
        const size_t g = g00 + (size_t)rr * ldg + lane + 1;
        __hip_atomic_store(newc + g, a[rr * LDT + lane + 1], RLX, AGENT);
      }
    for (int off = 32; off > 0; off >>= 1) {
      const u64 o = __shfl_xor(mymax, off, 64);
      mymax = o > mymax ? o : mymax;
    }
    const bool last_tile = I + 1 == G.TI && J + 1 == G.TJ;
    u64 sweep = mymax;
    if (lane == 0) {
      if (last_tile) {
        const u64 before = __hip_atomic_fetch_max(G.smax + (k % G.R), mymax, RLX, AGENT);
        sweep = before > mymax ? before : mymax;
      } else {
        __hip_atomic_fetch_max(G.smax + (k % G.R), mymax, RLX, AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every lane's stores and the maximum have left
    TRACE_MARK(5);  // stores, maximum, drain
    if (lane == 0) st_word(G.prog + (J * G.TI + I), uk);
    if (last_tile && lane == 0) {
      // The upper right tile is the last task of its sweep, and every other task of the sweep lies on a
      // chain of progress words that ends here: their maxima are in.  false for NaN: keeps sweeping,
      // as the reference does.
      const double ch = __longlong_as_double((long long)sweep);
      if (ch < G.maxError || k == G.maxit) {
        __hip_atomic_store(G.smax + G.R, sweep, RLX, AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_word(G.ctrl + C_FINAL, uk);
      } else {
        __hip_atomic_store(G.smax + (k % G.R), (u64)0, RLX, AGENT);  // for sweep k + R
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_word(G.ctrl + C_JUDGED, uk);
      }
    }
    __builtin_amdgcn_wave_barrier();
    TRACE_MARK(6);  // flag, judgement
  }
}

// Copies the final sweep's values into the caller's array and hands out the counters.
__global__ void k_sor_grid_finish(GridSor G, int *sweeps_out, double *change_out) {
  const unsigned kf = G.ctrl[C_FINAL];
  const bool ok = kf != 0 && kf != FINAL_ERROR;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (sweeps_out) *sweeps_out = ok ? (int)kf : -1;
    if (change_out) *change_out = ok ? __longlong_as_double((long long)G.smax[G.R]) : 0.0;
  }
  if (!ok) return;
  const int ldg = G.nx + 2;
  const double *fin = G.ring + (size_t)(kf % (unsigned)G.R) * (size_t)ldg * (G.ny + 2);
  const size_t n = (size_t)G.nx * G.ny;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (size_t)gridDim.x * blockDim.x) {
    const size_t j = t / G.nx, i = t - j * G.nx;
    const size_t g = (j + 1) * ldg + i + 1;
    G.a[g] = fin[g];
  }
}

}  // namespace

// Launches the solve; the sweep count and the last change land in the device words d_sweeps / d_change.
int nsk_sor_grid(nsk_ctx *ctx, int mode, double *a, const double *c1, const double *c2, const double *c3,
                 double beta, double dx, double dy, int nx, int ny, int maxit, double maxError, int *d_sweeps,
                 double *d_change) {
  GridSor G{};
  G.a = a, G.c1 = c1, G.c2 = c2, G.c3 = c3;
  G.beta = beta, G.dx = dx, G.dy = dy, G.maxError = maxError;
  G.nx = nx, G.ny = ny, G.maxit = maxit;
  int BJ = ctx->sor_grid_rows > 0 ? ctx->sor_grid_rows : 32;
  if (BJ > 64) BJ = 64;
  if (BJ < 4) BJ = 4;
  G.BJ = BJ;
  G.TI = (nx + TW - 1) / TW, G.TJ = (ny + BJ - 1) / BJ;
  NSK_REQUIRE(G.TI < 32768 && G.TJ < 32768, "SOR: grid of %d x %d tiles", G.TI, G.TJ);
  G.NT = G.TI * G.TJ;
  const int D = G.TI + G.TJ - 2;
  // copies of the unknown: the sweeps in flight (D/2 + 1) and two more, within 2 GB
  const size_t ncell = (size_t)(nx + 2) * (ny + 2);
  long long R = D / 2 + 3;
  const long long fit = (long long)((2ull << 30) / (ncell * sizeof(double)));
  if (R > fit) R = fit;
  if (R < 3) R = 3;
  if (ctx->sor_grid_copies >= 3) R = ctx->sor_grid_copies;  // (testing: fewer copies than sweeps in flight)
  if (R > maxit + 1) R = maxit + 1;
  if (R < 3) R = 3;
  G.R = (int)R;
  // With fewer copies than sweeps the grid can hold in flight, sweep k + R has to wait for the judgement of
  // sweep k, which ends D/2 rounds after the sweep started: the sweeps then start in batches of R, a
  // batch every P rounds, so that also this wait is for a task with a smaller number.
  G.P = R > D / 2 ? (int)R : D / 2 + 1;
  if (ctx->sor_snap_elems < ncell * (size_t)R) {
    if (ctx->sor_snap) NSK_HIP(hipFree(ctx->sor_snap));
    ctx->sor_snap = nullptr;
    ctx->sor_snap_elems = 0;
    NSK_HIP(hipMalloc(&ctx->sor_snap, ncell * (size_t)R * sizeof(double)));
    ctx->sor_snap_elems = ncell * (size_t)R;
  }
  G.ring = ctx->sor_snap;
  const long long rounds = (long long)maxit + (long long)((maxit - 1) / R) * (G.P - R) + D / 2;
  NSK_REQUIRE(rounds * G.NT < 0x7fffffffLL, "SOR: %lld tasks do not fit the task counter", rounds * G.NT);
  G.total = (int)(rounds * G.NT);
  // control block: [ctrl | prog | smax] zeroed per call, then the tile order
  const size_t zero_words = ((size_t)C_WORDS + G.NT + 2 * ((size_t)R + 1) + 3) / 4 * 4;
#ifdef NSK_SOR_GRID_TRACE
  const size_t need = zero_words + (size_t)G.NT + 2 + 16 * 4096;
#else
  const size_t need = zero_words + (size_t)G.NT;
#endif
  if (ctx->sor_grid_words < need) {
    if (ctx->sor_grid_ctl) NSK_HIP(hipFree(ctx->sor_grid_ctl));
    ctx->sor_grid_ctl = nullptr;
    ctx->sor_grid_words = 0;
    ctx->sor_grid_ti = ctx->sor_grid_tj = 0;
    NSK_HIP(hipMalloc(&ctx->sor_grid_ctl, need * sizeof(unsigned)));
    ctx->sor_grid_words = need;
  }
  unsigned *base = ctx->sor_grid_ctl;
  G.ctrl = base;
  G.smax = reinterpret_cast<u64 *>(base + C_WORDS);  // 8-byte aligned: C_WORDS is even
  G.prog = base + C_WORDS + 2 * ((size_t)R + 1);
  int *tiles = reinterpret_cast<int *>(base + zero_words);
  if (ctx->sor_grid_ti != G.TI || ctx->sor_grid_tj != G.TJ || ctx->sor_grid_zero != zero_words) {
    // tiles with I + J even, then the others (a task waits only for tasks of the other parity)
    std::vector<int> order;
    order.reserve(G.NT);
    for (int par = 0; par < 2; par++)
      for (int J = 0; J < G.TJ; J++)
        for (int I = 0; I < G.TI; I++)
          if (((I + J) & 1) == par) order.push_back(I | (J << 16));
    NSK_HIP(hipStreamSynchronize(ctx->stream));  // an earlier solve may still read the old table
    NSK_HIP(hipMemcpy(tiles, order.data(), order.size() * sizeof(int), hipMemcpyHostToDevice));
    ctx->sor_grid_ti = G.TI, ctx->sor_grid_tj = G.TJ, ctx->sor_grid_zero = zero_words;
  }
  G.tiles = tiles;
  NSK_HIP(hipMemsetAsync(base, 0, zero_words * sizeof(unsigned), ctx->stream));
  const size_t lds_bytes = (size_t)(mode == 0 ? 4 : 3) * (BJ + 2) * LDT * sizeof(double);
  // as many wavefronts as the chip holds at this LDS size, and no more than half the tiles can run at once anyway
  const int cus = ctx->num_simds / 4;
  int per_cu = (int)((160 * 1024) / lds_bytes);
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int blocks = cus * per_cu;
  if (blocks > G.NT) blocks = G.NT;
  if (mode == 0) {
    NSK_HIP(nsk_allow_lds(ctx, (const void *)k_sor_grid<0>, lds_bytes));
    k_sor_grid<0><<<blocks, 64, lds_bytes, ctx->stream>>>(G);
  } else {
    NSK_HIP(nsk_allow_lds(ctx, (const void *)k_sor_grid<1>, lds_bytes));
    k_sor_grid<1><<<blocks, 64, lds_bytes, ctx->stream>>>(G);
  }
  NSK_LAUNCH_CHECK();
  const size_t n = (size_t)nx * ny;
  int fb = (int)((n + 255) / 256);
  if (fb > 2048) fb = 2048;
  k_sor_grid_finish<<<fb, 256, 0, ctx->stream>>>(G, d_sweeps, d_change);
  NSK_LAUNCH_CHECK();
  return 0;
}

#ifdef NSK_SOR_GRID_TRACE
// debug builds only: the per-workgroup phase sums of the last solve (8 words per workgroup)
extern "C" int nsk_sor_grid_trace(nsk_ctx *ctx, unsigned long long *out, int max_blocks) {
  NSK_CTX(ctx);
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  const size_t nt = (size_t)ctx->sor_grid_ti * ctx->sor_grid_tj;
  const unsigned *base = ctx->sor_grid_ctl + ctx->sor_grid_zero + nt + (nt & 1);
  NSK_HIP(hipMemcpy(out, base, (size_t)max_blocks * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return 0;
}
#endif
