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

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ unsigned ld_word(unsigned *p) { return __hip_atomic_load(p, RLX, AGENT); }
__device__ __forceinline__ void st_word(unsigned *p, unsigned v) { __hip_atomic_store(p, v, RLX, AGENT); }

// Waits until *w >= v.  False: the solve has ended (a final sweep was found, or an error).
__device__ __forceinline__ bool wait_ge(unsigned *w, unsigned v, unsigned *ctrl, unsigned code) {
  for (unsigned spins = 0;; spins++) {
    if (ld_word(w) >= v) return true;
    if (ld_word(ctrl + C_FINAL) != 0) return false;
    if (spins > SPIN_LIMIT) {
      st_word(ctrl + C_ERR, code);
      st_word(ctrl + C_FINAL, FINAL_ERROR);
      return false;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}

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
  for (;;) {
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
    // ---- what this task waits for ---------------------------------------------------------------
    const unsigned uk = (unsigned)k;
    bool go = true;
    if (I == 0 && J == 0 && k > G.R) go = wait_ge(G.ctrl + C_JUDGED, uk - (unsigned)G.R, G.ctrl, 1);
    if (go && I > 0) go = wait_ge(G.prog + (J * G.TI + I - 1), uk, G.ctrl, 2);
    if (go && J > 0) go = wait_ge(G.prog + ((J - 1) * G.TI + I), uk, G.ctrl, 3);
    if (go && I + 1 < G.TI) go = wait_ge(G.prog + (J * G.TI + I + 1), uk - 1, G.ctrl, 4);
    if (go && J + 1 < G.TJ) go = wait_ge(G.prog + ((J + 1) * G.TI + I), uk - 1, G.ctrl, 5);
    if (go && I + 1 == G.TI && J + 1 == G.TJ) go = wait_ge(G.prog + (J * G.TI + I), uk - 1, G.ctrl, 6);
    if (!go) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the tile with its halo into LDS --------------------------------------------------------
    const int w = min(TW, G.nx - TW * I), h = min(G.BJ, G.ny - G.BJ * J);
    const double *oldc = k == 1 ? G.a : G.ring + (size_t)((k - 1) % G.R) * ncell;
    double *newc = G.ring + (size_t)(k % G.R) * ncell;
    // 0-based global index of tile element (c, rr), c = 0..w+1, rr = 0..h+1 (halo included)
    const size_t g00 = (size_t)(G.BJ * J) * ldg + (size_t)(TW * I);
    // Element (c, rr): ghost cells never change, the caller's array holds them; left / lower neighbours
    // come from this sweep's copy, everything else from the previous sweep's.
    auto fetch = [&](int c, int rr, double &va, double &v1, double &v2, double &v3) {
      const size_t g = g00 + (size_t)rr * ldg + c;
      const bool side = c == 0 || c == w + 1, ends = rr == 0 || rr == h + 1;
      const double *src = oldc;
      if (c == 0) src = I == 0 ? G.a : newc;
      else if (c == w + 1) src = I + 1 == G.TI ? G.a : oldc;
      else if (rr == 0) src = J == 0 ? G.a : newc;
      else if (rr == h + 1) src = J + 1 == G.TJ ? G.a : oldc;
      va = (side && ends) ? 0.0 : src[g];  // the corners are not part of the stencil
      v1 = G.c1[g];
      v2 = G.c2[g];
      v3 = MODE == 0 ? G.c3[g] : 0.0;
    };
    auto keep = [&](int c, int rr, double va, double v1, double v2, double v3) {
      const int t = rr * LDT + c;
      a[t] = va;
      if (MODE == 0) {
        k1[t] = v1;
        k2[t] = beta * v2;
        rt[t] = v3;
      } else {
        k1[t] = dx * v1;
        k2[t] = dy * v2;
      }
    };
    constexpr int CH = 8;  // rows requested together, so that their latencies overlap
    for (int rr0 = 0; rr0 <= h + 1; rr0 += CH) {
      double va[CH], v1[CH], v2[CH], v3[CH];
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (rr0 + u <= h + 1 && lane <= w + 1) fetch(lane, rr0 + u, va[u], v1[u], v2[u], v3[u]);
#pragma unroll
      for (int u = 0; u < CH; u++)
        if (rr0 + u <= h + 1 && lane <= w + 1) keep(lane, rr0 + u, va[u], v1[u], v2[u], v3[u]);
    }
    // columns 64 and 65 of a full-width tile: a lane per row
    for (int c = TW; c <= w + 1; c++)
      for (int rr = lane; rr <= h + 1; rr += TW) {
        double va, v1, v2, v3;
        fetch(c, rr, va, v1, v2, v3);
        keep(c, rr, va, v1, v2, v3);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- the sweep over the tile: lane = column, columns one step apart -------------------------
    const int ci = lane;
    double prev = 0.0;  // the value this lane produced in the previous step
    u64 mymax = 0;
    const int steps = w + h - 1;
    for (int s = 0; s < steps; s++) {
      const int r = s - ci;
      const double from_west = from_lane_below(prev);
      if (ci < w && r >= 0 && r < h) {
        const int q = (r + 1) * LDT + ci + 1;
        const double old = a[q], east = a[q + 1], north = a[q + LDT];
        const double west = ci == 0 ? a[q - 1] : from_west;
        const double south = r == 0 ? a[q - LDT] : prev;
        double nw;
        if (MODE == 0) {
          const double rc = rt[q], re = rt[q + 1], rw = rt[q - 1], rn = rt[q + LDT], rs = rt[q - LDT];
          const double ex = east / (dx * (re + rc)) + west / (dx * (rw + rc));
          const double ey = north / (dy * (rn + rc)) + south / (dy * (rs + rc));
          nw = (1.0 - beta) * old + k2[q] * ((1.0 / dx) * ex + (1.0 / dy) * ey - k1[q]);
        } else {
          nw = (1.0 - beta) * old + beta * (0.25 * (east + west + north + south + k1[q - 1] - k1[q] + k2[q - LDT] - k2[q]));
        }
        a[q] = nw;
        prev = nw;
        const u64 b = (u64)__double_as_longlong(fabs(old - nw));
        mymax = b > mymax ? b : mymax;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
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
  const size_t need = zero_words + (size_t)G.NT;
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
