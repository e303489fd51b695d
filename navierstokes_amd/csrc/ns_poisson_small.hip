// The Poisson passes (scripts/NavierStokes.jl:85-87) on grids whose fields fit on chip, as ONE launch
// per run of passes instead of one launch per K passes: at the reference's default 153 x 255 a launch
// of the multi-pass kernel carries 2-3 us of work under 5 us of launch cost.
//
// The grid is cut into tiles, one workgroup each.  A workgroup keeps the pressure of its tile and of a
// frame of H halo cells around it in LDS (two copies, read one / write the other) and dPrdτ and ρ/dt ∇V
// of the same cells in registers, one cell per thread and slot.  A pass is one sweep over the cells that
// are still valid (the frame shrinks by a cell per pass, as the halo rows of the multi-pass kernel do)
// and one workgroup barrier.  Every H passes the workgroups hand the cells of their tiles that lie within
// H of a neighbour to those neighbours (up to eight) through memory: write-through stores, a drain, one
// progress word per workgroup; the neighbour polls the words, runs one agent-scope acquire and loads.
// Two generations of exchange cells alternate, so nobody overwrites cells a neighbour may still read.
// Every value is formed by the reference's expression (association as in update_pair of the multi-pass
// kernel), the boundary conditions in the reference's order (bc_x! on its row range, then bc_yval!),
// so the results equal `count` single passes bit for bit.
//
// Division: the exact two-operation form of the multi-pass kernel where it holds for dx and dy; it
// needs pressure values that are 0 or >= 2^-848 and finite.  The first launch writes into the
// context's partner arrays and raises a flag when a value left that range; the second launch copies
// the result home -- or, with the flag up, runs all passes again from the untouched inputs with IEEE
// division.  No host round trip in between.
//
// All workgroups of a launch wait for each other, so they must be resident together: there are at most
// as many as compute units.  Every spin is bounded; one that runs out raises an error word that the
// next synchronising call of the context reports.
#include <map>
#include <mutex>

#include "ns_poisson_kernel.h"

namespace nsk_poisson {

namespace {

constexpr int NT = 1024;         // threads per workgroup
constexpr int MAXM = 6;          // cell slots per thread (the register budget of 1024 threads is 128 each: eight slots spill)
constexpr unsigned SPIN_LIMIT = 1u << 22;
enum { F_BAD = 0, F_ERR = 1, F_WORDS = 4 };  // words behind the progress words

struct SmallArgs {
  const double *pr_in, *dp_in, *divv;
  double *pr_out, *dp_out;
  double *xbuf;      // exchange cells: [generation 0/1][Pr / dPrdτ][ny][ld]
  unsigned *words;   // [G] stages completed per workgroup, then F_WORDS flags
  unsigned long long *err_host;  // pinned word the host looks at after a synchronisation
  int nx, ny, ld, ldd, dp_embedded, xbc_rows;
  int TC, TR, H, GX, GY, count;
  int trace_at;      // debug builds: where the phase sums start in `words`
  unsigned spin_limit;
  int test_absent;   // testing: workgroup test_absent - 1 leaves at once, as if it had never been scheduled
  int mode;          // 0: compute; 1: copy in -> out unless the range flag is up, then compute in place
  double dx, dy, rho_dt, keep, dtau, topval, rdx_h, rdx_l, rdy_h, rdy_l;
};

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

// Debug builds (-DNSK_SMALL_TRACE): every workgroup sums the time it spends in each phase (100 MHz clock)
// into eight words of its own behind the flags; tools/trace_small.py prints the averages.
#ifdef NSK_SMALL_TRACE
#define TRACE_MARK(i)                                                                                          \
  do {                                                                                                         \
    const unsigned long long now_ = wall_clock64();                                                            \
    if (tid == 0) reinterpret_cast<unsigned long long *>(A.words + A.trace_at)[8 * g + (i)] += now_ - tlast;   \
    tlast = now_;                                                                                              \
  } while (0)
#else
#define TRACE_MARK(i) \
  do {                \
  } while (0)
#endif

// what kind of cell a slot holds
enum {
  HAVE = 1,       // a cell of the kept region
  INNER = 2,      // not on the rim of the grid: the iteration updates it
  OWNED = 4,      // a cell of this workgroup's tile
  EDGE = 8,       // owned and within H of a neighbouring tile: handed over at every exchange
  C_FIRST = 16,   // column 1: writes column 0
  C_LAST = 32,    // column nx-2: writes column nx-1
  R_FIRST = 64,   // row 1: writes row 0
  R_LAST = 128,   // row ny-2: writes row ny-1
  XBC = 256,      // a row that bc_x! reaches
  BC_ANY = C_FIRST | C_LAST | R_FIRST | R_LAST
};

template <int DIV, int M>
__global__ void __launch_bounds__(NT) k_poisson_small(SmallArgs A) {
  extern __shared__ __attribute__((aligned(16))) char lds[];  // copy 0 | copy 1 | one flag word
  constexpr int BUF = NT * M * 8;                             // bytes from one LDS copy to the other
  const int tid = threadIdx.x, g = blockIdx.x;
  const int nx = A.nx, ny = A.ny, ld = A.ld, H = A.H;
  const int gx = g % A.GX, gy = g / A.GX;
  const int oc_lo = gx * A.TC, oc_hi = min(oc_lo + A.TC, nx), or_lo = gy * A.TR, or_hi = min(or_lo + A.TR, ny);  // the tile
  const int ec_lo = max(oc_lo - H, 0), ec_hi = min(oc_hi + H, nx), er_lo = max(or_lo - H, 0), er_hi = min(or_hi + H, ny);
  const int W = ec_hi - ec_lo, ER = er_hi - er_lo;  // the kept region
  const int G = A.GX * A.GY;
#ifdef NSK_SMALL_TRACE
  unsigned long long tlast = wall_clock64();
#endif
  auto dp_index = [&](int c, int r) -> int { return A.dp_embedded ? r * A.ldd + c : (r - 1) * A.ldd + (c - 1); };
  auto at = [&](int off) -> double & { return *reinterpret_cast<double *>(lds + off); };
  if (A.mode == 1 && __hip_atomic_load(A.words + G + F_BAD, RLX, AGENT) == 0) {
    // the first launch stayed inside the range of its division form: its result goes home
    const int OW = oc_hi - oc_lo;
    for (int e = tid; e < (or_hi - or_lo) * OW; e += NT) {
      const int r = or_lo + e / OW, c = oc_lo + e % OW;
      A.pr_out[(size_t)r * ld + c] = A.pr_in[(size_t)r * ld + c];
      if (c >= 1 && c <= nx - 2 && r >= 1 && r <= ny - 2) A.dp_out[dp_index(c, r)] = A.dp_in[dp_index(c, r)];
    }
    return;
  }
  if (g == A.test_absent - 1) return;
  // ---- meeting the neighbours -------------------------------------------------------------------------
  // words[g] counts the stages workgroup g has completed: 1 = has read its inputs, i + 2 = has handed
  // over its cells of exchange i.
  double *flag = reinterpret_cast<double *>(lds + 2 * BUF);
  auto publish = [&](unsigned stage) {  // (all threads) everything stored so far, then the word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(A.words + g, stage, RLX, AGENT);
  };
  auto meet = [&](unsigned stage) -> bool {  // (all threads) all neighbours have completed `stage`
    if (tid < 64) {                          // lanes 0..8 of the first wavefront watch one neighbour each
      int nb = -1;
      if (tid < 9 && tid != 4) {
        const int x = gx + tid % 3 - 1, y = gy + tid / 3 - 1;
        if (x >= 0 && x < A.GX && y >= 0 && y < A.GY) nb = y * A.GX + x;
      }
      bool ok = true;
      for (unsigned spins = 0;; spins++) {
        const bool there = nb < 0 || __hip_atomic_load(A.words + nb, RLX, AGENT) >= stage;
        if (__all(there)) break;
        if (spins > A.spin_limit || ((spins & 31) == 31 && __hip_atomic_load(A.words + G + F_ERR, RLX, AGENT) != 0)) {
          ok = false;
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
      // One acquire per workgroup: the vector cache it empties is the compute unit's, which all wavefronts of
      // the workgroup share, and they load only after the barrier below.  (An acquire in every wavefront
      // cost 18 us per exchange on 209 workgroups: the invalidations queue up per L2.)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) {
        if (!ok) {  // a neighbour never arrived: everybody gives up, the host is told
          __hip_atomic_store(A.words + G + F_ERR, 1u, RLX, AGENT);
          *A.err_host = 1;
        }
        *flag = ok ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    return *flag != 0.0;
  };
  // In mode 1 the passes run in place on the caller's arrays (the first launch did not touch them);
  // in place is safe because a workgroup stores its cells only after its neighbours -- the only
  // readers of those cells -- have completed stage 1.
  const double *pr_src = A.mode == 1 ? A.pr_out : A.pr_in, *dp_src = A.mode == 1 ? A.dp_out : A.dp_in;
  // ---- this thread's cells ------------------------------------------------------------------------------
  // ac: byte offset in an LDS copy of the cell's left neighbour (the cell below / above: see dn / up); gi: the
  // cell in the global arrays; df = 1024 * dep + kind, dep = the last pass of a block in which the cell is
  // still valid (its distance from the frame's rim; no limit towards the rim of the grid; 0 = never computed).
  // With more than four slots the offsets of the cells below and above are formed when needed (two additions
  // per cell and pass) instead of kept: 128 registers per thread have to do.
  constexpr bool KEEP = M <= 4;
  int ac[M], adn[KEEP ? M : 1], aup[KEEP ? M : 1], gi[M], df[M];
  double dP[M], gg[M];
  const int below = (1 - W) * 8, above = (1 + W) * 8;
  auto dn = [&](int m) { return KEEP ? adn[m] : ac[m] + below; };
  auto up = [&](int m) { return KEEP ? aup[m] : ac[m] + above; };
  auto dp_of = [&](int g_index) { return dp_index(g_index % ld, g_index / ld); };
  int emin = 0;
  bool nonfin = false;
  auto note = [&](double x) {
    if (DIV != DIV_IEEE) {
      emin = min(emin, __builtin_amdgcn_frexp_exp(x));
      nonfin |= non_finite(x);
    }
  };
  constexpr int FAR = 1 << 20;
#pragma unroll
  for (int m = 0; m < M; m++) {
    const int e = tid + NT * m;
    const bool have = e < ER * W;
    const int l = have ? e / W : 0, cl = have ? e - l * W : 0, c = ec_lo + cl, r = er_lo + l;
    const bool inner = have && c >= 1 && c <= nx - 2 && r >= 1 && r <= ny - 2;
    const bool owned = have && c >= oc_lo && c < oc_hi && r >= or_lo && r < or_hi;
    const bool edge = owned && ((gx > 0 && c < oc_lo + H) || (gx + 1 < A.GX && c >= oc_hi - H) || (gy > 0 && r < or_lo + H) ||
                                (gy + 1 < A.GY && r >= or_hi - H));
    const int q = l * W + cl;
    ac[m] = (q - 1) * 8;
    if (KEEP) adn[m] = (q - W) * 8, aup[m] = (q + W) * 8;
    gi[m] = have ? r * ld + c : 0;
    const int d = min(min(ec_lo > 0 ? cl : FAR, ec_hi < nx ? W - 1 - cl : FAR), min(er_lo > 0 ? l : FAR, er_hi < ny ? ER - 1 - l : FAR));
    const int kind = (have ? HAVE : 0) | (inner ? INNER : 0) | (owned ? OWNED : 0) | (edge ? EDGE : 0) | (inner && c == 1 ? C_FIRST : 0) |
            (inner && c == nx - 2 ? C_LAST : 0) | (inner && r == 1 ? R_FIRST : 0) | (inner && r == ny - 2 ? R_LAST : 0) |
            (r < A.xbc_rows ? XBC : 0);
    df[m] = (inner ? d : 0) * 1024 + kind;
    dP[m] = gg[m] = 0.0;
    if (have) {
      const double v = pr_src[gi[m]];
      at(ac[m] + 8) = v;
      note(v);
    }
    if (inner) {
      dP[m] = dp_src[dp_index(c, r)];
      gg[m] = A.rho_dt * A.divv[gi[m]];
    }
  }
  if (G > 1) publish(1);
  __syncthreads();
  TRACE_MARK(0);  // inputs

  // One pass: reads the copy at RD, writes the copy at WR.
  auto pass = [&](int s, const int RD, const int WR) {
#pragma unroll
    for (int m = 0; m < M; m++) {
      if (s * 1024 <= df[m]) {  // (s <= dep: the kind bits are below 1024)
        const int k = ac[m];
        const double lft = at(RD + k), cen = at(RD + k + 8), rgt = at(RD + k + 16), low = at(RD + dn(m)), upp = at(RD + up(m));
        const double d2x = (rgt - cen) - (cen - lft);
        const double d2y = (upp - cen) - (cen - low);
        const double qx = div_const<DIV>(div_const<DIV>(d2x, A.dx, A.rdx_h, A.rdx_l), A.dx, A.rdx_h, A.rdx_l);
        const double qy = div_const<DIV>(div_const<DIV>(d2y, A.dy, A.rdy_h, A.rdy_l), A.dy, A.rdy_h, A.rdy_l);
        const double L = qx + qy - gg[m];
        const double dnew = dP[m] * A.keep + A.dtau * L;
        const double pnew = cen + A.dtau * dnew;
        dP[m] = dnew;
        note(pnew);
        at(WR + k + 8) = pnew;
        // set_bc_Pr!: bc_x! (rows below its limit take their inner neighbour's new value, the others
        // keep theirs), then bc_yval! (bottom row = the row above it as bc_x! left it, top row = val)
        const int f = df[m];
        if (f & BC_ANY) {
          const double left = (f & XBC) ? pnew : lft, right = (f & XBC) ? pnew : rgt;
          if (f & C_FIRST) at(WR + k) = left;
          if (f & C_LAST) at(WR + k + 16) = right;
          if (f & R_FIRST) {
            at(WR + dn(m)) = pnew;
            if (f & C_FIRST) at(WR + dn(m) - 8) = left;
            if (f & C_LAST) at(WR + dn(m) + 8) = right;
          }
          if (f & R_LAST) {
            at(WR + up(m)) = A.topval;
            if (f & C_FIRST) at(WR + up(m) - 8) = A.topval;
            if (f & C_LAST) at(WR + up(m) + 8) = A.topval;
          }
        }
      }
    }
    __syncthreads();
  };

  int cur = 0, done = 0;
  unsigned exchanges = 0;
  while (done < A.count) {
    const int hb = min(H, A.count - done);
    for (int s = 1; s <= hb; s++) {
      if (cur == 0) pass(s, 0, BUF);
      else pass(s, BUF, 0);
      cur ^= 1;
    }
    done += hb;
    TRACE_MARK(1);  // passes
    if (done >= A.count || G == 1) continue;
    // ---- hand the tile's outer cells to the neighbours, take theirs -------------------------------------------
    // Two generations of exchange cells alternate: a workgroup writes generation i + 2 after it has met its
    // neighbours at stage i + 3, which they complete after reading generation i.
    double *x_pr = A.xbuf + (size_t)(exchanges & 1) * 2 * (size_t)ny * ld, *x_dp = x_pr + (size_t)ny * ld;
    const int CUR = cur ? BUF : 0;
#pragma unroll
    for (int m = 0; m < M; m++) {
      if (df[m] & EDGE) {
        __hip_atomic_store(x_pr + gi[m], at(CUR + ac[m] + 8), RLX, AGENT);
        if (df[m] & INNER) __hip_atomic_store(x_dp + gi[m], dP[m], RLX, AGENT);
      }
    }
    publish(exchanges + 2);
    TRACE_MARK(2);  // outer cells stored and drained, word out
    if (!meet(exchanges + 2)) return;
    TRACE_MARK(3);  // neighbours arrived
    exchanges++;
    // (all requests first, then the selection: a slot that holds no halo cell reads a cell of its own)
    double hp[M], hd[M];
#pragma unroll
    for (int m = 0; m < M; m++) {
      hp[m] = x_pr[gi[m]];
      hd[m] = x_dp[gi[m]];
    }
#pragma unroll
    for (int m = 0; m < M; m++) {
      if ((df[m] & (HAVE | OWNED)) == HAVE) {
        at(CUR + ac[m] + 8) = hp[m];
        note(hp[m]);
        if (df[m] & INNER) dP[m] = hd[m];
      }
    }
    __syncthreads();
    TRACE_MARK(4);  // halo cells in
  }
  // ---- the tile goes out (in place: not before the neighbours have read their inputs) -------------------------
  if (G > 1 && exchanges == 0 && !meet(1)) return;
  const int CUR = cur ? BUF : 0;
#pragma unroll
  for (int m = 0; m < M; m++) {
    if (df[m] & OWNED) {
      A.pr_out[gi[m]] = at(CUR + ac[m] + 8);
      if (df[m] & INNER) A.dp_out[dp_of(gi[m])] = dP[m];
    }
  }
  if (DIV != DIV_IEEE && (emin < E_MIN_OK || nonfin)) __hip_atomic_store(A.words + G + F_BAD, 1u, RLX, AGENT);
  TRACE_MARK(5);  // results out
}

template <int DIV, int M>
int launch_one(nsk_ctx *ctx, const SmallArgs &A) {
  const size_t lds_bytes = 2 * (size_t)NT * M * 8 + 16;
  NSK_HIP(nsk_allow_lds(ctx, (const void *)k_poisson_small<DIV, M>, lds_bytes));
  k_poisson_small<DIV, M><<<A.GX * A.GY, NT, lds_bytes, ctx->stream>>>(A);
  NSK_LAUNCH_CHECK();
  return 0;
}
// Two of these launches side by side on one device could each hold a part of the compute units and wait for
// workgroups that find no room (until the poll limits end both with an error).  Within a process they are
// therefore chained: a launch on another stream than the last one's waits for that one's end.  (Launches of
// other processes on the same device cannot be seen from here.)
struct LastLaunch {
  hipEvent_t done = nullptr;
  hipStream_t stream = nullptr;
  bool any = false;
};
std::mutex g_chain_mu;
std::map<int, LastLaunch> g_chain;  // by device

template <int DIV>
int launch_small(nsk_ctx *ctx, const SmallArgs &A, int slots) {
  NSK_REQUIRE(slots >= 1 && slots <= MAXM, "small-grid Poisson launch: %d cell slots per thread (1..%d)", slots, MAXM);
  std::lock_guard<std::mutex> lock(g_chain_mu);
  LastLaunch &last = g_chain[ctx->device];
  if (!last.done) NSK_HIP(hipEventCreateWithFlags(&last.done, hipEventDisableTiming));
  if (last.any && last.stream != ctx->stream) NSK_HIP(hipStreamWaitEvent(ctx->stream, last.done, 0));
  int rc;
  if (slots == 1) rc = launch_one<DIV, 1>(ctx, A);
  else if (slots == 2) rc = launch_one<DIV, 2>(ctx, A);
  else if (slots <= 4) rc = launch_one<DIV, 4>(ctx, A);
  else rc = launch_one<DIV, 6>(ctx, A);
  if (rc) return rc;
  NSK_HIP(hipEventRecord(last.done, ctx->stream));
  last.stream = ctx->stream;
  last.any = true;
  return 0;
}

}  // namespace

// Is the one-launch form the one for this call, and with which tiles?  Whole domains only.
bool small_grid_applies(const nsk_ctx *ctx, const nsk_ns_params *p, int count, SmallPlan *plan) {
  if (!ctx->poisson_small || p->iy0 != 0 || p->ny != p->nyg || count < 12) return false;
  if (ctx->poisson_k != 0 || ctx->poisson_rows != 0) return false;  // the caller has pinned the launch-per-K-passes plan
  const int nx = p->nx, ny = p->ny;
  if (nx < 3 || ny < 3) return false;
  const int cus = ctx->num_simds / 4;
  // H passes between two exchanges, on tiles of at least H x H cells (so that halo cells come from the eight
  // nearest workgroups only): the cost of a hand-over (a few microseconds) is shared by H passes, the price
  // is a frame of H/2 more cells per pass on average.  The smallest tiles that cover the grid with no more
  // workgroups than compute units are taken: the passes are bound by the arithmetic of one compute unit.
  for (int h : {ctx->poisson_small_halo > 0 ? ctx->poisson_small_halo : 8, 6, 4}) {
    for (int t = h; t <= 4096; t += (t < 128 ? 2 : 16)) {
      int tc = ctx->poisson_small_cols > 0 ? ctx->poisson_small_cols : t;
      int tr = ctx->poisson_small_rows > 0 ? ctx->poisson_small_rows : t;
      int hh = h;
      tc = tc > nx ? nx : tc;
      tr = tr > ny ? ny : tr;
      hh = hh > tc ? tc : hh;
      hh = hh > tr ? tr : hh;
      const long gxn = (nx + tc - 1) / tc, gyn = (ny + tr - 1) / tr;
      const long w = (tc + 2 * hh < nx ? tc + 2 * hh : nx), er = (tr + 2 * hh < ny ? tr + 2 * hh : ny);
      if (w * er > (long)NT * MAXM) break;  // (larger tiles only get larger)
      // The rim of the grid is written by the cells next to it, so those must be cells of the same tile:
      // no tile, the ragged last one included, may be a single column or row at the rim.
      if ((gxn > 1 && (tc < 2 || nx - (gxn - 1) * tc < 2)) || (gyn > 1 && (tr < 2 || ny - (gyn - 1) * tr < 2))) continue;
      if (gxn * gyn > cus) continue;
      *plan = SmallPlan{tc, tr, hh, (int)gxn, (int)gyn, (int)((w * er + NT - 1) / NT)};
      return true;
    }
  }
  return false;
}

// `count` passes in one launch (two with the exact two-operation division: see the head of this file).
// part_pr / part_dp: arrays with the layout of Pr / dPrdtau that receive the first launch's result.
int small_grid_iters(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV, const nsk_ns_params *p, int count,
                     const PoissonArgs &P, int div, double *part_pr, double *part_dp, const SmallPlan &plan) {
  NSK_REQUIRE((size_t)P.ld * p->ny < (1u << 30), "small-grid Poisson launch: arrays of %d x %d values", P.ld, p->ny);
  SmallArgs A{};
  A.divv = divV;
  A.nx = p->nx, A.ny = p->ny, A.ld = P.ld, A.ldd = P.ldd, A.dp_embedded = P.dp_embedded, A.xbc_rows = P.xbc_rows;
  A.TC = plan.tile_cols, A.TR = plan.tile_rows, A.H = plan.halo, A.GX = plan.gx, A.GY = plan.gy, A.count = count;
  A.dx = P.dx, A.dy = P.dy, A.rho_dt = P.rho_dt, A.keep = P.keep, A.dtau = P.dtau, A.topval = P.topval;
  A.rdx_h = P.rdx_h, A.rdx_l = P.rdx_l, A.rdy_h = P.rdy_h, A.rdy_l = P.rdy_l;
  const int G = A.GX * A.GY;
  // exchange cells (two generations of Pr and dPrdτ) and the progress words
  const size_t xelems = 4 * (size_t)p->ny * P.ld;
  size_t nwords = ((size_t)G + F_WORDS + 3) / 4 * 4;
#ifdef NSK_SMALL_TRACE
  A.trace_at = (int)nwords;
  nwords += 16 * (size_t)G;
  ctx->small_trace_at = A.trace_at;
  ctx->small_trace_groups = G;
#endif
  if (ctx->small_xbuf_elems < xelems || ctx->small_words < nwords) {
    if (ctx->small_xbuf) NSK_HIP(hipFree(ctx->small_xbuf));
    ctx->small_xbuf = nullptr;
    ctx->small_xbuf_elems = ctx->small_words = 0;
    NSK_HIP(hipMalloc(&ctx->small_xbuf, xelems * sizeof(double) + nwords * sizeof(unsigned)));
    // (slots that hold no halo cell read exchange cells nobody has written: harmless, but defined)
    NSK_HIP(hipMemsetAsync(ctx->small_xbuf, 0, xelems * sizeof(double), ctx->stream));
    ctx->small_xbuf_elems = xelems;
    ctx->small_words = nwords;
  }
  A.xbuf = ctx->small_xbuf;
  A.words = reinterpret_cast<unsigned *>(ctx->small_xbuf + ctx->small_xbuf_elems);
  A.err_host = ctx->red_host + NSK_DEVICE_ERROR_WORD;
  A.spin_limit = ctx->poisson_small_spin_limit > 0 ? (unsigned)ctx->poisson_small_spin_limit : SPIN_LIMIT;
  A.test_absent = ctx->poisson_small_test_absent;
  const bool fma2 = div == DIV_FMA2;
  NSK_HIP(hipMemsetAsync(A.words, 0, nwords * sizeof(unsigned), ctx->stream));
  if (fma2) {
    A.pr_in = Pr, A.dp_in = dPrdtau, A.pr_out = part_pr, A.dp_out = part_dp, A.mode = 0;
    if (int rc = launch_small<DIV_FMA2>(ctx, A, plan.slots)) return rc;
    // home, or once more with IEEE division (progress words start again: the flags stay)
    NSK_HIP(hipMemsetAsync(A.words, 0, (size_t)G * sizeof(unsigned), ctx->stream));
    A.pr_in = part_pr, A.dp_in = part_dp, A.pr_out = Pr, A.dp_out = dPrdtau, A.mode = 1;
  } else {  // in place
    A.pr_in = Pr, A.dp_in = dPrdtau, A.pr_out = Pr, A.dp_out = dPrdtau, A.mode = 0;
  }
  return launch_small<DIV_IEEE>(ctx, A, plan.slots);
}

}  // namespace nsk_poisson

#ifdef NSK_SMALL_TRACE
// debug builds only: the per-workgroup phase sums of the last call (8 words per workgroup)
extern "C" int nsk_small_trace(nsk_ctx *ctx, unsigned long long *out, int max_groups, int *groups) {
  NSK_CTX(ctx);
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  const int n = ctx->small_trace_groups < max_groups ? ctx->small_trace_groups : max_groups;
  const unsigned *base = reinterpret_cast<unsigned *>(ctx->small_xbuf + ctx->small_xbuf_elems) + ctx->small_trace_at;
  NSK_HIP(hipMemcpy(out, base, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  *groups = n;
  return 0;
}
#endif
