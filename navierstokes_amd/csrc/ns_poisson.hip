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
// wave keeps two rows of pass s-1 and the running y-difference in registers, so
// the vertical neighbours are register reads and the horizontal neighbours of a
// lane's cell pair come from the adjacent lanes by a one-lane wavefront shift
// (DPP).  LDS only stages the wave's own input rows; no barriers: waves are
// independent.  The outer HK = K (rounded up to
// even) columns on each side of a strip and K rows at each end of a chunk are
// recomputed by the neighbouring task (halo); source and destination fields
// are distinct buffers (ping-pong), because a neighbouring task still reads
// pass-0 values that this task replaces.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "nsk_internal.h"

int nsk_launch_maxabs(nsk_ctx *ctx, const double *A, size_t count);
int nsk_read_maxabs(nsk_ctx *ctx, double *out);

#include "ns_poisson_kernel.h"
using namespace nsk_poisson;

namespace nsk_poisson {
#define NSK_DECL(K) extern template int launch_k<K>(nsk_ctx *, hipStream_t, PoissonArgs &, bool, int, const TaskPlan &);
NSK_DECL(1) NSK_DECL(2) NSK_DECL(3) NSK_DECL(4) NSK_DECL(5) NSK_DECL(6) NSK_DECL(7) NSK_DECL(8)
#undef NSK_DECL
}  // namespace nsk_poisson

// Ping-pong partners of Pr and dPrdτ for callers that cannot allocate (nsk_ns_poisson_iters_static):
// two partners of NSK_STATIC_ELEMS doubles each, 64 MiB per GPU, held until the process ends.
// In host code the array's name is only a handle for hipGetSymbolAddress; kernels and memsets use
// the per-device address looked up once by nsk_i_static_scratch_init.
constexpr size_t NSK_STATIC_ELEMS = (size_t)4 << 20;
static __device__ __align__(256) double scratch[2 * NSK_STATIC_ELEMS];

namespace {
constexpr int NSK_MAX_DEVICES = 64;
double *g_static_ptr[NSK_MAX_DEVICES];  // address of `scratch` on each device, null = not looked up
struct StaticLayout {                   // what the last call on a device left in the partners
  int nx, ld, ldd, ny, embedded;
} g_static_layout[NSK_MAX_DEVICES];
std::mutex g_static_mu;                 // guards the two tables (contexts of several threads share a device's array)
}  // namespace

// Called by nsk_create with `device` current: the address of this device's copy of `scratch`.
int nsk_i_static_scratch_init(int device, double **out) {
  NSK_REQUIRE(device >= 0 && device < NSK_MAX_DEVICES, "device %d beyond the %d the static scratch table holds", device,
              NSK_MAX_DEVICES);
  std::lock_guard<std::mutex> lock(g_static_mu);
  if (!g_static_ptr[device]) {
    double *ptr = nullptr;
    NSK_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
    NSK_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned a null address for the static scratch");
    g_static_ptr[device] = ptr;
    g_static_layout[device] = StaticLayout{0, 0, 0, 0, 0};
  }
  *out = g_static_ptr[device];
  return 0;
}

namespace {

int launch_poisson(nsk_ctx *ctx, hipStream_t stream, PoissonArgs &A, bool aligned, int div, const TaskPlan &plan) {
  switch (plan.K) {
    case 1: return launch_k<1>(ctx, stream, A, aligned, div, plan);
    case 2: return launch_k<2>(ctx, stream, A, aligned, div, plan);
    case 3: return launch_k<3>(ctx, stream, A, aligned, div, plan);
    case 4: return launch_k<4>(ctx, stream, A, aligned, div, plan);
    case 5: return launch_k<5>(ctx, stream, A, aligned, div, plan);
    case 6: return launch_k<6>(ctx, stream, A, aligned, div, plan);
    case 7: return launch_k<7>(ctx, stream, A, aligned, div, plan);
    case 8: return launch_k<8>(ctx, stream, A, aligned, div, plan);
  }
  nsk_set_error("passes per launch %d outside 1..8", plan.K);
  return 2;
}

// Does q = RN(x*rh + RN(x*rl)) equal RN(x/d) for EVERY finite x (barring underflow)?
// The argument of the final rounding is x/d (1 + e) with |e| < 2^-104 (rh + rl = 1/d to 2^-106,
// RN(x*rl) adds at most 2^-106), so the form can only fail where x/d lies that close to a midpoint
// between two adjacent doubles.  Write d = D 2^b and x = X 2^a with 2^52 <= D, X < 2^53.  For
// X >= D the quotient X/D lies in [1,2) and the midpoints are O 2^-53 with O odd; for X < D it lies
// in (1/2,1) and they are O 2^-54.  The distance to a midpoint is |X 2^s - O D| / (D 2^s), s = 53
// or 54, a nonzero integer N over D 2^s; relative to X/D that is N / (X 2^s), which is below 2^-100
// only for |N| < 128.  All (X, O) with X 2^s - O D = N are found from O = -N / D (mod 2^s): a handful
// per N.  The form is evaluated on every such X and compared with the IEEE quotient; outside this set
// x/d stays 16 times further from every midpoint than the form's error, so passing here is a proof
// for this divisor.  (About one random divisor in a hundred fails.)
bool two_op_division_is_exact(double d, double rh, double rl) {
  int e2 = 0;
  const double m = std::frexp(d, &e2);                  // d = m 2^e2, 0.5 <= m < 1
  const unsigned long long D = (unsigned long long)std::ldexp(m, 53);
  if (D < (1ull << 52) || D >= (1ull << 53)) return false;
  const int k = __builtin_ctzll(D);                      // D = 2^k D', D' odd
  if (k > 8) return false;                               // (many candidates: not worth it, use another form)
  const unsigned long long Dp = D >> k;
  unsigned long long inv = Dp;                           // inverse of D' modulo 2^64 (Newton)
  for (int i = 0; i < 6; i++) inv *= 2 - Dp * inv;
  const double dn = std::ldexp(m, 53);                   // the divisor scaled to an integer: same significand
  for (int s = 53; s <= 54; s++) {
    const int mb = s - k;                                // O is fixed modulo 2^mb
    const unsigned long long mod_mask = (1ull << mb) - 1;
    for (long long N = -128; N <= 128; N++) {
      if (N == 0 || (N & ((1ll << k) - 1)) != 0) continue;   // X 2^s - O D = N needs 2^k | N
      const unsigned long long O0 = ((unsigned long long)(-(N >> k)) * inv) & mod_mask;
      for (unsigned long long O = O0; O < (1ull << 54); O += (1ull << mb)) {
        if (O <= (1ull << 53) || !(O & 1)) continue;
        const __int128 num = (__int128)O * (__int128)D + (__int128)N;
        if (num <= 0 || (num & (((__int128)1 << s) - 1)) != 0) continue;
        const unsigned long long X = (unsigned long long)(num >> s);
        if (X < (1ull << 52) || X >= (1ull << 53)) continue;
        // the form is invariant under scaling x and d by powers of two: test with integers X, D
        const double x = (double)X;
        const double rhs = std::ldexp(rh, e2 - 53), rls = std::ldexp(rl, e2 - 53);   // 1/dn as hi + lo
        if (std::fma(x, rhs, x * rls) != x / dn) return false;
      }
    }
  }
  return true;
}

// Reciprocal of a divisor as hi + lo and the set of division forms that are exact for it
// (bit f set = form f usable; IEEE division always is).
unsigned reciprocal_for(double d, double *rh, double *rl) {
  *rh = 1.0 / d;
  *rl = 0.0;
  unsigned ok = 1u << DIV_IEEE;
  if (!(d >= 0x1p-30 && d <= 0x1p30)) return ok;
  const double e = std::fma(-d, *rh, 1.0);  // 1 - d*rh, exact
  *rl = e / d;                              // e/d = 1/d - rh exactly, one rounding
  ok |= 1u << DIV_FMA4;
  if (std::fabs(e) <= 0x1p-54) ok |= 1u << DIV_FMA3;
  if (two_op_division_is_exact(d, *rh, *rl)) ok |= 1u << DIV_FMA2;
  return ok;
}

// The form to run: the requested one (or the cheapest, want < 0) if every divisor supports it,
// otherwise the next one down the chain 2 -> 3 -> 4 -> IEEE.
int pick_division_form(unsigned ok, int want) {
  static const int chain[4] = {DIV_FMA2, DIV_FMA3, DIV_FMA4, DIV_IEEE};
  int i = 0;
  if (want >= 0)
    while (i < 3 && chain[i] != want) i++;
  while (i < 3 && !(ok & (1u << chain[i]))) i++;
  return chain[i];
}

// Residual (NavierStokes.jl:145) with max|.| over the owned rows; pitched Pr/divV.
__global__ void __launch_bounds__(256) k_residual_max(double *__restrict__ Rp,
                                                      const double *__restrict__ Pr,
                                                      const double *__restrict__ divV, int ld,
                                                      double rho_dt, double dx, double dy, int nx,
                                                      int ny, int own_lo, int own_hi,
                                                      unsigned long long *__restrict__ out) {
  // thread -> interior cell (ix,iy) of dPrdτ/Rp indexing (1-based, nx-2 x ny-2)
  // (a workgroup walks down the rows in steps of 4*gridDim.y: few workgroups, hence few atomics
  // on the one result word, which would otherwise cost more than the pass over the data)
  const int ix = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  unsigned long long m = 0;
  for (int iy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1; ix <= nx - 2 && iy <= ny - 2; iy += 4 * gridDim.y) {
    double c = A2(Pr, ld, ix + 1, iy + 1);
    double d2x = (A2(Pr, ld, ix + 2, iy + 1) - c) - (c - A2(Pr, ld, ix, iy + 1));
    double d2y = (A2(Pr, ld, ix + 1, iy + 2) - c) - (c - A2(Pr, ld, ix + 1, iy));
    double res = d2x / dx / dx + d2y / dy / dy - rho_dt * A2(divV, ld, ix + 1, iy + 1);
    if (Rp) A2(Rp, nx - 2, ix, iy) = res;
    if (iy + 1 >= own_lo && iy + 1 <= own_hi) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(fabs(res));
      m = b > m ? b : m;
    }
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

}  // namespace
int nsk_i_check_params(const nsk_ns_params *p) { return check_params(p); }
namespace {

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

// Arrays that the vectorised kernel cannot take as they are (dense with an odd leading
// dimension, or dPrdτ as its own (nx-2) x (ny-2) array: the reference's arrays) are copied into
// pitched staging arrays, iterated there and copied back: five array copies against `count`
// passes that each run about 20 % faster, worth it from a few dozen passes on.
constexpr int STAGE_MIN_PASSES = 32;

int staged_iters(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV, const nsk_ns_params *p,
                 int count) {
  const int nx = p->nx, ny = p->ny, ld = ld_pr_of(p), ldd = ld_dp_of(p);
  const int lds = (nx + 1 + 15) / 16 * 16;  // even, whole 128-byte lines, room for column nx
  const size_t one = (size_t)lds * ny;
  if (ctx->stage_elems < 3 * one) {
    NSK_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->stage) NSK_HIP(hipFree(ctx->stage));
    ctx->stage = nullptr;
    ctx->stage_elems = 0;
    NSK_HIP(hipMalloc(&ctx->stage, 3 * one * sizeof(double)));
    ctx->stage_elems = 3 * one;
  }
  double *sPr = ctx->stage, *sDp = ctx->stage + one, *sDv = ctx->stage + 2 * one;
  // (pad columns and the rim of the embedded dPrdτ must hold numbers, not stale NaN patterns)
  NSK_HIP(hipMemsetAsync(ctx->stage, 0, 3 * one * sizeof(double), ctx->stream));
  const size_t w = sizeof(double);
  NSK_HIP(hipMemcpy2DAsync(sPr, lds * w, Pr, ld * w, nx * w, ny, hipMemcpyDeviceToDevice, ctx->stream));
  NSK_HIP(hipMemcpy2DAsync(sDv, lds * w, divV, ld * w, nx * w, ny, hipMemcpyDeviceToDevice, ctx->stream));
  // dPrdτ value of pressure cell (ix,iy) goes to the offset of that cell
  double *dp_in = p->dp_embedded ? dPrdtau + ldd + 1 : dPrdtau;
  NSK_HIP(hipMemcpy2DAsync(sDp + lds + 1, lds * w, dp_in, ldd * w, (nx - 2) * w, ny - 2,
                           hipMemcpyDeviceToDevice, ctx->stream));
  nsk_ns_params q = *p;
  q.ld_pr = lds;
  q.ld_dp = 0;
  q.dp_embedded = 1;
  if (int rc = nsk_ns_poisson_iters(ctx, sPr, sDp, sDv, &q, count)) return rc;
  NSK_HIP(hipMemcpy2DAsync(Pr, ld * w, sPr, lds * w, nx * w, ny, hipMemcpyDeviceToDevice, ctx->stream));
  NSK_HIP(hipMemcpy2DAsync(dp_in, ldd * w, sDp + lds + 1, lds * w, (nx - 2) * w, ny - 2,
                           hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}

}  // namespace

namespace {

bool vector_storage(const double *Pr, const double *dPrdtau, const double *divV, const nsk_ns_params *p) {
  const int nx = p->nx, ld = ld_pr_of(p);
  return p->dp_embedded && (ld % 2 == 0) && ld >= ((nx + 1) & ~1) && ((uintptr_t)Pr % 16 == 0) &&
         ((uintptr_t)dPrdtau % 16 == 0) && ((uintptr_t)divV % 16 == 0);
}

// Rows of a window whose final values somebody is waiting for (nsk_ns_poisson_iters_edges_first).
struct EdgesFirst {
  int skip_lo, edge_lo, edge_hi, skip_hi;
  hipStream_t stream;  // the caller's stream that is to follow the edge rows
};
// Height of the tasks that produce the edge rows in the last launch: short, so that they are done
// long before the interior's (whose height the planner picks for throughput).
constexpr int EDGE_TASK_ROWS = 6;

// The parts of the kernel arguments that depend on the parameters only; returns the division form.
int constants_of(nsk_ctx *ctx, const nsk_ns_params *p, const double *divV, PoissonArgs *out) {
  const int nx = p->nx, ny = p->ny, ld = ld_pr_of(p), ldd = ld_dp_of(p);
  PoissonArgs &A = *out;
  memset(&A, 0, sizeof A);
  A.divv = divV;
  A.nx = nx;
  A.ny = ny;
  A.row_begin = 0;
  A.row_end = ny;
  A.row2_begin = A.row2_end = 0;
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
  // one form for both divisors, exact for both; a requested form (testing) is honoured only if it
  // is, so no setting can change a result bit
  const unsigned forms = reciprocal_for(p->dx, &A.rdx_h, &A.rdx_l) & reciprocal_for(p->dy, &A.rdy_h, &A.rdy_l);
  const int div = pick_division_form(forms, ctx->poisson_div);
  ctx->last_div = div;
  return div;
}

// `count` passes in place, with part_pr / part_dp as the ping-pong partners (same layout as the
// caller's arrays).  Allocates nothing.  With `ef`, the last launch is cut in three: the edge rows
// (both edges in one launch of short tasks, on ef->stream), the interior (on the context's stream)
// and the skipped rows, which are not produced.  The interior is planned for fewer wavefront slots
// than the chip has, so that the edge tasks find theirs although the interior is dispatched a few
// microseconds ahead of them (unplanned, the interior's tasks took every slot and the edge rows
// were ready no earlier than the interior -- measured).  What the caller queues on ef->stream
// afterwards (the transfer) follows the edge tasks in that stream's order, on a hardware queue the
// interior is not on; three streams (edge, interior, transfer) were too many: HIP maps streams
// onto four hardware queues and two of the three shared one -- measured.
int iters_with_partners(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV, const nsk_ns_params *p,
                        int count, double *part_pr, double *part_dp, const EdgesFirst *ef = nullptr) {
  const int nx = p->nx, ny = p->ny, ld = ld_pr_of(p), ldd = ld_dp_of(p);
  const bool aligned = vector_storage(Pr, dPrdtau, divV, p);
  const size_t pr_elems = (size_t)ld * ny;
  const size_t dp_elems = p->dp_embedded ? pr_elems : (size_t)ldd * (ny - 2);

  PoissonArgs A;
  const int div = constants_of(ctx, p, divV, &A);

  // Passes per launch and task shape: forced by the options, otherwise the cheapest by the
  // planner's estimate for this grid.
  // (tasks in pairs: pitched storage, whole-window launches -- of an edges-first call all but the one that is
  //  cut in three --, a caller that may allocate; the planner weighs them against plain tasks unless the option
  //  says "wherever they fit")
  const int pairs_likely = aligned && ctx->poisson_pairs && !ctx->no_alloc_call && (div == DIV_FMA2 || div == DIV_IEEE)
                               ? (ctx->poisson_pairs == 1 ? 1 : 2) : 0;
  const int ep = ctx->poisson_edge_percent;
  const TaskPlan top = choose_plan(nx, ny, ctx->num_simds, aligned, ctx->poisson_k, count < 8 ? count : 8,
                                   ctx->poisson_rows, ep, pairs_likely);
  const int kmax = top.K;
  ctx->last_plan_k = top.K;
  ctx->last_plan_rows = top.rows;

  // Split `count` into launches of at most kmax passes; make the number of
  // launches even when possible so the result lands in the caller's arrays.
  int nl = (count + kmax - 1) / kmax;
  if ((nl & 1) && count >= nl + 1) nl++;
  double *src_pr = Pr, *src_dp = dPrdtau, *dst_pr = part_pr, *dst_dp = part_dp;
  int left = count;
  bool edges_cut = false;
  for (int i = 0; i < nl; i++) {
    int k = (left + (nl - i) - 1) / (nl - i);
    A.pr_src = src_pr;
    A.dp_src = src_dp;
    A.pr_dst = dst_pr;
    A.dp_dst = dst_dp;
    // The cut is made only when the last launch writes into the caller's arrays (an even number
    // of launches): a copy home afterwards would undo it.
    if (ef && i == nl - 1 && dst_pr == Pr) {
      const int lo0 = ef->skip_lo, lo1 = lo0 + ef->edge_lo, hi1 = ny - ef->skip_hi, hi0 = hi1 - ef->edge_hi;
      NSK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
      NSK_HIP(hipStreamWaitEvent(ef->stream, ctx->ev_fork, 0));
      // both edges in one launch of short tasks
      A.row_begin = lo0;
      A.row_end = lo1;
      A.row2_begin = hi0;
      A.row2_end = hi1;
      const TaskPlan pe = plan_two_ranges(k, nx, lo1 - lo0, hi1 - hi0,
                                          ctx->edge_task_rows > 0 ? ctx->edge_task_rows : EDGE_TASK_ROWS, ep);
      if (int rc = launch_poisson(ctx, ef->stream, A, aligned, div, pe)) return rc;
      NSK_HIP(hipEventRecord(ctx->ev_edge, ef->stream));
      // the interior, planned for the wavefront slots the edge tasks do not hold
      A.row_begin = lo1;
      A.row_end = hi0;
      A.row2_begin = A.row2_end = 0;
      long simds_left = ctx->num_simds - (pe.tasks + NSK_WAVES - 1) / NSK_WAVES;
      if (simds_left < ctx->num_simds / 2) simds_left = ctx->num_simds / 2;
      // (plain tasks: workgroups of paired tasks take a quarter of a compute unit's LDS each and left the edge
      //  tasks no room -- the edge rows were then ready 52 us after the interior was dispatched instead of 13)
      const TaskPlan pi = choose_plan(nx, hi0 - lo1, (int)simds_left, aligned, k, k, ctx->poisson_rows, ep);
      if (int rc = launch_poisson(ctx, ctx->stream, A, aligned, div, pi)) return rc;
      // later work on the context's stream sees the edge rows as well
      NSK_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_edge, 0));
      A.row_begin = 0;
      A.row_end = ny;
      edges_cut = true;
    } else {
      const TaskPlan plan = k == top.K ? top : choose_plan(nx, ny, ctx->num_simds, aligned, k, k, ctx->poisson_rows, ep, pairs_likely);
      if (int rc = launch_poisson(ctx, ctx->stream, A, aligned, div, plan)) return rc;
    }
    left -= k;
    double *t = src_pr; src_pr = dst_pr; dst_pr = t;
    t = src_dp; src_dp = dst_dp; dst_dp = t;
  }
  if (src_pr != Pr) {  // odd number of launches: bring the result home
    NSK_HIP(hipMemcpyAsync(Pr, src_pr, pr_elems * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    NSK_HIP(hipMemcpyAsync(dPrdtau, src_dp, dp_elems * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (ef && !edges_cut) {  // everything ran on the context's stream: the edge stream follows it
    NSK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    NSK_HIP(hipStreamWaitEvent(ef->stream, ctx->ev_fork, 0));
  }
  return 0;
}

}  // namespace

extern "C" int nsk_ns_poisson_iters(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV,
                                    const nsk_ns_params *p, int count) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, dPrdtau, divV);
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(count >= 0, "negative pass count %d", count);
  if (count == 0) return 0;
  // ping-pong partners use the layout of the caller's arrays
  const size_t pr_elems = (size_t)ld_pr_of(p) * p->ny;
  const size_t dp_elems = p->dp_embedded ? pr_elems : (size_t)ld_dp_of(p) * (p->ny - 2);
  SmallPlan tiles;
  if (small_grid_applies(ctx, p, count, &tiles)) {  // fields that fit on chip: one launch
    if (int rc = ensure_pingpong(ctx, pr_elems > dp_elems ? pr_elems : dp_elems)) return rc;
    PoissonArgs A;
    const int div = constants_of(ctx, p, divV, &A);
    ctx->last_plan_k = count;
    ctx->last_plan_rows = tiles.tile_rows;
    ctx->last_div = div == DIV_FMA2 ? DIV_FMA2 : DIV_IEEE;
    return small_grid_iters(ctx, Pr, dPrdtau, divV, p, count, A, div == DIV_FMA2 ? DIV_FMA2 : DIV_IEEE, ctx->pp_pr,
                            ctx->pp_dp, tiles);
  }
  if (!vector_storage(Pr, dPrdtau, divV, p) && ctx->poisson_stage && count >= STAGE_MIN_PASSES)
    return staged_iters(ctx, Pr, dPrdtau, divV, p, count);
  if (int rc = ensure_pingpong(ctx, pr_elems > dp_elems ? pr_elems : dp_elems)) return rc;
  return iters_with_partners(ctx, Pr, dPrdtau, divV, p, count, ctx->pp_pr, ctx->pp_dp);
}

// nsk_ns_poisson_iters for a window whose edge rows somebody is waiting for (a halo exchange):
// in the last launch the rows [skip_lo, skip_lo+edge_lo) and [ny-skip_hi-edge_hi, ny-skip_hi) are
// produced first, by short tasks on `edge_stream`, the rows between them on the context's stream,
// and the skip_lo first / skip_hi last rows (halo rows about to be received) are not produced: they
// keep undefined contents.  Work queued on edge_stream afterwards starts when the edge rows are
// final, i.e. while the interior is still being computed; work queued on the context's stream
// afterwards sees all rows.  Values are those of nsk_ns_poisson_iters.
extern "C" int nsk_ns_poisson_iters_edges_first(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV,
                                                const nsk_ns_params *p, int count, int skip_lo, int edge_lo,
                                                int edge_hi, int skip_hi, void *edge_stream) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, dPrdtau, divV);
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(count >= 0, "negative pass count %d", count);
  NSK_REQUIRE(edge_stream != nullptr, "nsk_ns_poisson_iters_edges_first: edge_stream is null");
  NSK_REQUIRE(skip_lo >= 0 && edge_lo >= 0 && edge_hi >= 0 && skip_hi >= 0 &&
                  (long)skip_lo + edge_lo + edge_hi + skip_hi <= p->ny,
              "edge rows %d+%d / %d+%d do not fit a window of %d rows", skip_lo, edge_lo, edge_hi, skip_hi, p->ny);
  if (!ctx->ev_fork) {
    NSK_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    NSK_HIP(hipEventCreateWithFlags(&ctx->ev_edge, hipEventDisableTiming));
  }
  EdgesFirst ef{skip_lo, edge_lo, edge_hi, skip_hi, static_cast<hipStream_t>(edge_stream)};
  const bool pitched = vector_storage(Pr, dPrdtau, divV, p);
  // no interior rows between the edges, nothing to run ahead, or arrays that are iterated on
  // staging copies: the plain sequence, with the edge stream ordered behind it
  const bool cut = count > 0 && (edge_lo > 0 || edge_hi > 0) && skip_lo + edge_lo + edge_hi + skip_hi < p->ny &&
                   (pitched || !(ctx->poisson_stage && count >= STAGE_MIN_PASSES));
  if (!cut) {
    if (int rc = nsk_ns_poisson_iters(ctx, Pr, dPrdtau, divV, p, count)) return rc;
    NSK_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    NSK_HIP(hipStreamWaitEvent(ef.stream, ctx->ev_fork, 0));
    return 0;
  }
  const size_t pr_elems = (size_t)ld_pr_of(p) * p->ny;
  const size_t dp_elems = p->dp_embedded ? pr_elems : (size_t)ld_dp_of(p) * (p->ny - 2);
  if (int rc = ensure_pingpong(ctx, pr_elems > dp_elems ? pr_elems : dp_elems)) return rc;
  return iters_with_partners(ctx, Pr, dPrdtau, divV, p, count, ctx->pp_pr, ctx->pp_dp, &ef);
}

// nsk_ns_poisson_iters for callers that cannot allocate: the ping-pong partners live in the
// static device array `scratch`, so the call performs no hipMalloc / hipFree whatever the
// arguments.  Arrays that the vectorised kernel cannot take run the register-input kernel in
// place (the staging copies of nsk_ns_poisson_iters would have to be allocated).
extern "C" int nsk_ns_poisson_iters_static(nsk_ctx *ctx, double *Pr, double *dPrdtau, const double *divV,
                                           const nsk_ns_params *p, int count) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, dPrdtau, divV);
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(count >= 0, "negative pass count %d", count);
  NSK_REQUIRE(ctx->static_scratch != nullptr, "nsk_ns_poisson_iters_static: the static scratch of device %d is unavailable (%s)",
              ctx->device, "hipGetSymbolAddress failed in nsk_create");
  const int ld = ld_pr_of(p), ldd = ld_dp_of(p);
  const size_t pr_elems = (size_t)ld * p->ny;
  const size_t dp_elems = p->dp_embedded ? pr_elems : (size_t)ldd * (p->ny - 2);
  const size_t elems = pr_elems > dp_elems ? pr_elems : dp_elems;
  NSK_REQUIRE(elems <= NSK_STATIC_ELEMS,
              "nsk_ns_poisson_iters_static: a window of %d x %d values (leading dimension x rows = %zu) exceeds the "
              "static scratch of %zu values per field; use nsk_ns_poisson_iters",
              ld, p->ny, elems, NSK_STATIC_ELEMS);
  if (count == 0) return 0;
  double *part_pr = ctx->static_scratch, *part_dp = ctx->static_scratch + NSK_STATIC_ELEMS;
  // Every call on this GPU shares the partners and they keep the last call's contents.  A launch
  // writes every value the next launch uses, but pad columns and unused rim cells are read as
  // don't-care lanes: with a new layout they would hold whatever the last layout left there, so
  // the part in use is cleared when the layout changes.
  // (nx is part of the layout: grids of different width can share a pitch, and the columns between the
  // narrower grid's last one and the pitch would otherwise keep the wider grid's values)
  bool clear;
  {
    std::lock_guard<std::mutex> lock(g_static_mu);
    StaticLayout &last = g_static_layout[ctx->device];
    clear = last.nx != p->nx || last.ld != ld || last.ldd != ldd || last.ny != p->ny || last.embedded != p->dp_embedded;
    last = StaticLayout{p->nx, ld, ldd, p->ny, p->dp_embedded};
  }
  if (clear) {
    NSK_HIP(hipMemsetAsync(part_pr, 0, elems * sizeof(double), ctx->stream));
    NSK_HIP(hipMemsetAsync(part_dp, 0, elems * sizeof(double), ctx->stream));
  }
  ctx->no_alloc_call = true;  // (the paired tasks keep their task list in device memory)
  const int rc = iters_with_partners(ctx, Pr, dPrdtau, divV, p, count, part_pr, part_dp);
  ctx->no_alloc_call = false;
  return rc;
}

extern "C" int nsk_division_form(double d, int *form_out, double *rh_out, double *rl_out) {
  NSK_REQUIRE(form_out != nullptr, "nsk_division_form: form_out is null");
  double rh = 0.0, rl = 0.0;
  *form_out = pick_division_form(reciprocal_for(d, &rh, &rl), -1);
  if (rh_out) *rh_out = rh;
  if (rl_out) *rl_out = rl;
  return 0;
}

extern "C" int nsk_poisson_plan(int nx, int ny, int num_simds, int staged, int k_forced, int k_limit,
                                int rows_forced, int edge_percent, int *out) {
  NSK_REQUIRE(out != nullptr && nx >= 3 && ny >= 3 && num_simds >= 1 && k_forced >= 0 && k_forced <= 8 &&
                  k_limit >= 1 && rows_forced >= 0,
              "nsk_poisson_plan: bad arguments");
  const TaskPlan t = choose_plan(nx, ny, num_simds, staged != 0, k_forced, k_limit < 8 ? k_limit : 8, rows_forced,
                                 edge_percent);
  out[0] = t.K;
  out[1] = t.rows;
  out[2] = t.rows_edge;
  out[3] = t.nstrips;
  out[4] = (int)t.tasks;
  out[5] = (int)(t.est_us_per_pass * 1000.0);
  return 0;
}

extern "C" int nsk_ns_residual(nsk_ctx *ctx, double *Rp, const double *Pr, const double *divV,
                               const nsk_ns_params *p, int own_lo, int own_hi, double *maxabs_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, divV);  // Rp may be null: only the maximum is wanted
  if (int rc = check_params(p)) return rc;
  NSK_REQUIRE(maxabs_out != nullptr, "nsk_ns_residual: maxabs_out is null");
  const int nx = p->nx, ny = p->ny;
  NSK_HIP(hipMemsetAsync(ctx->red_dev, 0, sizeof(unsigned long long), ctx->stream));
  const unsigned gy = nsk_cdiv(ny - 2, 4);
  dim3 grid(nsk_cdiv(nx - 2, 64), gy < 64 ? gy : 64);
  k_residual_max<<<grid, 256, 0, ctx->stream>>>(Rp, Pr, divV, ld_pr_of(p), p->rho / p->dt, p->dx,
                                               p->dy, nx, ny, own_lo, own_hi, ctx->red_dev);
  NSK_LAUNCH_CHECK();
  return nsk_read_maxabs(ctx, maxabs_out);
}

extern "C" int nsk_ns_poisson(nsk_ctx *ctx, double *Pr, double *dPrdtau, double *Rp,
                              const double *divV, const nsk_ns_params *p, int *iters_out,
                              double *err_hist, int *iter_hist, int hist_cap, int *nhist_out) {
  NSK_CTX(ctx);
  NSK_NONNULL(Pr, dPrdtau, Rp, divV);
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
      double err = m * (p->ly * p->ly) / p->psc;  // NavierStokes.jl:90, m*ly^2/psc: the power binds first
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
