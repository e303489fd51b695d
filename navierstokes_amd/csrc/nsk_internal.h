// Internal declarations shared by the libnsk translation units (not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <array>
#include <map>
#include <utility>

#include "../../include/nsk.h"


struct nsk_ctx {
  int device = 0;
  int num_simds = 1024;  // 4 per compute unit
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;  // stream in use (own_stream or the caller's)
  // Reduction slot for max|.| (bit pattern of a non-negative double) and its
  // pinned host mirror.
  unsigned long long *red_dev = nullptr;
  unsigned long long *red_host = nullptr;
  // Ping-pong partners of Pr / dPrdtau for the multi-iteration Poisson kernel,
  // (re)allocated on demand by the fused entry points.
  double *pp_pr = nullptr, *pp_dp = nullptr;
  size_t pp_elems = 0;
  // This device's copy of the static ping-pong partners (nsk_ns_poisson_iters_static); null if
  // its address could not be obtained.
  double *static_scratch = nullptr;
  // Pitched staging copies of Pr, dPrdtau, divV for callers whose arrays are dense (odd leading
  // dimension): long runs of passes are done on these and copied back.
  double *stage = nullptr;
  size_t stage_elems = 0;
  int poisson_stage = 1;  // 0 = never stage (nsk_set_option "poisson_stage")
  // Per-iteration max|old-new| history of the batched iterative solvers (device + pinned host).
  unsigned long long *hist_dev = nullptr, *hist_host = nullptr;
  size_t hist_cap = 0;
  double *iter_pp = nullptr;  // ping-pong partner of p for those solvers
  size_t iter_pp_elems = 0;
  int poisson_k = 0;     // passes per launch, 0 = chosen per grid (NSK_POISSON_K / nsk_set_option)
  int last_plan_k = 0, last_plan_rows = 0;  // what the last fused Poisson call used (nsk_get_option)
  int last_launch_pairs = 0;                // its last launch ran its tasks in pairs (k_poisson_pairs)
  int poisson_div = -1;  // -1 = automatic; 0 / 2 / 3 / 4 ask for a division form (testing)
  int last_div = -1;     // form the last fused Poisson call used (nsk_get_option "poisson_div_used")
  int poisson_edge_percent = 0;  // rows per task of the edge strips in % of poisson_rows, 0 = default
  int poisson_pairs = 1;  // 1 = tasks of the multi-pass kernel in pairs that walk away from a common row (no halo rows between them)
  bool no_alloc_call = false;  // inside nsk_ns_poisson_iters_static: nothing may be allocated
  // task lists of those launches, by the shape of the launch: (device pointer, workgroups)
  std::map<std::array<int, 8>, std::pair<void *, unsigned>> pair_tables;
  int poisson_tall_percent = 0;  // interior chunks in pairs of poisson_rows*(100 +- this)/100 rows, 0 = all alike
  int poisson_xcd = 1;   // XCD-aware task numbering of the Poisson kernel (0 = plain)
  int poisson_stream = -1;  // non-temporal requests for Pr and dPrdtau rows: -1 = by the size of the fields, 0 / 1
  int edge_task_rows = 0;  // task height of the edge rows in nsk_ns_poisson_iters_edges_first, 0 = default
  // nsk_ns_poisson_iters_edges_first (created on first use): fork and edge-rows-final events
  hipEvent_t ev_fork = nullptr, ev_edge = nullptr;
  int bub_merge_launches = 1;  // bubble time steps: 1 = element-wise kernels as phases of the solver launches on grids one workgroup holds
  int sor_fast_division = 1;  // sweeps-in-flight pressure solve: exact division through reciprocals (IEEE rerun when out of range)
  int sor_pipeline = 1;  // lexicographic SOR: 1 = several sweeps in flight when everything fits LDS, 0 = one at a time
  double *sor_snap = nullptr;  // per-sweep snapshots of the unknown for that kernel
  size_t sor_snap_elems = 0;
  unsigned long long *run_words = nullptr;  // per-step counters of nsk_bub_dns_run / nsk_bub_code3_run
  size_t run_words_cap = 0;
  int step_fused = 1;    // nsk_ns_step: 1 = three fused kernels around the Poisson loop, 0 = one launch per call site
  int poisson_rows = 0;  // rows per task chunk, 0 = automatic (NSK_POISSON_ROWS / nsk_set_option)
  int code3_steps_done = 0;  // complete time steps of the last nsk_bub_code3_step / _run call
  // lexicographic SOR on the whole chip (bubble_sor_grid.hip): control words, progress words and tile order
  int sor_grid = 1;       // 1 = grids beyond one workgroup's LDS are solved by tiles on the whole chip, 0 = by one workgroup
  int sor_grid_rows = 0;  // tile height, 0 = default
  int sor_grid_copies = 0;  // copies of the unknown in the ring, 0 = by the grid (testing)
  unsigned *sor_grid_ctl = nullptr;
  size_t sor_grid_words = 0, sor_grid_zero = 0;
  int sor_grid_ti = 0, sor_grid_tj = 0;
  // Poisson passes of small grids in one launch (ns_poisson_small.hip): exchange rows + progress words
  int poisson_small = 1;       // 1 = grids whose fields fit on chip run a whole call in one launch, 0 = never
  int poisson_small_halo = 0;  // passes between two exchanges of edge rows, 0 = default
  int poisson_small_rows = 0, poisson_small_cols = 0;  // tile of a workgroup, 0 = by the grid
  int poisson_small_spin_limit = 0;   // polls before a workgroup gives up on a neighbour, 0 = default
  int poisson_small_test_absent = 0;  // testing: 1 + the workgroup that leaves at once
  double *small_xbuf = nullptr;
  size_t small_xbuf_elems = 0, small_words = 0;
  int small_trace_at = 0, small_trace_groups = 0;  // debug builds (-DNSK_SMALL_TRACE)
};

// Word of red_host that a kernel sets when one of its bounded waits ran out.
constexpr int NSK_DEVICE_ERROR_WORD = 24;
// After a synchronisation of the context's stream: has a kernel given up on a wait since the last look?
int nsk_device_error_check(nsk_ctx *ctx, const char *where);

// bubble_sor_grid.hip
int nsk_sor_grid(nsk_ctx *ctx, int mode, double *a, const double *c1, const double *c2, const double *c3,
                 double beta, double dx, double dy, int nx, int ny, int maxit, double maxError, int *d_sweeps,
                 double *d_change);

void nsk_set_error(const char *fmt, ...);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) unless the kernel has already been allowed at least
// `bytes` on this context's device.  The attribute belongs to (kernel, device), not to a context: the
// table is shared by all contexts and only ever raises a kernel's limit.
hipError_t nsk_allow_lds(nsk_ctx *ctx, const void *kernel, size_t bytes);

#define NSK_HIP(call)                                                          \
  do {                                                                         \
    hipError_t e_ = (call);                                                    \
    if (e_ != hipSuccess) {                                                    \
      nsk_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                    __FILE__, __LINE__);                                       \
      return 1;                                                                \
    }                                                                          \
  } while (0)

#define NSK_REQUIRE(cond, ...)                                                 \
  do {                                                                         \
    if (!(cond)) {                                                             \
      nsk_set_error(__VA_ARGS__);                                              \
      return 2;                                                                \
    }                                                                          \
  } while (0)

// as NSK_REQUIRE with a return value of its own (NSK_ERR_* of include/nsk.h)
#define NSK_REQUIRE_RC(cond, rc, ...)                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      nsk_set_error(__VA_ARGS__);                                              \
      return (rc);                                                             \
    }                                                                          \
  } while (0)

#define NSK_CTX(ctx)                                                           \
  NSK_REQUIRE((ctx) != nullptr, "null context");                               \
  NSK_HIP(hipSetDevice((ctx)->device))

// A null device array would be dereferenced on the GPU (a memory fault there can take the
// whole device down), so every entry point rejects it on the host.
template <class... T>
static inline bool nsk_all_nonnull(const T *...p) {
  return (... && (p != nullptr));
}
#define NSK_NONNULL(...) \
  NSK_REQUIRE(nsk_all_nonnull(__VA_ARGS__), "%s: null array argument", __func__)

#define NSK_LAUNCH_CHECK() NSK_HIP(hipGetLastError())

// Column-major element: A[ix,iy], 1-based like the reference's kernels.
#define A2(a, ld, ix, iy) (a)[(size_t)((iy)-1) * (size_t)(ld) + (size_t)((ix)-1)]

static inline unsigned nsk_cdiv(unsigned a, unsigned b) { return (a + b - 1) / b; }
