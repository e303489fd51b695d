// Slab decomposition of the pressure-Poisson loop behind the C ABI (nsk_slabs_*).
//
// The reference is single-device (no MPI / NCCL anywhere in it); this is the multi-GPU form of its
// loop scripts/NavierStokes.jl:84-95.  The global grid (nx x nyg) is cut along iy into `world`
// slabs of ny_loc rows; slab g holds its rows plus `halo` rows per interior side.  One pass moves
// information by one row, so a window with `halo` valid halo rows runs `halo` passes between two
// exchanges of `halo` rows of Pr and dPrdτ (bit-identical to the single domain: every owned cell
// is computed by the same expression from the same inputs).
//
// A group object holds the slabs of ONE process: all of them (a single host thread drives every
// GPU of the node, the natural shape for a Julia driver) or one (one process per GPU).  Halo rows
// travel either through RCCL (ncclSend / ncclRecv of whole row blocks inside one ncclGroup per
// exchange, ncclAllReduce(max) of the residual's bit pattern) or, when every slab lives in this
// process, through peer-to-peer device copies on the slabs' communication streams.  RCCL is
// loaded at run time (dlopen), so a single-GPU user of the library does not depend on it.
//
// Streams: each slab's kernels run on its context's stream; transfers run on a second stream per
// slab, ordered against the kernels with events, never with host waits -- the host thread only
// queues work, so the slabs of a group run concurrently.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cmath>
#include <thread>
#include <vector>

#include "nsk_internal.h"

// ns_step_fused.hip
int nsk_i_step_predictor(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p);
int nsk_i_step_corrector(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p);
int nsk_i_step_advect_out(nsk_ctx *ctx, const nsk_ns_state *s, const nsk_ns_params *p);

namespace {

// ---- RCCL entry points, resolved on first use ---------------------------------------------------
struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  char why[256] = "";  // when not ok
};

Rccl load_rccl() {
  Rccl r;
  // a copy already mapped into the process (a host framework's) is found first by its soname
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    const char *why = dlerror();
    snprintf(r.why, sizeof r.why, "%s", why ? why : "librccl.so.1 not found");
    return r;
  }
#define NSK_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym))
  NSK_SYM(GetUniqueId, "ncclGetUniqueId");
  NSK_SYM(CommInitRank, "ncclCommInitRank");
  NSK_SYM(CommDestroy, "ncclCommDestroy");
  NSK_SYM(GroupStart, "ncclGroupStart");
  NSK_SYM(GroupEnd, "ncclGroupEnd");
  NSK_SYM(Send, "ncclSend");
  NSK_SYM(Recv, "ncclRecv");
  NSK_SYM(AllReduce, "ncclAllReduce");
  NSK_SYM(GetErrorString, "ncclGetErrorString");
#undef NSK_SYM
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv &&
         r.AllReduce && r.GetErrorString;
  if (!r.ok) snprintf(r.why, sizeof r.why, "the library lacks an entry point this code needs");
  return r;
}

// Loaded once (the initialisation of a function-local static is thread-safe); null when unusable.
const Rccl &rccl_state() {
  static const Rccl r = load_rccl();
  return r;
}
const Rccl *rccl() { return rccl_state().ok ? &rccl_state() : nullptr; }

// An RCCL group left open by an early return would swallow every later call of the process:
// the guard closes it on every path.
struct GroupGuard {
  const Rccl *R;
  bool open = false;
  explicit GroupGuard(const Rccl *r) : R(r) {}
  ncclResult_t start() {
    const ncclResult_t e = R->GroupStart();
    open = e == ncclSuccess;
    return e;
  }
  ncclResult_t end() {
    open = false;
    return R->GroupEnd();
  }
  ~GroupGuard() {
    if (open) (void)R->GroupEnd();
  }
};

#define NSK_NCCL(call)                                                                              \
  do {                                                                                              \
    ncclResult_t r_ = (call);                                                                       \
    if (r_ != ncclSuccess) {                                                                        \
      nsk_set_error("%s failed: %s (%s:%d)", #call, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)

enum { TRANSPORT_AUTO = 0, TRANSPORT_RCCL = 1, TRANSPORT_COPY = 2 };

struct Slab {
  int rank = 0, device = 0;
  int lo_halo = 0, hi_halo = 0, iy0 = 0, ny_win = 0, own_lo = 0, own_hi = 0;  // 0-based window rows, own_hi exclusive
  nsk_ctx *ctx = nullptr;
  hipStream_t comm_stream = nullptr;
  hipStream_t edge_stream = nullptr;  // last launch of a block: the rows the neighbours wait for
  hipEvent_t ev_done = nullptr, ev_comm = nullptr;
  double *Pr = nullptr, *dP = nullptr, *divV = nullptr;  // (ny_win x ld), dPrdτ on the Pr layout
  // whole-time-step state (nsk_slabs_init_flow), dense windows: Vx (nx+1 x ny_win), Vy (nx x ny_win+1),
  // the rest (nx x ny_win) and txy (nx-1 x ny_win-1)
  double *Vx = nullptr, *Vy = nullptr, *C = nullptr, *Vx_o = nullptr, *Vy_o = nullptr, *C_o = nullptr;
  double *txx = nullptr, *tyy = nullptr, *txy = nullptr, *Vprof = nullptr;
  unsigned long long *red = nullptr;                     // one word for the all-reduce
  ncclComm_t comm = nullptr;
  nsk_ns_params prm{};
};

}  // namespace

struct nsk_slabs {
  int world = 1, first = 0, nlocal = 1, ny_loc = 0, halo = 0, ld = 0, transport = TRANSPORT_COPY;
  nsk_ns_params global{};
  std::vector<Slab> slabs;
  unsigned long long *red_host = nullptr;  // pinned
  int exchanges = 0;
  bool flow = false;  // the velocity / tracer windows exist
  // Poisson blocks end with the rows to be sent produced first (nsk_ns_poisson_iters_edges_first),
  // so that they travel while the rest of the window is still being computed
  bool edges_first = true;
  // Host-side waits are loops over hipStreamQuery with a deadline: a transfer that never completes (a
  // mismatched send / receive on the first run on a real node, a peer that died) ends the call with
  // NSK_ERR_TIMEOUT and a description instead of hanging.  first_contact_ms applies until the first
  // wait behind an exchange or an all-reduce has succeeded, deadline_ms afterwards; 0 = wait for good.
  int first_contact_ms = 180000, deadline_ms = 60000;
  bool contact_made = false;
  int test_stall_ms = 0;  // (testing) the next exchange is held back on the device for this long
  bool step_fused = true;  // nsk_slabs_step: three fused kernels around the Poisson loop instead of eleven launches
  size_t last_exchange_bytes = 0;
};

namespace {

inline int pitch_of(int nx) { return (nx + 1 + 15) / 16 * 16; }

// Fields a slab exchanges halo rows of.  Cell fields have ny_win rows; the face field Vy has
// ny_win + 1: face f is the lower face of cell row f, a slab owns the faces of its own cell rows
// (the top slab also the last one), so h + 1 faces travel towards the lower neighbour and h upwards.
enum { F_PR = 0, F_DP, F_DIVV, F_VX, F_VY, F_C, F_VXO, F_VYO, F_CO, F_COUNT };
struct XField {
  int id;
  size_t ld;
  bool faces;
};
inline double *field_ptr(const Slab &s, int id) {
  switch (id) {
    case F_PR: return s.Pr;
    case F_DP: return s.dP;
    case F_DIVV: return s.divV;
    case F_VX: return s.Vx;
    case F_VY: return s.Vy;
    case F_C: return s.C;
    case F_VXO: return s.Vx_o;
    case F_VYO: return s.Vy_o;
    case F_CO: return s.C_o;
  }
  return nullptr;
}
inline XField xfield(const nsk_slabs *g, int id) {
  const size_t nx = (size_t)g->global.nx;
  if (id <= F_DIVV) return {id, (size_t)g->ld, false};
  if (id == F_VX || id == F_VXO) return {id, nx + 1, false};
  if (id == F_VY || id == F_VYO) return {id, nx, true};
  return {id, nx, false};
}

// A kernel that does nothing for `ms` milliseconds (100 MHz clock) and then ends: the stand-in for a
// transfer that does not complete, used by the deadline test.  Bounded, so the stream always drains.
__global__ void k_stall(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

// Waits for `st` (of slab `s`) with the group's deadline; see nsk_slabs::deadline_ms.
int wait_stream(nsk_slabs *g, Slab &s, hipStream_t st, const char *what) {
  const int limit = g->contact_made ? g->deadline_ms : g->first_contact_ms;
  NSK_HIP(hipSetDevice(s.device));
  if (limit <= 0) {
    NSK_HIP(hipStreamSynchronize(st));
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) {
      nsk_set_error("%s: hipStreamQuery failed: %s", what, hipGetErrorString(e));
      return NSK_ERR_HIP;
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (ms > limit) {
      const bool last_done = s.ev_comm && hipEventQuery(s.ev_comm) == hipSuccess;
      nsk_set_error("%s: no completion within %d ms on slab %d of %d (device %d, neighbours %d and %d): %d exchanges "
                    "queued so far, the last one (%zu bytes per neighbour and direction, transport %s, %s schedule) has "
                    "%s on this slab; the group is not usable after this",
                    what, limit, s.rank, g->world, s.device, s.rank > 0 ? s.rank - 1 : -1,
                    s.rank < g->world - 1 ? s.rank + 1 : -1, g->exchanges, g->last_exchange_bytes,
                    g->transport == TRANSPORT_RCCL ? "RCCL" : "device copies", g->edges_first ? "edges-first" : "plain",
                    last_done ? "completed" : "NOT completed");
      return NSK_ERR_TIMEOUT;
    }
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
}

// Refresh `h` halo rows per interior side of the listed fields on every local slab.  The rows to
// be sent are final when the work queued so far on the slab's stream is done -- or, after
// nsk_ns_poisson_iters_edges_first, on its edge stream (from_edge).
int exchange_list(nsk_slabs *g, const XField *xf, int nf, int h, bool from_edge = false) {
  if (g->world == 1 || nf == 0 || h == 0) return 0;
  // After an edges-first block the transfer is queued on the edge stream itself, behind the edge
  // tasks: a third stream could share a hardware queue with the interior's (HIP maps streams onto
  // four hardware queues) and would then wait for the interior.
  auto xstream = [&](Slab &s) { return from_edge ? s.edge_stream : s.comm_stream; };
  for (Slab &s : g->slabs) {  // transfers start when the rows to be sent are final
    NSK_HIP(hipSetDevice(s.device));
    NSK_HIP(hipEventRecord(s.ev_done, from_edge ? s.edge_stream : s.ctx->stream));
  }
  g->last_exchange_bytes = 0;
  for (int f = 0; f < nf; f++) g->last_exchange_bytes += (size_t)(xf[f].faces ? h + 1 : h) * xf[f].ld * sizeof(double);
  if (g->test_stall_ms > 0) {
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      if (!from_edge) NSK_HIP(hipStreamWaitEvent(s.comm_stream, s.ev_done, 0));
      k_stall<<<1, 64, 0, xstream(s)>>>((long long)g->test_stall_ms * 100000);
      NSK_LAUNCH_CHECK();
    }
    g->test_stall_ms = 0;
  }
  if (g->transport == TRANSPORT_RCCL) {
    const Rccl *R = rccl();
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      if (!from_edge) NSK_HIP(hipStreamWaitEvent(s.comm_stream, s.ev_done, 0));
    }
    GroupGuard group(R);
    NSK_NCCL(group.start());
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      for (int f = 0; f < nf; f++) {
        double *a = field_ptr(s, xf[f].id);
        const size_t ld = xf[f].ld, up = (size_t)(xf[f].faces ? h + 1 : h) * ld, dn = (size_t)h * ld;
        if (s.rank > 0) {  // lower neighbour: my first owned rows go down, its last owned rows come up
          NSK_NCCL(R->Send(a + (size_t)s.own_lo * ld, up, ncclDouble, s.rank - 1, s.comm, xstream(s)));
          NSK_NCCL(R->Recv(a + (size_t)(s.own_lo - h) * ld, dn, ncclDouble, s.rank - 1, s.comm, xstream(s)));
        }
        if (s.rank < g->world - 1) {
          NSK_NCCL(R->Send(a + (size_t)(s.own_hi - h) * ld, dn, ncclDouble, s.rank + 1, s.comm, xstream(s)));
          NSK_NCCL(R->Recv(a + (size_t)s.own_hi * ld, up, ncclDouble, s.rank + 1, s.comm, xstream(s)));
        }
      }
    }
    NSK_NCCL(group.end());
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      NSK_HIP(hipEventRecord(s.ev_comm, xstream(s)));
      NSK_HIP(hipStreamWaitEvent(s.ctx->stream, s.ev_comm, 0));
    }
  } else {
    // Every slab pulls its halo rows from its neighbours' owned rows on its own communication
    // stream, once its own kernels (which still read and write the halo rows) and the
    // neighbour's (which produce the rows) are done.
    const int n = (int)g->slabs.size();
    auto copy = [&](Slab &to, double *dst, Slab &from, const double *src, size_t count) -> hipError_t {
      if (from.device == to.device)
        return hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToDevice, xstream(to));
      return hipMemcpyPeerAsync(dst, to.device, src, from.device, count * sizeof(double), xstream(to));
    };
    for (int i = 0; i < n; i++) {
      Slab &s = g->slabs[i];
      NSK_HIP(hipSetDevice(s.device));
      if (!from_edge) NSK_HIP(hipStreamWaitEvent(s.comm_stream, s.ev_done, 0));
      if (i > 0) NSK_HIP(hipStreamWaitEvent(xstream(s), g->slabs[i - 1].ev_done, 0));
      if (i < n - 1) NSK_HIP(hipStreamWaitEvent(xstream(s), g->slabs[i + 1].ev_done, 0));
      for (int f = 0; f < nf; f++) {
        double *a = field_ptr(s, xf[f].id);
        const size_t ld = xf[f].ld, up = (size_t)(xf[f].faces ? h + 1 : h) * ld, dn = (size_t)h * ld;
        if (i > 0) {  // the lower neighbour's last owned rows
          Slab &lo = g->slabs[i - 1];
          NSK_HIP(copy(s, a + (size_t)(s.own_lo - h) * ld, lo, field_ptr(lo, xf[f].id) + (size_t)(lo.own_hi - h) * ld, dn));
        }
        if (i < n - 1) {  // the upper neighbour's first owned rows (one more face)
          Slab &hi = g->slabs[i + 1];
          NSK_HIP(copy(s, a + (size_t)s.own_hi * ld, hi, field_ptr(hi, xf[f].id) + (size_t)hi.own_lo * ld, up));
        }
      }
      NSK_HIP(hipEventRecord(s.ev_comm, xstream(s)));
    }
    // a slab's kernels go on when its halo rows have arrived and its neighbours have read its rows
    for (int i = 0; i < n; i++) {
      Slab &s = g->slabs[i];
      NSK_HIP(hipSetDevice(s.device));
      NSK_HIP(hipStreamWaitEvent(s.ctx->stream, s.ev_comm, 0));
      if (i > 0) NSK_HIP(hipStreamWaitEvent(s.ctx->stream, g->slabs[i - 1].ev_comm, 0));
      if (i < n - 1) NSK_HIP(hipStreamWaitEvent(s.ctx->stream, g->slabs[i + 1].ev_comm, 0));
    }
  }
  g->exchanges++;
  return 0;
}

// Fields `which` (bit 0 Pr, bit 1 dPrdτ, bit 2 ∇V), `halo` rows.
int exchange_fields(nsk_slabs *g, int which, bool from_edge = false) {
  XField xf[3];
  int nf = 0;
  for (int f = 0; f < 3; f++)
    if (which & (1 << f)) xf[nf++] = xfield(g, f);
  return exchange_list(g, xf, nf, g->halo, from_edge);
}

// Maximum over all slabs of the world of a non-negative double given as its bit pattern (the integer
// order is the IEEE order and NaN stays on top).  `m` holds the maximum over the local slabs.
int allreduce_max(nsk_slabs *g, unsigned long long *m) {
  if (g->transport != TRANSPORT_RCCL) return 0;  // all slabs are local
  const Rccl *R = rccl();
  g->red_host[0] = *m;
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_HIP(hipMemcpyAsync(s.red, g->red_host, sizeof *m, hipMemcpyHostToDevice, s.comm_stream));
  }
  GroupGuard group(R);
  NSK_NCCL(group.start());
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_NCCL(R->AllReduce(s.red, s.red, 1, ncclUint64, ncclMax, s.comm, s.comm_stream));
  }
  NSK_NCCL(group.end());
  Slab &s0 = g->slabs[0];
  NSK_HIP(hipSetDevice(s0.device));
  NSK_HIP(hipMemcpyAsync(g->red_host + 1, s0.red, sizeof *m, hipMemcpyDeviceToHost, s0.comm_stream));
  for (Slab &s : g->slabs)
    if (int rc = wait_stream(g, s, s.comm_stream, "all-reduce of the residual")) return rc;
  g->contact_made = g->contact_made || g->exchanges > 0;
  *m = g->red_host[1];
  return 0;
}

void release(nsk_slabs *g) {
  for (Slab &s : g->slabs) {
    (void)hipSetDevice(s.device);
    if (s.ctx) (void)hipStreamSynchronize(s.ctx->stream);
    if (s.comm_stream) (void)hipStreamSynchronize(s.comm_stream);
    if (s.edge_stream) {
      (void)hipStreamSynchronize(s.edge_stream);
      (void)hipStreamDestroy(s.edge_stream);
    }
    if (s.comm && rccl()) (void)rccl()->CommDestroy(s.comm);
    for (double *p : {s.Pr, s.dP, s.divV, s.Vx, s.Vy, s.C, s.Vx_o, s.Vy_o, s.C_o, s.txx, s.tyy, s.txy, s.Vprof})
      if (p) (void)hipFree(p);
    if (s.red) (void)hipFree(s.red);
    if (s.ev_done) (void)hipEventDestroy(s.ev_done);
    if (s.ev_comm) (void)hipEventDestroy(s.ev_comm);
    if (s.comm_stream) (void)hipStreamDestroy(s.comm_stream);
    if (s.ctx) (void)nsk_destroy(s.ctx);
  }
  if (g->red_host) (void)hipHostFree(g->red_host);
  delete g;
}

#define NSK_GROUP(g, local)                                         \
  NSK_REQUIRE((g) != nullptr, "null slab group");                   \
  NSK_REQUIRE((local) >= 0 && (local) < (g)->nlocal, "local slab index %d outside 0..%d", (local), (g)->nlocal - 1)

}  // namespace

extern "C" int nsk_rccl_unique_id(void *id128) {
  NSK_REQUIRE(id128 != nullptr, "nsk_rccl_unique_id: null buffer");
  const Rccl *R = rccl();
  NSK_REQUIRE(R != nullptr, "RCCL could not be loaded: %s", rccl_state().why);
  static_assert(sizeof(ncclUniqueId) == 128, "the ABI documents a 128-byte id");
  ncclUniqueId id;
  NSK_NCCL(R->GetUniqueId(&id));
  memcpy(id128, &id, sizeof id);
  return 0;
}

extern "C" int nsk_slabs_create(int world, int first_rank, int nlocal, const int *devices,
                                const nsk_ns_params *global, int ny_loc, int halo, int transport,
                                const void *unique_id, nsk_slabs **out) {
  NSK_REQUIRE(out != nullptr, "nsk_slabs_create: out is null");
  *out = nullptr;
  NSK_REQUIRE(global != nullptr && devices != nullptr, "nsk_slabs_create: null argument");
  NSK_REQUIRE(world >= 1 && nlocal >= 1 && first_rank >= 0 && first_rank + nlocal <= world,
              "slabs %d..%d outside a world of %d", first_rank, first_rank + nlocal - 1, world);
  NSK_REQUIRE(global->nx >= 3 && ny_loc >= 3 && global->ny == ny_loc * world,
              "global grid has %d rows, %d slabs of %d rows were asked for", global->ny, world, ny_loc);
  if (world == 1) halo = 0;
  NSK_REQUIRE(world == 1 || (halo >= 1 && halo <= ny_loc), "halo %d must be 1..%d rows", halo, ny_loc);
  if (transport == TRANSPORT_AUTO) transport = nlocal == world ? TRANSPORT_COPY : TRANSPORT_RCCL;
  NSK_REQUIRE(transport == TRANSPORT_RCCL || transport == TRANSPORT_COPY, "unknown transport %d", transport);
  NSK_REQUIRE(transport == TRANSPORT_RCCL || nlocal == world,
              "device copies need every slab in this process (%d of %d are)", nlocal, world);
  ncclUniqueId id;
  if (transport == TRANSPORT_RCCL) {
    NSK_REQUIRE(rccl() != nullptr, "RCCL could not be loaded: %s", rccl_state().why);
    if (unique_id) {
      memcpy(&id, unique_id, sizeof id);
    } else {
      NSK_REQUIRE(nlocal == world, "slabs in several processes need the id of nsk_rccl_unique_id from one of them");
      NSK_NCCL(rccl()->GetUniqueId(&id));
    }
  }
  nsk_slabs *g = new nsk_slabs();
  g->world = world;
  g->first = first_rank;
  g->nlocal = nlocal;
  g->ny_loc = ny_loc;
  g->halo = halo;
  g->ld = pitch_of(global->nx);
  g->transport = transport;
  g->global = *global;
  g->slabs.resize(nlocal);
#define NSK_FAIL_IF(cond, ...)  \
  do {                          \
    if (cond) {                 \
      nsk_set_error(__VA_ARGS__); \
      release(g);               \
      return 1;                 \
    }                           \
  } while (0)
  NSK_FAIL_IF(hipHostMalloc(&g->red_host, 64, hipHostMallocDefault) != hipSuccess, "pinned allocation failed");
  for (int i = 0; i < nlocal; i++) {
    Slab &s = g->slabs[i];
    s.rank = first_rank + i;
    s.device = devices[i];
    s.lo_halo = s.rank > 0 ? halo : 0;
    s.hi_halo = s.rank < world - 1 ? halo : 0;
    s.iy0 = s.rank * ny_loc - s.lo_halo;
    s.ny_win = ny_loc + s.lo_halo + s.hi_halo;
    s.own_lo = s.lo_halo;
    s.own_hi = s.lo_halo + ny_loc;
    s.prm = *global;
    s.prm.ny = s.ny_win;
    s.prm.iy0 = s.iy0;
    s.prm.nyg = ny_loc * world;
    s.prm.ld_pr = g->ld;
    s.prm.ld_dp = 0;
    s.prm.dp_embedded = 1;
    NSK_FAIL_IF(nsk_create(s.device, &s.ctx) != 0, "slab %d: %s", s.rank, nsk_last_error());
    const size_t bytes = (size_t)s.ny_win * g->ld * sizeof(double);
    bool ok = hipStreamCreateWithFlags(&s.comm_stream, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&s.edge_stream, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s.ev_comm, hipEventDisableTiming) == hipSuccess &&
              hipMalloc(&s.red, 64) == hipSuccess;
    for (double **p : {&s.Pr, &s.dP, &s.divV})
      ok = ok && hipMalloc(p, bytes) == hipSuccess && hipMemsetAsync(*p, 0, bytes, s.ctx->stream) == hipSuccess;
    NSK_FAIL_IF(!ok, "slab %d: stream, event or %zu-byte field allocation failed", s.rank, bytes);
  }
  if (transport == TRANSPORT_COPY) {
    for (int i = 0; i + 1 < nlocal; i++) {  // neighbours on different devices copy peer to peer
      const int a = g->slabs[i].device, b = g->slabs[i + 1].device;
      if (a == b) continue;
      int can = 0;
      NSK_FAIL_IF(hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can, "devices %d and %d cannot access each other", a, b);
      for (int k = 0; k < 2; k++) {
        (void)hipSetDevice(k ? b : a);
        hipError_t e = hipDeviceEnablePeerAccess(k ? a : b, 0);
        NSK_FAIL_IF(e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled, "enabling peer access %d <-> %d failed", a, b);
      }
    }
    (void)hipGetLastError();
  } else {
    const Rccl *R = rccl();
    bool ok = R->GroupStart() == ncclSuccess;
    for (Slab &s : g->slabs) {
      ok = ok && hipSetDevice(s.device) == hipSuccess;
      ok = ok && R->CommInitRank(&s.comm, world, id, s.rank) == ncclSuccess;
    }
    ok = (R->GroupEnd() == ncclSuccess) && ok;
    NSK_FAIL_IF(!ok, "RCCL communicator set-up failed for slabs %d..%d of %d", first_rank, first_rank + nlocal - 1, world);
  }
#undef NSK_FAIL_IF
  *out = g;
  return 0;
}

extern "C" int nsk_slabs_destroy(nsk_slabs *g) {
  if (g) release(g);
  return 0;
}

extern "C" int nsk_slabs_window(nsk_slabs *g, int local, int *rank, int *device, int *iy0, int *ny_win,
                                int *own_lo, int *own_hi, int *ld) {
  NSK_GROUP(g, local);
  const Slab &s = g->slabs[local];
  if (rank) *rank = s.rank;
  if (device) *device = s.device;
  if (iy0) *iy0 = s.iy0;
  if (ny_win) *ny_win = s.ny_win;
  if (own_lo) *own_lo = s.own_lo;
  if (own_hi) *own_hi = s.own_hi;
  if (ld) *ld = g->ld;
  return 0;
}

extern "C" int nsk_slabs_fields(nsk_slabs *g, int local, double **Pr, double **dPrdtau, double **divV) {
  NSK_GROUP(g, local);
  const Slab &s = g->slabs[local];
  if (Pr) *Pr = s.Pr;
  if (dPrdtau) *dPrdtau = s.dP;
  if (divV) *divV = s.divV;
  return 0;
}

// field: 0 Pr (nx x nyg), 1 dPrdτ ((nx-2) x (nyg-2), the reference's array), 2 ∇V (nx x nyg)
extern "C" int nsk_slabs_upload(nsk_slabs *g, int field, const double *host) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(host != nullptr && field >= 0 && field <= 5, "nsk_slabs_upload: bad arguments");
  NSK_REQUIRE(field <= 2 || g->flow, "nsk_slabs_upload: field %d needs nsk_slabs_init_flow first", field);
  const int nx = g->global.nx, nyg = g->ny_loc * g->world;
  const size_t w = sizeof(double);
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    if (field >= 3) {  // dense windows of Vx (nx+1 x nyg), Vy (nx x nyg+1), C (nx x nyg), halo rows included
      const size_t n1 = field == 3 ? nx + 1 : nx, rows = field == 4 ? s.ny_win + 1 : s.ny_win;
      double *dst = field == 3 ? s.Vx : field == 4 ? s.Vy : s.C;
      NSK_HIP(hipMemcpyAsync(dst, host + (size_t)s.iy0 * n1, rows * n1 * w, hipMemcpyHostToDevice, s.ctx->stream));
    } else if (field == 1) {  // value of pressure cell (ix, iy) goes to the offset of that cell; the rim is unused
      const int a = s.iy0 > 1 ? s.iy0 : 1, b = (s.iy0 + s.ny_win < nyg - 1) ? s.iy0 + s.ny_win : nyg - 1;  // Pr rows [a, b)
      NSK_HIP(hipMemcpy2DAsync(s.dP + (size_t)(a - s.iy0) * g->ld + 1, g->ld * w, host + (size_t)(a - 1) * (nx - 2),
                               (size_t)(nx - 2) * w, (size_t)(nx - 2) * w, b - a, hipMemcpyHostToDevice, s.ctx->stream));
    } else {
      double *dst = field == 0 ? s.Pr : s.divV;
      NSK_HIP(hipMemcpy2DAsync(dst, g->ld * w, host + (size_t)s.iy0 * nx, (size_t)nx * w, (size_t)nx * w, s.ny_win,
                               hipMemcpyHostToDevice, s.ctx->stream));
    }
  }
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_HIP(hipStreamSynchronize(s.ctx->stream));
  }
  return 0;
}

// Writes the rows the local slabs own into the global host array (other rows are left alone).
extern "C" int nsk_slabs_download(nsk_slabs *g, int field, double *host) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(host != nullptr && field >= 0 && field <= 5, "nsk_slabs_download: bad arguments");
  NSK_REQUIRE(field <= 2 || g->flow, "nsk_slabs_download: field %d needs nsk_slabs_init_flow first", field);
  const int nx = g->global.nx, nyg = g->ny_loc * g->world;
  const size_t w = sizeof(double);
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    const int g0 = s.iy0 + s.own_lo, g1 = s.iy0 + s.own_hi;  // global rows [g0, g1)
    if (field >= 3) {  // owned rows; of Vy the owned faces, the top slab one more
      const size_t n1 = field == 3 ? nx + 1 : nx;
      const int rows = g1 - g0 + ((field == 4 && s.rank == g->world - 1) ? 1 : 0);
      const double *src = field == 3 ? s.Vx : field == 4 ? s.Vy : s.C;
      NSK_HIP(hipMemcpyAsync(host + (size_t)g0 * n1, src + (size_t)s.own_lo * n1, (size_t)rows * n1 * w,
                             hipMemcpyDeviceToHost, s.ctx->stream));
    } else if (field == 1) {
      const int a = g0 > 1 ? g0 : 1, b = g1 < nyg - 1 ? g1 : nyg - 1;
      NSK_HIP(hipMemcpy2DAsync(host + (size_t)(a - 1) * (nx - 2), (size_t)(nx - 2) * w,
                               s.dP + (size_t)(a - s.iy0) * g->ld + 1, g->ld * w, (size_t)(nx - 2) * w, b - a,
                               hipMemcpyDeviceToHost, s.ctx->stream));
    } else {
      const double *src = field == 0 ? s.Pr : s.divV;
      NSK_HIP(hipMemcpy2DAsync(host + (size_t)g0 * nx, (size_t)nx * w, src + (size_t)s.own_lo * g->ld, g->ld * w,
                               (size_t)nx * w, g1 - g0, hipMemcpyDeviceToHost, s.ctx->stream));
    }
  }
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_HIP(hipStreamSynchronize(s.ctx->stream));
  }
  return 0;
}

extern "C" int nsk_slabs_exchange(nsk_slabs *g, int field_mask) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(field_mask >= 0 && field_mask <= 7, "field mask %d outside 0..7", field_mask);
  return exchange_fields(g, field_mask);
}

extern "C" int nsk_slabs_sync(nsk_slabs *g) {
  NSK_GROUP(g, 0);
  for (Slab &s : g->slabs) {
    if (int rc = wait_stream(g, s, s.comm_stream, "nsk_slabs_sync (communication stream)")) return rc;
    if (int rc = wait_stream(g, s, s.edge_stream, "nsk_slabs_sync (edge stream)")) return rc;
    if (int rc = wait_stream(g, s, s.ctx->stream, "nsk_slabs_sync (kernel stream)")) return rc;
  }
  g->contact_made = g->contact_made || g->exchanges > 0;
  return 0;
}

extern "C" int nsk_slabs_poisson_iters(nsk_slabs *g, int count) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(count >= 0, "negative pass count %d", count);
  int left = count;
  while (left > 0) {
    const int n = g->world == 1 ? left : (left < g->halo ? left : g->halo);
    // rows to be sent first, halo rows (about to be received) not at all in the block's last launch,
    // when a slab has rows left between its two edges
    const bool ef = g->edges_first && g->world > 1 && g->ny_loc > 2 * g->halo;
    for (Slab &s : g->slabs) {
      const int rc = ef ? nsk_ns_poisson_iters_edges_first(s.ctx, s.Pr, s.dP, s.divV, &s.prm, n, s.lo_halo,
                                                           s.lo_halo ? g->halo : 0, s.hi_halo ? g->halo : 0,
                                                           s.hi_halo, s.edge_stream)
                        : nsk_ns_poisson_iters(s.ctx, s.Pr, s.dP, s.divV, &s.prm, n);
      if (rc) return rc;
    }
    if (int rc = exchange_fields(g, 3, ef)) return rc;
    left -= n;
  }
  return 0;
}

// max|Rp| over the whole grid (NavierStokes.jl:89-90): local maxima over owned rows, then the
// maximum of the bit patterns (|x| >= 0, so the integer order is the IEEE order and NaN stays on top).
extern "C" int nsk_slabs_residual(nsk_slabs *g, double *maxabs_out) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(maxabs_out != nullptr, "nsk_slabs_residual: maxabs_out is null");
  unsigned long long m = 0;
  // the kernel streams are ordered behind the last exchange: wait for them here, where a transfer that
  // never completes ends in an error (nsk_ns_residual itself waits for good)
  for (Slab &s : g->slabs)
    if (int rc = wait_stream(g, s, s.ctx->stream, "residual (kernel stream, behind the last exchange)")) return rc;
  for (Slab &s : g->slabs) {
    double v = 0.0;
    if (int rc = nsk_ns_residual(s.ctx, nullptr, s.Pr, s.divV, &s.prm, s.own_lo + 1, s.own_hi, &v)) return rc;
    unsigned long long b;
    v = std::fabs(v);
    memcpy(&b, &v, sizeof b);
    m = b > m ? b : m;
  }
  if (int rc = allreduce_max(g, &m)) return rc;  // (also with one slab: the reduction is then the identity)
  memcpy(maxabs_out, &m, sizeof m);
  return 0;
}

// The Poisson loop NavierStokes.jl:84-95 on the slab-decomposed grid.
extern "C" int nsk_slabs_poisson(nsk_slabs *g, int *iters_out, double *err_hist, int *iter_hist, int hist_cap,
                                 int *nhist_out) {
  NSK_GROUP(g, 0);
  const nsk_ns_params *p = &g->global;
  const int niter = p->niter, nchk = p->nchk;
  int done = 0, nh = 0;
  while (done < niter) {
    int chunk = nchk > 0 ? nchk - (done % nchk) : niter - done;
    if (chunk > niter - done) chunk = niter - done;
    if (int rc = nsk_slabs_poisson_iters(g, chunk)) return rc;
    done += chunk;
    if (nchk > 0 && done % nchk == 0) {
      double m = 0.0;
      if (int rc = nsk_slabs_residual(g, &m)) return rc;
      const double err = m * (p->ly * p->ly) / p->psc;  // NavierStokes.jl:90
      if (nh < hist_cap) {
        if (err_hist) err_hist[nh] = err;
        if (iter_hist) iter_hist[nh] = done;
      }
      nh++;
      if (err < p->eps_it || !std::isfinite(err)) break;  // :93
    }
  }
  if (int rc = nsk_slabs_sync(g)) return rc;
  if (iters_out) *iters_out = done;
  if (nhist_out) *nhist_out = nh;
  return 0;
}

// Transport check that needs no second GPU: every local slab sends the first `rows` owned rows of
// Pr to ITSELF through RCCL and receives them into the first `rows` owned rows of divV, with the
// same stream and event ordering as a halo exchange.
extern "C" int nsk_slabs_rccl_self_exchange(nsk_slabs *g, int rows) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(g->transport == TRANSPORT_RCCL, "the group does not use RCCL");
  NSK_REQUIRE(rows >= 1 && rows <= g->ny_loc, "rows %d outside 1..%d", rows, g->ny_loc);
  const Rccl *R = rccl();
  const size_t ld = (size_t)g->ld, count = (size_t)rows * ld;
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_HIP(hipEventRecord(s.ev_done, s.ctx->stream));
    NSK_HIP(hipStreamWaitEvent(s.comm_stream, s.ev_done, 0));
  }
  GroupGuard group(R);
  NSK_NCCL(group.start());
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_NCCL(R->Send(s.Pr + (size_t)s.own_lo * ld, count, ncclDouble, s.rank, s.comm, s.comm_stream));
    NSK_NCCL(R->Recv(s.divV + (size_t)s.own_lo * ld, count, ncclDouble, s.rank, s.comm, s.comm_stream));
  }
  NSK_NCCL(group.end());
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    NSK_HIP(hipEventRecord(s.ev_comm, s.comm_stream));
    NSK_HIP(hipStreamWaitEvent(s.ctx->stream, s.ev_comm, 0));
  }
  return 0;
}

// Velocity and tracer windows of runme() (NavierStokes.jl:58-72) on every local slab, all zero
// except the inflow row Vy[:,1] = Vprof (:73) on the bottom slab.  Vprof: nx host values.
extern "C" int nsk_slabs_init_flow(nsk_slabs *g, const double *Vprof_host) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(Vprof_host != nullptr, "nsk_slabs_init_flow: Vprof is null");
  NSK_REQUIRE(g->world == 1 || g->halo >= 4, "a slab time step needs at least 4 halo rows, got %d", g->halo);
  const size_t nx = (size_t)g->global.nx, w = sizeof(double);
  for (Slab &s : g->slabs) {
    NSK_HIP(hipSetDevice(s.device));
    const size_t nw = (size_t)s.ny_win;
    struct { double **p; size_t n; } arr[] = {
        {&s.Vx, (nx + 1) * nw}, {&s.Vx_o, (nx + 1) * nw}, {&s.Vy, nx * (nw + 1)}, {&s.Vy_o, nx * (nw + 1)},
        {&s.C, nx * nw},        {&s.C_o, nx * nw},        {&s.txx, nx * nw},      {&s.tyy, nx * nw},
        {&s.txy, (nx - 1) * (nw - 1)}, {&s.Vprof, nx}};
    for (auto &a : arr) {
      if (!*a.p) NSK_HIP(hipMalloc(a.p, a.n * w));
      NSK_HIP(hipMemsetAsync(*a.p, 0, a.n * w, s.ctx->stream));
    }
    NSK_HIP(hipMemcpyAsync(s.Vprof, Vprof_host, nx * w, hipMemcpyHostToDevice, s.ctx->stream));
    if (s.iy0 == 0) NSK_HIP(hipMemcpyAsync(s.Vy, Vprof_host, nx * w, hipMemcpyHostToDevice, s.ctx->stream));
    NSK_HIP(hipStreamSynchronize(s.ctx->stream));
  }
  g->flow = true;
  return 0;
}

// One pass of the time-loop body NavierStokes.jl:79-100 on the slab-decomposed grid.  The kernels
// act on whole windows with their guards on global rows; a kernel that needs a neighbour row leaves
// the window's outermost halo row stale, so validity shrinks by one row per dependent kernel and is
// restored by exchanges: ∇V before the Poisson loop (which needs all `halo` rows), Pr and dPrdτ
// inside it, and Vx, Vy, C after advect!.  The back-trace of advect! reaches |V| dt / d + 1 rows; a
// max|Vy| check over all slabs makes a too narrow halo an error instead of a wrong answer.
extern "C" int nsk_slabs_step(nsk_slabs *g, int *iters_out, double *err_hist, int *iter_hist, int hist_cap,
                              int *nhist_out) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(g->flow, "nsk_slabs_step: call nsk_slabs_init_flow first");
  const nsk_ns_params *p = &g->global;
  const int nx = p->nx, nyg = g->ny_loc * g->world;
#define NSK_EACH(call)      \
  for (Slab &s : g->slabs) \
    if (int rc_ = (call)) return rc_
  // rows the back-trace of advect! may reach, |Vy| dt/dy + 1 below / above (one halo row is stale by then:
  // predict_V! / correct_V! do not update a window's outermost row)
  auto reach_of_advect = [&](double *reach, bool corrected_in_o = false) -> int {
    unsigned long long m = 0;
    for (Slab &s : g->slabs) {
      double v = 0.0;
      if (int rc = nsk_maxabs(s.ctx, corrected_in_o ? s.Vy_o : s.Vy, (size_t)nx * (s.ny_win + 1), &v)) return rc;
      unsigned long long b;
      memcpy(&b, &v, sizeof b);
      m = b > m ? b : m;
    }
    if (int rc = allreduce_max(g, &m)) return rc;
    double vmax;
    memcpy(&vmax, &m, sizeof vmax);
    *reach = vmax * p->dt / p->dy + 2;
    return 0;
  };
  if (g->world > 1) {
    // with the velocities as they are on entry: a halo that is too narrow already is reported before
    // anything has been changed
    double reach = 0.0;
    if (int rc = reach_of_advect(&reach)) return rc;
    NSK_REQUIRE(reach <= g->halo - 1, "advect! would reach %.1f rows but only %d valid halo rows exist: raise the halo width "
                "(nothing was changed)", reach, g->halo - 1);
  }
  if (g->step_fused) {
    // Three kernels around the Poisson loop (ns_step_fused.hip), on windows: the predictor leaves the new
    // fields in the *_o arrays, the corrector works there in place, advect! brings them back.
    auto state_of = [](const Slab &s) {
      nsk_ns_state st{};
      st.Pr = s.Pr, st.dPrdtau = s.dP, st.divV = s.divV, st.Vprof = s.Vprof;
      st.Vx = s.Vx, st.Vy = s.Vy, st.C = s.C, st.Vx_o = s.Vx_o, st.Vy_o = s.Vy_o, st.C_o = s.C_o;
      return st;
    };
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      const nsk_ns_state st = state_of(s);
      if (int rc = nsk_i_step_predictor(s.ctx, &st, &s.prm)) return rc;  // :79-82
    }
    if (int rc = exchange_fields(g, 4)) return rc;
    if (int rc = nsk_slabs_poisson(g, iters_out, err_hist, iter_hist, hist_cap, nhist_out)) return rc;  // :84-95
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      const nsk_ns_state st = state_of(s);
      if (int rc = nsk_i_step_corrector(s.ctx, &st, &s.prm)) return rc;  // :96-98
    }
    if (g->world > 1) {
      double reach = 0.0;
      if (int rc = reach_of_advect(&reach, true)) return rc;
      NSK_REQUIRE(reach <= g->halo - 1, "advect! reaches %.1f rows but only %d valid halo rows exist: the windows hold a "
                  "half-finished time step; set the group up again with a wider halo", reach, g->halo - 1);
    }
    for (Slab &s : g->slabs) {
      NSK_HIP(hipSetDevice(s.device));
      const nsk_ns_state st = state_of(s);
      if (int rc = nsk_i_step_advect_out(s.ctx, &st, &s.prm)) return rc;  // :99-100
    }
  } else {
    // :79-82
    NSK_EACH(nsk_update_tau(s.ctx, s.txx, s.tyy, s.txy, s.Vx, s.Vy, p->mu, p->dx, p->dy, nx, s.ny_win));
    NSK_EACH(nsk_predict_V(s.ctx, s.Vx, s.Vy, s.txx, s.tyy, s.txy, p->rho, p->g, p->dt, p->dx, p->dy, nx, s.ny_win));
    NSK_EACH(nsk_set_sphere(s.ctx, s.C, s.Vx, s.Vy, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx, p->ly, p->dx,
                            p->dy, nx, s.ny_win, s.iy0));
    NSK_EACH(nsk_update_divV_ld(s.ctx, s.divV, g->ld, s.Vx, s.Vy, p->dx, p->dy, nx, s.ny_win));
    if (int rc = exchange_fields(g, 4)) return rc;
    // :84-95
    if (int rc = nsk_slabs_poisson(g, iters_out, err_hist, iter_hist, hist_cap, nhist_out)) return rc;
    // :96-98
    NSK_EACH(nsk_correct_V_ld(s.ctx, s.Vx, s.Vy, s.Pr, g->ld, p->dt, p->rho, p->dx, p->dy, nx, s.ny_win));
    NSK_EACH(nsk_set_sphere(s.ctx, s.C, s.Vx, s.Vy, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx, p->ly, p->dx,
                            p->dy, nx, s.ny_win, s.iy0));
    NSK_EACH(nsk_set_bc_Vel(s.ctx, s.Vx, s.Vy, s.Vprof, nx, s.ny_win, p->full_rows, s.iy0, nyg));
    if (g->world > 1) {
      double reach = 0.0;
      if (int rc = reach_of_advect(&reach)) return rc;
      NSK_REQUIRE(reach <= g->halo - 1, "advect! reaches %.1f rows but only %d valid halo rows exist: the windows hold a "
                  "half-finished time step; set the group up again with a wider halo", reach, g->halo - 1);
    }
    // :99-100
    NSK_EACH(nsk_copy(s.ctx, s.Vx_o, s.Vx, (size_t)(nx + 1) * s.ny_win));
    NSK_EACH(nsk_copy(s.ctx, s.Vy_o, s.Vy, (size_t)nx * (s.ny_win + 1)));
    NSK_EACH(nsk_copy(s.ctx, s.C_o, s.C, (size_t)nx * s.ny_win));
    NSK_EACH(nsk_advect(s.ctx, s.Vx, s.Vx_o, s.Vy, s.Vy_o, s.C, s.C_o, p->dt, p->dx, p->dy, nx, s.ny_win, s.iy0, nyg));
  }
#undef NSK_EACH
  const XField xf[3] = {xfield(g, F_VX), xfield(g, F_C), xfield(g, F_VY)};
  if (int rc = exchange_list(g, xf, 3, g->halo)) return rc;
  return nsk_slabs_sync(g);
}

extern "C" int nsk_slabs_set_option(nsk_slabs *g, const char *name, int value) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(name != nullptr, "nsk_slabs_set_option: name is null");
  if (!strcmp(name, "edges_first")) {
    g->edges_first = value != 0;
    return 0;
  }
  if (!strcmp(name, "step_fused")) {
    g->step_fused = value != 0;
    return 0;
  }
  if (!strcmp(name, "deadline_ms") || !strcmp(name, "first_contact_ms") || !strcmp(name, "test_stall_ms")) {
    NSK_REQUIRE(value >= 0, "%s: %d", name, value);
    (name[0] == 'd' ? g->deadline_ms : name[0] == 'f' ? g->first_contact_ms : g->test_stall_ms) = value;
    return 0;
  }
  nsk_set_error("nsk_slabs_set_option: unknown option '%s'", name);
  return 2;
}

extern "C" int nsk_slabs_get_option(nsk_slabs *g, const char *name, int *value) {
  NSK_GROUP(g, 0);
  NSK_REQUIRE(name != nullptr && value != nullptr, "nsk_slabs_get_option: null argument");
  if (!strcmp(name, "edges_first")) *value = g->edges_first;
  else if (!strcmp(name, "step_fused")) *value = g->step_fused;
  else if (!strcmp(name, "deadline_ms")) *value = g->deadline_ms;
  else if (!strcmp(name, "first_contact_ms")) *value = g->first_contact_ms;
  else return nsk_get_option(g->slabs[0].ctx, name, value);  // what the first local slab's kernels use
  return 0;
}

extern "C" int nsk_slabs_info(nsk_slabs *g, int *world, int *nlocal, int *transport, int *exchanges) {
  NSK_GROUP(g, 0);
  if (world) *world = g->world;
  if (nlocal) *nlocal = g->nlocal;
  if (transport) *transport = g->transport;
  if (exchanges) *exchanges = g->exchanges;
  return 0;
}
