// Context, error, stream and memory entry points of libnsk.
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "nsk_internal.h"

static thread_local char g_err[512] = "";

int nsk_i_static_scratch_init(int device, double **out);

void nsk_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char *nsk_last_error(void) { return g_err; }
extern "C" const char *nsk_version(void) { return "libnsk 0.1 (gfx950)"; }

extern "C" int nsk_create(int device, nsk_ctx **out) {
  NSK_REQUIRE(out != nullptr, "nsk_create: out is null");
  *out = nullptr;
  int ndev = 0;
  NSK_HIP(hipGetDeviceCount(&ndev));
  NSK_REQUIRE(device >= 0 && device < ndev, "nsk_create: device %d out of range (0..%d)",
              device, ndev - 1);
  NSK_HIP(hipSetDevice(device));
  nsk_ctx *c = new nsk_ctx();
  c->device = device;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
    c->num_simds = 4 * cus;
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    nsk_set_error("nsk_create: hipStreamCreate failed");
    delete c;
    return 1;
  }
  c->stream = c->own_stream;
  if (hipMalloc(&c->red_dev, 256) != hipSuccess ||
      hipHostMalloc(&c->red_host, 256, hipHostMallocDefault) != hipSuccess) {
    nsk_set_error("nsk_create: scratch allocation failed");
    nsk_destroy(c);
    return 1;
  }
  memset(c->red_host, 0, 256);
  // (a failure here only disables nsk_ns_poisson_iters_static, which then reports it)
  if (nsk_i_static_scratch_init(device, &c->static_scratch) != 0) c->static_scratch = nullptr;
  if (const char *k = getenv("NSK_POISSON_K")) {
    const int v = atoi(k);
    if (v >= 0 && v <= 8) c->poisson_k = v;
  }
  if (const char *q = getenv("NSK_POISSON_PAIRS")) {  // (A/B runs of programs that set no options)
    const int v = atoi(q);
    if (v >= 0 && v <= 3) c->poisson_pairs = v;
  }
  if (const char *r = getenv("NSK_POISSON_ROWS")) {
    const int v = atoi(r);
    if (v >= 0) c->poisson_rows = v;
  }
  *out = c;
  return 0;
}

hipError_t nsk_allow_lds(nsk_ctx *ctx, const void *kernel, size_t bytes) {
  struct Key {
    const void *kernel;
    int device;
    bool operator<(const Key &o) const { return kernel != o.kernel ? kernel < o.kernel : device < o.device; }
  };
  static std::mutex mu;
  static std::map<Key, size_t> allowed;
  std::lock_guard<std::mutex> lock(mu);
  size_t &have = allowed[Key{kernel, ctx->device}];
  if (have >= bytes) return hipSuccess;
  const hipError_t rc = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (rc == hipSuccess) have = bytes;
  return rc;
}

extern "C" int nsk_destroy(nsk_ctx *c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  if (c->own_stream) {
    (void)hipStreamSynchronize(c->own_stream);
    (void)hipStreamDestroy(c->own_stream);
  }
  if (c->red_dev) (void)hipFree(c->red_dev);
  if (c->red_host) (void)hipHostFree(c->red_host);
  if (c->pp_pr) (void)hipFree(c->pp_pr);
  if (c->pp_dp) (void)hipFree(c->pp_dp);
  if (c->stage) (void)hipFree(c->stage);
  if (c->iter_pp) (void)hipFree(c->iter_pp);
  if (c->sor_snap) (void)hipFree(c->sor_snap);
  if (c->sor_grid_ctl) (void)hipFree(c->sor_grid_ctl);
  if (c->run_words) (void)hipFree(c->run_words);
  if (c->small_xbuf) (void)hipFree(c->small_xbuf);
  for (auto &kv : c->pair_tables) (void)hipFree(kv.second.first);
  if (c->ev_edge) (void)hipEventDestroy(c->ev_edge);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->hist_dev) (void)hipFree(c->hist_dev);
  if (c->hist_host) (void)hipHostFree(c->hist_host);
  delete c;
  return 0;
}

extern "C" int nsk_set_stream(nsk_ctx *ctx, void *hip_stream) {
  NSK_REQUIRE(ctx != nullptr, "null context");
  ctx->stream = (hipStream_t)hip_stream;  // NULL is HIP's default stream
  return 0;
}

extern "C" int nsk_use_own_stream(nsk_ctx *ctx) {
  NSK_REQUIRE(ctx != nullptr, "null context");
  ctx->stream = ctx->own_stream;
  return 0;
}

extern "C" int nsk_set_option(nsk_ctx *ctx, const char *name, int value) {
  NSK_REQUIRE(ctx != nullptr && name != nullptr, "nsk_set_option: null argument");
  if (!strcmp(name, "poisson_k")) {
    NSK_REQUIRE(value >= 0 && value <= 8, "poisson_k must be 0 (default) .. 8, got %d", value);
    ctx->poisson_k = value;
    return 0;
  }
  if (!strcmp(name, "poisson_rows")) {
    NSK_REQUIRE(value >= 0, "poisson_rows must be >= 0, got %d", value);
    ctx->poisson_rows = value;
    return 0;
  }
  if (!strcmp(name, "poisson_xcd")) {
    ctx->poisson_xcd = value != 0;
    return 0;
  }
  if (!strcmp(name, "poisson_stream")) {
    NSK_REQUIRE(value >= -1 && value <= 1, "poisson_stream must be -1 (by size), 0 or 1, got %d", value);
    ctx->poisson_stream = value;
    return 0;
  }
  if (!strcmp(name, "poisson_stage")) {
    ctx->poisson_stage = value != 0;
    return 0;
  }
  if (!strcmp(name, "step_fused")) {
    ctx->step_fused = value != 0;
    return 0;
  }
  if (!strcmp(name, "sor_pipeline")) {
    ctx->sor_pipeline = value != 0;
    return 0;
  }
  if (!strcmp(name, "sor_grid")) {
    ctx->sor_grid = value != 0;
    return 0;
  }
  if (!strcmp(name, "sor_grid_rows")) {
    NSK_REQUIRE(value >= 0 && value <= 64, "sor_grid_rows: %d (0 = default, else 4..64)", value);
    ctx->sor_grid_rows = value;
    return 0;
  }
  if (!strcmp(name, "sor_grid_copies")) {
    NSK_REQUIRE(value == 0 || value >= 3, "sor_grid_copies: %d (0 = by the grid, else >= 3)", value);
    ctx->sor_grid_copies = value;
    return 0;
  }
  if (!strcmp(name, "sor_fast_division")) {
    ctx->sor_fast_division = value != 0;
    return 0;
  }
  if (!strcmp(name, "bub_merge_launches")) {
    ctx->bub_merge_launches = value != 0;
    return 0;
  }
  if (!strcmp(name, "poisson_edge_task_rows")) {
    NSK_REQUIRE(value >= 0, "poisson_edge_task_rows must be >= 0, got %d", value);
    ctx->edge_task_rows = value;
    return 0;
  }
  if (!strcmp(name, "poisson_edge_percent")) {
    NSK_REQUIRE(value >= 0 && value <= 100, "poisson_edge_percent must be 0..100, got %d", value);
    ctx->poisson_edge_percent = value;
    return 0;
  }
  if (!strcmp(name, "poisson_small")) {
    NSK_REQUIRE(value == 0 || value == 1, "poisson_small must be 0 or 1, got %d", value);
    ctx->poisson_small = value;
    return 0;
  }
  if (!strcmp(name, "poisson_pairs")) {
    NSK_REQUIRE(value >= 0 && value <= 3, "poisson_pairs must be 0 .. 3, got %d", value);
    ctx->poisson_pairs = value;
    return 0;
  }
  if (!strcmp(name, "poisson_small_spin_limit") || !strcmp(name, "poisson_small_test_absent")) {
    NSK_REQUIRE(value >= 0, "%s must be >= 0, got %d", name, value);
    (name[14] == 's' ? ctx->poisson_small_spin_limit : ctx->poisson_small_test_absent) = value;
    return 0;
  }
  if (!strcmp(name, "poisson_small_halo")) {
    NSK_REQUIRE(value >= 0 && value <= 64, "poisson_small_halo must be 0..64, got %d", value);
    ctx->poisson_small_halo = value;
    return 0;
  }
  if (!strcmp(name, "poisson_small_rows") || !strcmp(name, "poisson_small_cols")) {
    NSK_REQUIRE(value >= 0, "%s must be >= 0, got %d", name, value);
    (name[14] == 'r' ? ctx->poisson_small_rows : ctx->poisson_small_cols) = value;
    return 0;
  }
  if (!strcmp(name, "poisson_tall_percent")) {
    NSK_REQUIRE(value >= -1 && value <= 95, "poisson_tall_percent must be -1..95, got %d", value);
    ctx->poisson_tall_percent = value;
    return 0;
  }
  if (!strcmp(name, "poisson_div")) {
    NSK_REQUIRE(value == -1 || value == 0 || value == 2 || value == 3 || value == 4,
                "poisson_div must be -1 (automatic), 0, 2, 3 or 4, got %d", value);
    ctx->poisson_div = value;
    return 0;
  }
  nsk_set_error("nsk_set_option: unknown option '%s'", name);
  return 2;
}

extern "C" int nsk_get_option(nsk_ctx *ctx, const char *name, int *value) {
  NSK_REQUIRE(ctx != nullptr && name != nullptr && value != nullptr, "nsk_get_option: null argument");
  if (!strcmp(name, "poisson_k")) {
    // forced value, else what the planner chose in the last fused Poisson call (5 before any)
    *value = ctx->poisson_k > 0 ? ctx->poisson_k : (ctx->last_plan_k > 0 ? ctx->last_plan_k : 5);
    return 0;
  }
  if (!strcmp(name, "poisson_rows")) {
    *value = ctx->poisson_rows;
    return 0;
  }
  if (!strcmp(name, "poisson_rows_used")) {
    *value = ctx->last_plan_rows;
    return 0;
  }
  if (!strcmp(name, "poisson_stage")) {
    *value = ctx->poisson_stage;
    return 0;
  }
  if (!strcmp(name, "step_fused")) {
    *value = ctx->step_fused;
    return 0;
  }
  if (!strcmp(name, "sor_pipeline")) {
    *value = ctx->sor_pipeline;
    return 0;
  }
  if (!strcmp(name, "sor_grid")) {
    *value = ctx->sor_grid;
    return 0;
  }
  if (!strcmp(name, "sor_grid_rows")) {
    *value = ctx->sor_grid_rows;
    return 0;
  }
  if (!strcmp(name, "sor_grid_copies")) {
    *value = ctx->sor_grid_copies;
    return 0;
  }
  if (!strcmp(name, "sor_fast_division")) {
    *value = ctx->sor_fast_division;
    return 0;
  }
  if (!strcmp(name, "bub_merge_launches")) {
    *value = ctx->bub_merge_launches;
    return 0;
  }
  if (!strcmp(name, "poisson_edge_task_rows")) {
    *value = ctx->edge_task_rows;
    return 0;
  }
  if (!strcmp(name, "poisson_xcd")) {
    *value = ctx->poisson_xcd;
    return 0;
  }
  if (!strcmp(name, "poisson_stream")) {
    *value = ctx->poisson_stream;
    return 0;
  }
  if (!strcmp(name, "poisson_edge_percent")) {
    *value = ctx->poisson_edge_percent;
    return 0;
  }
  if (!strcmp(name, "poisson_small")) {
    *value = ctx->poisson_small;
    return 0;
  }
  if (!strcmp(name, "poisson_pairs_used")) {
    *value = ctx->last_launch_pairs;
    return 0;
  }
  if (!strcmp(name, "poisson_pairs")) {
    *value = ctx->poisson_pairs;
    return 0;
  }
  if (!strcmp(name, "poisson_small_halo")) {
    *value = ctx->poisson_small_halo;
    return 0;
  }
  if (!strcmp(name, "poisson_small_rows") || !strcmp(name, "poisson_small_cols")) {
    *value = name[14] == 'r' ? ctx->poisson_small_rows : ctx->poisson_small_cols;
    return 0;
  }
  if (!strcmp(name, "poisson_tall_percent")) {
    *value = ctx->poisson_tall_percent;
    return 0;
  }
  if (!strcmp(name, "poisson_div")) {
    *value = ctx->poisson_div;
    return 0;
  }
  if (!strcmp(name, "poisson_div_used")) {
    *value = ctx->last_div;
    return 0;
  }
  nsk_set_error("nsk_get_option: unknown option '%s'", name);
  return 2;
}

int nsk_device_error_check(nsk_ctx *ctx, const char *where) {
  volatile unsigned long long *word = ctx->red_host + NSK_DEVICE_ERROR_WORD;
  if (*word == 0) return 0;
  *word = 0;
  nsk_set_error("%s: a kernel of this context gave up waiting for another workgroup of its launch (a bounded wait "
                "between workgroups ran out); the arrays that launch was writing hold undefined values",
                where);
  return NSK_ERR_TIMEOUT;
}

extern "C" int nsk_sync(nsk_ctx *ctx) {
  NSK_CTX(ctx);
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return nsk_device_error_check(ctx, "nsk_sync");
}

extern "C" int nsk_zeros(nsk_ctx *ctx, int n1, int n2, double **out) {
  NSK_CTX(ctx);
  NSK_REQUIRE(out != nullptr && n1 >= 0 && n2 >= 0, "nsk_zeros: bad arguments");
  size_t bytes = sizeof(double) * (size_t)n1 * (size_t)n2;
  if (bytes == 0) bytes = sizeof(double);
  NSK_HIP(hipMalloc(out, bytes));
  NSK_HIP(hipMemsetAsync(*out, 0, bytes, ctx->stream));
  return 0;
}

extern "C" int nsk_free(nsk_ctx *ctx, double *dev) {
  NSK_CTX(ctx);
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  NSK_HIP(hipFree(dev));
  return 0;
}

extern "C" int nsk_upload(nsk_ctx *ctx, double *dev, const double *host, size_t count) {
  NSK_CTX(ctx);
  if (count == 0) return 0;
  NSK_NONNULL(dev, host);
  NSK_HIP(hipMemcpyAsync(dev, host, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int nsk_download(nsk_ctx *ctx, double *host, const double *dev, size_t count) {
  NSK_CTX(ctx);
  if (count == 0) return 0;
  NSK_NONNULL(host, dev);
  NSK_HIP(hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return nsk_device_error_check(ctx, "nsk_download");
}

extern "C" int nsk_copy(nsk_ctx *ctx, double *dst, const double *src, size_t count) {
  NSK_CTX(ctx);
  if (count == 0) return 0;
  NSK_NONNULL(dst, src);
  NSK_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
