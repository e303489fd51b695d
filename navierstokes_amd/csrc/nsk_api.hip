// Context, error, stream and memory entry points of libnsk.
#include <cstdlib>
#include <cstring>

#include "nsk_internal.h"

static thread_local char g_err[512] = "";

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
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    nsk_set_error("nsk_create: hipStreamCreate failed");
    delete c;
    return 1;
  }
  c->stream = c->own_stream;
  if (hipMalloc(&c->red_dev, 64) != hipSuccess ||
      hipHostMalloc(&c->red_host, 64, hipHostMallocDefault) != hipSuccess) {
    nsk_set_error("nsk_create: scratch allocation failed");
    nsk_destroy(c);
    return 1;
  }
  const char *k = getenv("NSK_POISSON_K");
  if (k) c->poisson_k = atoi(k);
  *out = c;
  return 0;
}

extern "C" int nsk_destroy(nsk_ctx *c) {
  if (!c) return 0;
  hipSetDevice(c->device);
  if (c->own_stream) {
    hipStreamSynchronize(c->own_stream);
    hipStreamDestroy(c->own_stream);
  }
  if (c->red_dev) hipFree(c->red_dev);
  if (c->red_host) hipHostFree(c->red_host);
  if (c->pp_pr) hipFree(c->pp_pr);
  if (c->pp_dp) hipFree(c->pp_dp);
  delete c;
  return 0;
}

extern "C" int nsk_set_stream(nsk_ctx *ctx, void *hip_stream) {
  NSK_REQUIRE(ctx != nullptr, "null context");
  ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  return 0;
}

extern "C" int nsk_sync(nsk_ctx *ctx) {
  NSK_CTX(ctx);
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
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
  NSK_HIP(hipMemcpyAsync(dev, host, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int nsk_download(nsk_ctx *ctx, double *host, const double *dev, size_t count) {
  NSK_CTX(ctx);
  NSK_HIP(hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NSK_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

extern "C" int nsk_copy(nsk_ctx *ctx, double *dst, const double *src, size_t count) {
  NSK_CTX(ctx);
  NSK_HIP(hipMemcpyAsync(dst, src, count * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
