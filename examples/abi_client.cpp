// A caller of libnsk that is neither Python nor Julia: plain C++ on the C ABI of include/nsk.h, the way a
// compiled host framework would use it.  Reads a problem file (the parameter block followed by Pr, dPrdtau
// and divV as the reference keeps them: dense column-major Float64, dPrdtau (nx-2) x (ny-2)), runs `count`
// Poisson passes (scripts/NavierStokes.jl:85-87) and the residual (:89-90), writes Pr, dPrdtau and the
// residual maximum, and prints the time per pass.  tests/test_native_client_gpu.py compares the output
// with the oracle bit for bit.
//
//   abi_client <problem file> <result file> <count> [repetitions]
//
// problem file: int32 magic 0x4e534b31, int32 sizeof(nsk_ns_params), the struct, then the three arrays.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../include/nsk.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    if (int rc_ = (call)) {                                                          \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, nsk_last_error());    \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

static bool read_all(std::FILE *f, void *p, size_t bytes) { return std::fread(p, 1, bytes, f) == bytes; }

int main(int argc, char **argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <problem file> <result file> <count> [repetitions]\n", argv[0]);
    return 2;
  }
  const int count = std::atoi(argv[3]), reps = argc > 4 ? std::atoi(argv[4]) : 1;
  std::FILE *f = std::fopen(argv[1], "rb");
  if (!f) return std::perror(argv[1]), 2;
  int32_t head[2];
  nsk_ns_params p;
  if (!read_all(f, head, sizeof head) || head[0] != 0x4e534b31 || head[1] != (int32_t)sizeof p || !read_all(f, &p, sizeof p)) {
    std::fprintf(stderr, "%s: not a problem file of this build (struct of %zu bytes expected)\n", argv[1], sizeof p);
    return 2;
  }
  const size_t n = (size_t)p.nx * p.ny, ni = (size_t)(p.nx - 2) * (p.ny - 2);
  std::vector<double> Pr(n), dP(ni), dv(n);
  if (!read_all(f, Pr.data(), n * 8) || !read_all(f, dP.data(), ni * 8) || !read_all(f, dv.data(), n * 8)) {
    std::fprintf(stderr, "%s: short file\n", argv[1]);
    return 2;
  }
  std::fclose(f);

  nsk_ctx *ctx = nullptr;
  CHECK(nsk_create(0, &ctx));
  double *dPr = nullptr, *ddP = nullptr, *ddv = nullptr;
  CHECK(nsk_zeros(ctx, p.nx, p.ny, &dPr));          // @zeros(nx, ny)
  CHECK(nsk_zeros(ctx, p.nx - 2, p.ny - 2, &ddP));  // @zeros(nx-2, ny-2)
  CHECK(nsk_zeros(ctx, p.nx, p.ny, &ddv));
  double secs = 0.0;
  for (int r = 0; r < reps; r++) {  // every repetition starts from the file's fields: the result is that of `count` passes
    CHECK(nsk_upload(ctx, dPr, Pr.data(), n));
    CHECK(nsk_upload(ctx, ddP, dP.data(), ni));
    CHECK(nsk_upload(ctx, ddv, dv.data(), n));
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(nsk_ns_poisson_iters(ctx, dPr, ddP, ddv, &p, count));
    CHECK(nsk_sync(ctx));
    secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  double resid = 0.0;
  CHECK(nsk_ns_residual(ctx, nullptr, dPr, ddv, &p, 2, p.ny - 1, &resid));  // max|Rp| (NavierStokes.jl:89-90), Rp not kept
  int k_used = 0, rows_used = 0;
  CHECK(nsk_get_option(ctx, "poisson_k", &k_used));
  CHECK(nsk_get_option(ctx, "poisson_rows_used", &rows_used));
  CHECK(nsk_download(ctx, Pr.data(), dPr, n));
  CHECK(nsk_download(ctx, dP.data(), ddP, ni));
  CHECK(nsk_free(ctx, dPr));
  CHECK(nsk_free(ctx, ddP));
  CHECK(nsk_free(ctx, ddv));
  CHECK(nsk_destroy(ctx));

  f = std::fopen(argv[2], "wb");
  if (!f) return std::perror(argv[2]), 2;
  std::fwrite(Pr.data(), 8, n, f);
  std::fwrite(dP.data(), 8, ni, f);
  std::fwrite(&resid, 8, 1, f);
  std::fclose(f);
  std::printf("{\"nx\": %d, \"ny\": %d, \"passes\": %d, \"us_per_pass\": %.3f, \"passes_per_launch\": %d, \"rows_per_task\": %d, "
              "\"max_abs_residual\": %.17g}\n",
              p.nx, p.ny, count, secs / count * 1e6, k_used, rows_used, resid);
  return 0;
}
