// Microbenchmark: does the last-level cache (256 MB, memory side) turn a reversed traversal into hits?
// A launch reads x (two fields) and g (one field) and writes y (two fields); the next launch reads what
// this one wrote (x and y swap), as the Poisson launches do.  Blocks are dispatched in index order, so
// the address a launch touches grows with time; "zig-zag" reverses the block order in every other
// launch, so that a launch starts with the data the previous one touched last.  Field sizes from
// below the cache to above it; loads of x plain or non-temporal.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ void __launch_bounds__(256) sweep(const v2d *__restrict__ x, const v2d *__restrict__ g, v2d *__restrict__ y,
                                             size_t nfield, int reverse) {
  const size_t nb = gridDim.x, b = reverse ? nb - 1 - blockIdx.x : blockIdx.x;
  const size_t i = b * 256 + threadIdx.x;   // index into one field, in 16-byte units
  if (i >= nfield) return;
  v2d a0, a1;
  if (NT) {
    a0 = __builtin_nontemporal_load(x + i);
    a1 = __builtin_nontemporal_load(x + nfield + i);
  } else {
    a0 = x[i];
    a1 = x[nfield + i];
  }
  const v2d c = g[i];
  y[i] = a0 * 0.5 + c;
  y[nfield + i] = a1 * 0.5 - c;
}
int main() {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (double mb : {21.0, 42.0, 63.0, 84.0, 105.0, 168.0}) {    // bytes of ONE field; a launch moves five
    const size_t nfield = (size_t)(mb * 1e6 / 16);
    v2d *x, *y, *g;
    if (hipMalloc(&x, nfield * 32) != hipSuccess || hipMalloc(&y, nfield * 32) != hipSuccess ||
        hipMalloc(&g, nfield * 16) != hipSuccess) return 1;
    (void)hipMemset(x, 0, nfield * 32);
    (void)hipMemset(y, 0, nfield * 32);
    (void)hipMemset(g, 0, nfield * 16);
    const unsigned blocks = (unsigned)((nfield + 255) / 256);
    for (int nt = 0; nt < 2; nt++)
      for (int zig = 0; zig < 2; zig++) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          const int launches = 40;
          (void)hipEventRecord(e0);
          for (int l = 0; l < launches; l++) {
            const int rev = zig ? (l & 1) : 0;
            if (nt) sweep<true><<<blocks, 256>>>(x, g, y, nfield, rev);
            else sweep<false><<<blocks, 256>>>(x, g, y, nfield, rev);
            v2d *t = x; x = y; y = t;
          }
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms = 0;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms / launches < best) best = ms / launches;
        }
        printf("field %5.0f MB (five fields %4.0f MB)  %s loads  %s: %7.1f us per launch, %6.0f GB/s\n", mb, 5 * mb,
               nt ? "nt   " : "plain", zig ? "zig-zag" : "forward", best * 1e3, 5 * mb * 1e6 / (best * 1e-3) / 1e9);
      }
    (void)hipFree(x);
    (void)hipFree(y);
    (void)hipFree(g);
  }
  return 0;
}
