// Microbenchmark: how fast does a lone wavefront run?  One chain of dependent fp64 FMAs (latency
// bound: the time per FMA is the pipeline latency in cycles over the clock) executed by a single
// wavefront on an otherwise idle GPU, then by one wavefront per SIMD of the whole chip, back to back
// for long enough that power management has settled.  If the lone wavefront's time per FMA is much
// longer, small single-workgroup kernels (the bubble solvers' SOR) run at a reduced clock.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) chain(double *out, int iters, double a, double b) {
  double x = a + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 16; r++) x = __builtin_fma(x, a, b);
  }
  out[blockIdx.x * 64 + threadIdx.x] = x;
}
int main() {
  double *out;
  hipMalloc(&out, sizeof(double) * 64 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;  // 320 000 dependent FMAs per launch
  const int cases[4] = {1, 1024, 1, 1024};
  for (int c = 0; c < 4; c++) {
    const int blocks = cases[c];
    for (int rep = 0; rep < 40; rep++) chain<<<blocks, 64>>>(out, iters, 1.0000001, 1e-9);  // settle
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; rep++) chain<<<blocks, 64>>>(out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%5d wavefront(s): %.2f ns per dependent fp64 FMA\n", blocks, ms * 1e6 / (20.0 * iters * 16));
  }
  return 0;
}
