// Microbenchmark: cost of one __syncthreads() round of a single workgroup on gfx950 as a function of
// its number of wavefronts (the bubble solvers' SOR runs one barrier per anti-diagonal step).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) loop(double *out, int iters) {
  __shared__ double s[1024];
  double x = threadIdx.x;
  for (int it = 0; it < iters; it++) {
    s[threadIdx.x] = x;
    __syncthreads();
    x = s[(threadIdx.x + 1) & (blockDim.x - 1)] + 1.0;  // a neighbour's value: the barrier cannot be dropped
  }
  out[threadIdx.x] = x;
}
int main() {
  double *out;
  (void)hipMalloc(&out, sizeof(double) * 1024);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int iters = 20000;
  for (int threads : {64, 128, 256, 512, 1024}) {
    loop<<<1, threads>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    loop<<<1, threads>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%2d wavefronts: %.0f ns per step (LDS write, barrier, LDS read, add)\n", threads / 64, ms * 1e6 / iters);
  }
  return 0;
}
