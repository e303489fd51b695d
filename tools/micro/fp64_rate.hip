// Microbenchmark: issue rate of fp64 VALU instructions on gfx950 for 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int CHAINS, int KIND>
__global__ void __launch_bounds__(64) k(double* out, int iters, double a, double b) {
  double x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = a + i + threadIdx.x;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
#pragma unroll
      for (int i = 0; i < CHAINS; i++) {
        if (KIND == 0) x[i] = __builtin_fma(x[i], a, b);
        if (KIND == 1) x[i] = x[i] + b;
        if (KIND == 2) x[i] = x[i] * a;
        if (KIND == 3) {  // dpp wave shift of a double (2 x v_mov_b32_dpp)
          int lo = __double2loint(x[i]), hi = __double2hiint(x[i]);
          lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false);
          hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false);
          x[i] = __hiloint2double(hi, lo) + b;
        }
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int CHAINS, int KIND>
void run(const char* name, int blocks) {
  double* out;
  hipMalloc(&out, sizeof(double) * 64 * 8192);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<CHAINS, KIND><<<blocks, 64>>>(out, 10, 1.0000001, 1e-9);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS, KIND><<<blocks, 64>>>(out, iters, 1.0000001, 1e-9);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double instr = (double)iters * 8 * CHAINS;           // per wave
  double waves_per_simd = blocks / 1024.0;
  double ns_per_instr_simd = ms * 1e6 / (instr * waves_per_simd);
  printf("%-6s chains=%2d waves/SIMD=%.0f: %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n",
         name, CHAINS, waves_per_simd, ms, ns_per_instr_simd, ns_per_instr_simd * 2.4);
  hipFree(out);
}
int main() {
  for (int blocks : {1024, 2048, 4096}) {
    run<1, 0>("fma", blocks); run<2, 0>("fma", blocks); run<4, 0>("fma", blocks); run<8, 0>("fma", blocks); run<16, 0>("fma", blocks);
    run<8, 1>("add", blocks); run<8, 2>("mul", blocks); run<8, 3>("dpp+add", blocks);
  }
  return 0;
}
