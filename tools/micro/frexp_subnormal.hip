// What v_frexp_exp_i32_f64 returns for the values the Poisson kernel's safety tracker relies on:
// 0 for zero, Inf and NaN; the true exponent for subnormals.  Prints one line per case and exits
// nonzero on a surprise.   hipcc --offload-arch=gfx950 -O2 -o frexp_subnormal frexp_subnormal.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>

__global__ void k(const double *x, int *e, int n) {
  int i = threadIdx.x;
  if (i < n) e[i] = __builtin_amdgcn_frexp_exp(x[i]);
}

int main() {
  const double in[] = {0.0, -0.0, 1.0, 0.75, 0x1p-848, 0x1p-849, 0x1p-1022, 0x1p-1023, 0x1.8p-1040, 0x1p-1074,
                       -0x1p-1074, 1e-318, INFINITY, -INFINITY, NAN, 0x1p1023};
  const int n = sizeof in / sizeof in[0];
  double *dx; int *de; int out[n];
  if (hipMalloc(&dx, sizeof in) != hipSuccess || hipMalloc(&de, sizeof out) != hipSuccess) return 2;
  hipMemcpy(dx, in, sizeof in, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, de, n);
  if (hipMemcpy(out, de, sizeof out, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad = 0;
  for (int i = 0; i < n; i++) {
    int want = 0;
    if (std::isfinite(in[i]) && in[i] != 0.0) std::frexp(in[i], &want);
    printf("%-24a device %6d  host frexp %6d %s\n", in[i], out[i], want, out[i] == want ? "" : "  <-- differs");
    bad += out[i] != want;
  }
  return bad ? 1 : 0;
}
