// Microbenchmark: the Poisson kernel's memory access pattern without its arithmetic.  2560 x 4095
// doubles per field, five fields (reads x0, x1, g; writes y0, y1; x and y swap between launches); one
// wavefront per task = a strip of 128 columns (1 KiB per row) x a chunk of rows, all tasks resident at
// once, every wave marching through its rows with the loads of D rows in flight.  Compared with the
// block-sequential sweep of zigzag_llc.hip this shows what the many concurrent row streams cost.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int LD = 2560, NY = 4095, STRIPS = LD / 128;
template <bool NT, int D>
__global__ void __launch_bounds__(64) tasks(const double *__restrict__ x0, const double *__restrict__ x1,
                                            const double *__restrict__ g, double *__restrict__ y0,
                                            double *__restrict__ y1, int rows_per_chunk) {
  const int strip = blockIdx.x % STRIPS, chunk = blockIdx.x / STRIPS;
  const int r0 = chunk * rows_per_chunk, r1 = min(r0 + rows_per_chunk, NY);
  const size_t col = (size_t)strip * 128 + 2 * threadIdx.x;
  v2d a[D], b[D], c[D];
  auto ld = [&](int r, int s) {
    const size_t o = (size_t)r * LD + col;
    if (NT) {
      a[s] = __builtin_nontemporal_load((const v2d *)(x0 + o));
      b[s] = __builtin_nontemporal_load((const v2d *)(x1 + o));
    } else {
      a[s] = *(const v2d *)(x0 + o);
      b[s] = *(const v2d *)(x1 + o);
    }
    c[s] = *(const v2d *)(g + o);
  };
#pragma unroll
  for (int s = 0; s < D; s++)
    if (r0 + s < r1) ld(r0 + s, s);
  for (int r = r0; r < r1; r += D) {
#pragma unroll
    for (int s = 0; s < D; s++) {
      if (r + s < r1) {
        const v2d p = a[s] * 0.5 + c[s], q = b[s] * 0.5 - c[s];
        if (r + s + D < r1) ld(r + s + D, s);
        const size_t o = (size_t)(r + s) * LD + col;
        *(v2d *)(y0 + o) = p;
        *(v2d *)(y1 + o) = q;
      }
    }
  }
}
int main() {
  const size_t n = (size_t)LD * NY;
  double *f[5];
  for (auto &p : f) {
    if (hipMalloc(&p, n * 8) != hipSuccess) return 1;
    (void)hipMemset(p, 0, n * 8);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rows : {24, 40, 48, 96}) {
    const int chunks = (NY + rows - 1) / rows;
    const unsigned grid = (unsigned)(chunks * STRIPS);
    for (int nt = 0; nt < 2; nt++)
      for (int d : {2, 4}) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
          const int launches = 40;
          (void)hipEventRecord(e0);
          for (int l = 0; l < launches; l++) {
            double *x0 = f[(l & 1) ? 3 : 0], *x1 = f[(l & 1) ? 4 : 1], *y0 = f[(l & 1) ? 0 : 3], *y1 = f[(l & 1) ? 1 : 4];
            if (nt && d == 2) tasks<true, 2><<<grid, 64>>>(x0, x1, f[2], y0, y1, rows);
            else if (nt) tasks<true, 4><<<grid, 64>>>(x0, x1, f[2], y0, y1, rows);
            else if (d == 2) tasks<false, 2><<<grid, 64>>>(x0, x1, f[2], y0, y1, rows);
            else tasks<false, 4><<<grid, 64>>>(x0, x1, f[2], y0, y1, rows);
          }
          (void)hipEventRecord(e1);
          (void)hipEventSynchronize(e1);
          float ms = 0;
          (void)hipEventElapsedTime(&ms, e0, e1);
          if (ms / launches < best) best = ms / launches;
        }
        printf("rows per task %3d (%5u waves)  %s loads, %d rows in flight: %6.1f us per launch, %5.0f GB/s\n", rows, grid,
               nt ? "nt   " : "plain", d, best * 1e3, 5.0 * n * 8 / (best * 1e-3) / 1e9);
      }
  }
  return 0;
}
