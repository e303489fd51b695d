// One explicit instantiation of the multi-pass Poisson kernel per passes-per-launch value;
// built as ns_poisson_k<K>.o with -DNSK_K=<K> so the eight variants compile in parallel.
#include "ns_poisson_kernel.h"

namespace nsk_poisson {
template int launch_k<NSK_K>(nsk_ctx *, hipStream_t, PoissonArgs &, bool, int, const TaskPlan &);
}  // namespace nsk_poisson

#ifdef NSK_POISSON_TRACE
#define NSK_CAT_(a, b) a##b
#define NSK_CAT(a, b) NSK_CAT_(a, b)
extern "C" int NSK_CAT(nsk_debug_trace_k, NSK_K)(unsigned long long *out, int ntasks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsk_poisson::nsk_trace_buf),
                                  sizeof(unsigned long long) * 4 * (size_t)ntasks);
}
// counters of the FMA-division safety net: copies the four counters out and clears them
extern "C" int NSK_CAT(nsk_debug_counters_k, NSK_K)(unsigned long long *out) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nsk_poisson::nsk_trace_cnt), sizeof(unsigned long long) * 4);
  unsigned long long zero[4] = {0, 0, 0, 0};
  if (!rc) rc = (int)hipMemcpyToSymbol(HIP_SYMBOL(nsk_poisson::nsk_trace_cnt), zero, sizeof zero);
  return rc;
}
#endif
