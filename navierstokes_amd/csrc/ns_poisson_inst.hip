// One explicit instantiation of the multi-pass Poisson kernel per passes-per-launch value;
// built as ns_poisson_k<K>.o with -DNSK_K=<K> so the eight variants compile in parallel.
#include "ns_poisson_kernel.h"

namespace nsk_poisson {
template int launch_k<NSK_K>(nsk_ctx *, PoissonArgs &, bool, int, int, int);
}  // namespace nsk_poisson
