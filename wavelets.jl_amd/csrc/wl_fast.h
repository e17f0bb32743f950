// wl_fast.h -- forward filter-bank level loop with fast-path dispatch (wl_fwd.hip).
// Every fast kernel must produce results bit-identical to the generic kernels.
#pragma once
#include "wl_internal.h"

namespace wl {

// Enqueue all L forward levels of a filter-bank transform.  path: 0 = pick the best kernel
// per level (streaming kernels for large levels, one LDS-resident kernel for the tail,
// generic otherwise), 1 = generic kernels only.  *kernel_name = dominant kernel used.
template <typename T>
int filter_fwd_levels(void *ws, int cu_count, int path, hipStream_t st, const BoxSpec &b,
                      T *y, const T *x, const Taps<T> &taps, int L,
                      const char **kernel_name, int *hip_err);

}  // namespace wl
