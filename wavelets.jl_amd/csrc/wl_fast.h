// wl_fast.h -- fast-path dispatcher (large power-of-two levels, short even filters).
// Every fast kernel must produce results bit-identical to the generic kernels.
#pragma once
#include "wl_internal.h"

namespace wl {

// Forward filter-bank transform.  Sets *handled = 1 when the whole transform (all L levels)
// was enqueued by fast kernels; otherwise leaves *handled = 0 and enqueues nothing.
template <typename T>
int fast_filter_fwd(void *ws, int cu_count, hipStream_t st, int nd, int nt, const int64_t dims[3], Strides3 full,
                    T *y, const T *x, const Taps<T> &taps, int L,
                    int *handled, const char **kernel_name, int *hip_err);

}  // namespace wl
