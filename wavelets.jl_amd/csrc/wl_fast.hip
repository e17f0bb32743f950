// wl_fast.hip -- fast paths (placeholder dispatcher; kernels land in wl_fwd2d.hip etc.)
#include "wl_fast.h"

namespace wl {

template <typename T>
int fast_filter_fwd(void *, int, hipStream_t, int, int, const int64_t[3], Strides3, T *, const T *,
                    const Taps<T> &, int, int *handled, const char **, int *)
{
    *handled = 0;
    return WL_OK;
}
template int fast_filter_fwd<float>(void *, int, hipStream_t, int, int, const int64_t[3], Strides3, float *, const float *,
                                    const Taps<float> &, int, int *, const char **, int *);
template int fast_filter_fwd<double>(void *, int, hipStream_t, int, int, const int64_t[3], Strides3, double *, const double *,
                                     const Taps<double> &, int, int *, const char **, int *);
}  // namespace wl
