// wl_internal.h -- shared declarations of libwavelets_mi355x (not part of the ABI).
//
// Arithmetic contract (all kernels): products and sums are rounded separately
// (no FMA contraction: the library is built with -ffp-contract=off and this header
// sets `#pragma clang fp contract(off)`; the GPU parity tests compare bits) and sums run in the order of
// the reference's shift-register loops:
//   forward  s[k] = ((h0*x[2k] + h1*x[2k+1]) + h2*x[2k+2]) + ...            (m ascending)
//            d[k] = ((g[F-1]*x[2k+2-F] + g[F-2]*x[2k+3-F]) + ...) + g0*x[2k+1] (m descending)
//   inverse  S[o] = sum over m descending, (o-m) even, of h[m]*s[(o-m)/2]
//            D[o] = sum over m ascending, (o+m-1) even, of g[m]*d[(o+m-1)/2]
//            x[o] = S[o] + D[o]
// (transforms_filter.jl:362-369 @filtermainloop, :387-433 filtdown!, :467-541 filtup!;
//  g[m] = (-1)^m h[m] = Util.mirror, util_main.jl:30), all indices periodic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/wavelets_mi355x.h"

#ifdef WL_FMA
// the opt-in fused build (libwavelets_mi355x_fma.so, `make FMA=1`): a*b+c may contract to one rounding; results agree with the
// reference to the tolerance SURVEY.md 8(c) states, not bit for bit.  Same sources, same summation order.
#pragma clang fp contract(fast)
#else
#pragma clang fp contract(off)
#endif

namespace wl {

// ---- per-context options (wl_ctx_set_option) ------------------------------------------------
// Tuning / test switches live on the context, never in the process environment.  Every ABI entry point installs
// its context's table for the duration of the call (CallScope, wl_ctx.h); kernels' launchers read it with opt().
// None of them changes a result: they select between kernel families that are bit-identical by construction.
struct Opts {
    enum { kMax = 32, kKeyLen = 32 };
    int n = 0;
    char key[kMax][kKeyLen];
    long long val[kMax];
};
extern thread_local const Opts *tl_opts;
// The calling context's hand-over block (wl_ctx::sync): kSyncWords zero-initialised 32-bit words in device memory for launches whose
// workgroups hand data to one another (the fused pair + tile launch, wl_pair2d.hip).  Every such launch leaves the block all zero.
enum { kSyncWords = 16384 };
extern thread_local unsigned *tl_sync;
inline long long opt(const char *name, long long dflt)
{
    const Opts *o = tl_opts;
    if (o)
        for (int i = 0; i < o->n; ++i)
            if (strcmp(o->key[i], name) == 0) return o->val[i];
    return dflt;
}

template <typename T>
struct Taps {
    int F;
    T h[WL_MAX_FLEN];   // qmf converted to T (makereverseqmfpair: copyto!(Vector{T}, qmf))
    T g[WL_MAX_FLEN];   // mirror(h)
};

// makereverseqmfpair (wt_main.jl:172-183): taps converted to T first, mirror sign applied in T
template <typename T>
inline void make_taps(const double *qmf, int flen, Taps<T> &t)
{
    t.F = flen;
    for (int i = 0; i < WL_MAX_FLEN; ++i) { t.h[i] = (T)0; t.g[i] = (T)0; }
    for (int i = 0; i < flen; ++i) {
        t.h[i] = (T)qmf[i];                                   // copyto!(Vector{T}, qmf)
        t.g[i] = (i % 2 == 0) ? t.h[i] : (T)(t.h[i] * (T)-1); // mirror(h), util_main.jl:30
    }
}

template <typename T>
struct LiftStep {
    int is_update;      // 0: Predict (writes the s half), 1: Update (writes the d half)
    int nc;
    int shift;
    T c[WL_MAX_NCOEF];  // direction-adjusted (makescheme)
};
template <typename T>
struct LiftScheme {
    int nsteps;
    LiftStep<T> step[WL_MAX_STEPS];
    T norm1, norm2;     // direction-adjusted
};

// A strided 3-D view in elements.
struct Strides3 { int64_t s[3]; };
struct Extent3 { int64_t n[3]; };

// ---- generic (any size, any filter length, any axis) kernels: wl_generic.hip ----
template <typename T>
hipError_t generic_fwd_filter_pass(hipStream_t st, const Taps<T> &taps,
                                   const T *src, Strides3 sst,
                                   T *dst, Strides3 dst_st,
                                   T *ll, Strides3 ll_st,
                                   Extent3 n, int axis, Extent3 lo, const uint8_t *mask = nullptr);
template <typename T>
hipError_t generic_inv_filter_pass(hipStream_t st, const Taps<T> &taps,
                                   const T *src, Strides3 sst,
                                   const T *ll, Strides3 ll_st,
                                   T *dst, Strides3 dst_st,
                                   Extent3 n, int axis, Extent3 lo, const uint8_t *mask = nullptr);
// lifting building blocks (box n, [s;d] layout along `axis` in the dense work buffer w)
template <typename T>
hipError_t generic_lift_split(hipStream_t st, const T *src, Strides3 sst, T *w, Strides3 wst,
                              Extent3 n, int axis, const uint8_t *mask = nullptr);
template <typename T>
hipError_t generic_lift_step(hipStream_t st, const LiftStep<T> &step, T *w, Strides3 wst,
                             Extent3 n, int axis, const uint8_t *mask = nullptr);
template <typename T>
hipError_t generic_lift_finish_fwd(hipStream_t st, T n1, T n2, const T *w, Strides3 wst,
                                   T *dst, Strides3 dst_st, T *ll, Strides3 ll_st,
                                   Extent3 n, int axis, Extent3 lo, const uint8_t *mask = nullptr);
template <typename T>
hipError_t generic_lift_norm_inv(hipStream_t st, T n1, T n2, const T *src, Strides3 sst,
                                 const T *ll, Strides3 ll_st, T *w, Strides3 wst,
                                 Extent3 n, int axis, Extent3 lo, const uint8_t *mask = nullptr);
template <typename T>
hipError_t generic_lift_merge(hipStream_t st, const T *w, Strides3 wst, T *dst, Strides3 dst_st,
                              Extent3 n, int axis, const uint8_t *mask = nullptr);
template <typename T>
hipError_t generic_copy_box(hipStream_t st, const T *src, Strides3 sst, T *dst, Strides3 dst_st, Extent3 n);

// ---- workspace carve-up shared by the level loops (elements of T; N = box elements, nt = transformed axes) ----
//   A, B   : (N >> nt) + 64 each   approximation ping-pong (level l writes the one level l+1 reads; the largest tenant is
//                                  the level-1 approximation, or the level-2 reconstruction of an inverse)
//   T0, T1 : N each                inter-pass buffers of the generic / long-filter / 3-D families (T1 only for 3-D)
//   W      : N                     lifting work buffer
// The fast filter-bank paths only ever touch A and B: their calls reserve ws_ab_elems(); a level that needs T0 / T1 / W
// without having them returns WL_RETRY_GEN (internal) and the ABI layer repeats the call with the full workspace
// (out-of-place filter transforms only, so repeating is harmless).
constexpr int WL_RETRY_GEN = -1000;
// (internal) level 1 of a batch was asked to read virtually shifted planes (SrcView) but no tier that understands the view
// is eligible: returned BEFORE anything is enqueued -- the source holds fewer planes than the batch box declares
constexpr int WL_RETRY_NOVIEW = -1001;
inline size_t ws_ab_each(int64_t N, int nt) { return (size_t)((N >> nt) + 64); }
inline size_t ws_ab_elems(int64_t N, int nt) { return 2 * ws_ab_each(N, nt); }
inline size_t ws_elems(int64_t N, int nt = 1) { return ws_ab_elems(N, nt) + (size_t)(3 * N + 64); }
template <typename T>
struct Work { T *T0, *T1, *W, *A, *B; };
template <typename T>
inline Work<T> carve(void *ws, int64_t N, int nt = 1, bool with_gen = true)
{
    Work<T> w;
    T *p = (T *)ws;
    w.A = p; p += ws_ab_each(N, nt);
    w.B = p; p += ws_ab_each(N, nt);
    if (with_gen) { w.T0 = p; p += N; w.T1 = p; p += N; w.W = p; }
    else { w.T0 = nullptr; w.T1 = nullptr; w.W = nullptr; }
    return w;
}
inline Strides3 dense_strides(const int64_t n[3])
{
    Strides3 s;
    s.s[0] = 1; s.s[1] = n[0]; s.s[2] = n[0] * n[1];
    return s;
}
// A "box transform" covers dwt (all nd axes transformed) and dwtc (axis 0 of a len x nsignals box).
struct BoxSpec {
    int nd;                 // rank of the array (1..3)
    int nt;                 // transformed axes are 0..nt-1
    int64_t dims[3];        // full extents (unused dims = 1)
    Strides3 full;          // strides of x / y
};
inline void level_box(const BoxSpec &b, int l /*1-based*/, int64_t n[3])
{
    for (int d = 0; d < 3; ++d) n[d] = (d < b.nt) ? (b.dims[d] >> (l - 1)) : b.dims[d];
}
inline Extent3 low_corner(const BoxSpec &b, const int64_t n[3])
{
    Extent3 lo;
    for (int d = 0; d < 3; ++d) lo.n[d] = (d < b.nt) ? (n[d] >> 1) : n[d];
    return lo;
}

__device__ __forceinline__ int64_t pmod(int64_t a, int64_t n)
{
    int64_t r = a % n;
    return r < 0 ? r + n : r;
}

}  // namespace wl
