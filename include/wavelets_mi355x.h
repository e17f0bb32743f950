/*
 * wavelets_mi355x.h -- C ABI of libwavelets_mi355x.so
 *
 * MI355X-native (CDNA4 / gfx950, hand-written HIP) backend for the one data-parallel
 * hot path of JuliaDSP/Wavelets.jl v0.10.1: the periodic orthogonal filter-bank
 * DWT/IDWT and the lifting DWT/IDWT behind dwt / idwt / dwt! / idwt! / wpt! / iwpt!.
 *
 * The reference has NO native interface (it is 100 % Julia): the seam this ABI
 * plugs into is the internal `_dwt!` / `_wpt!` method table that the public API
 * ends in (src/Transforms/transforms_main.jl:105-130, 134-176) -- exactly the seam
 * the reference's own KernelAbstractions extension uses
 * (ext/WaveletsGPUExt/filter_transforms_gpu.jl:171-214, lifting_transforms_gpu.jl:171).
 * A Julia maintainer adds `_dwt!(y::ROCArray, x::ROCArray, filter::OrthoFilter, L, fw)`
 * methods that `ccall` the entry points below (INTEGRATION.md shows the stub).
 *
 * Conventions
 *  - All array pointers are DEVICE pointers (HBM) owned by the caller.  Arrays are in
 *    Julia layout: column-major, dims[0] fastest, dense (no padding) unless an `ld`
 *    argument says otherwise.
 *  - dtype: WL_F32 (Float32) or WL_F64 (Float64).  Taps / lifting coefficients are
 *    passed as Float64 exactly as the reference stores them (OrthoFilter.qmf,
 *    LSStep.param.coef) and are converted to the element type before use, as
 *    WT.makereverseqmfpair / makescheme do (wt_main.jl:172-183,
 *    transforms_lifting.jl:13-25).  They are copied at call time; no host pointer
 *    is retained.
 *  - Every call is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *    default stream) and returns without synchronising.  A wl_ctx owns a grow-only
 *    device workspace and is NOT thread-safe: use one context per stream.
 *  - Return value: 0 (WL_OK) or a negative wl_status.  Nothing throws or aborts.
 *    The Julia glue maps the codes to the exceptions the reference throws
 *    (transforms_filter.jl:25-34): WL_EDIMS -> DimensionMismatch, WL_EINVAL_* /
 *    WL_EALIAS -> ArgumentError.
 *  - Arithmetic: sums are evaluated in the reference's order with separate multiply
 *    and add roundings (no FMA contraction), so Float32/Float64 results are
 *    bit-identical to the reference CPU loops on the same taps.
 *  - There is no CPU fallback: without a gfx950 device every entry point that needs
 *    one returns WL_EHIP / WL_ENODEVICE.
 */
#ifndef WAVELETS_MI355X_H
#define WAVELETS_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WL_VERSION 100 /* 0.1.0 */

#if defined(WL_BUILDING_LIB)
#define WL_API __attribute__((visibility("default")))
#else
#define WL_API
#endif

enum wl_dtype { WL_F32 = 0, WL_F64 = 1 };

typedef enum wl_status {
    WL_OK = 0,
    WL_EINVAL_SIZE = -1,   /* "size must have a sufficient power of 2 factor" (transforms_filter.jl:29-30) */
    WL_EINVAL_L = -2,      /* "L must be positive" (transforms_filter.jl:27-28)                          */
    WL_EALIAS = -3,        /* "in array is out array" (transforms_filter.jl:31-32)                       */
    WL_EDIMS = -4,         /* ndims not in 1..3 / non-positive extent / bad ld (DimensionMismatch)        */
    WL_EINVAL_CUBE = -5,   /* "array must be square/cube" (transforms_lifting.jl:131-132)                */
    WL_EINVAL_TREE = -6,   /* "invalid tree" (transforms_filter.jl:315-316)                              */
    WL_EINVAL_SCHEME = -7, /* lifting step list malformed (nsteps, ncoef out of range)                   */
    WL_EINVAL_DTYPE = -8,
    WL_EINVAL_FILTER = -9, /* flen < 2 or flen > WL_MAX_FLEN                                             */
    WL_EINVAL_ARG = -10,   /* NULL pointer etc.                                                          */
    WL_ENOMEM = -11,       /* device workspace allocation failed                                         */
    WL_EHIP = -12,         /* a HIP runtime call failed (wl_last_hip_error gives the hipError_t)         */
    WL_ENODEVICE = -13     /* no HIP device / device is not gfx950                                       */
} wl_status;

#define WL_MAX_FLEN 64     /* longest OrthoFilter supported (batt6 has 59 taps)   */
#define WL_MAX_STEPS 16    /* lifting steps per scheme                            */
#define WL_MAX_NCOEF 3     /* coefficients per lifting step (reference supports 1..3,
                              transforms_lifting.jl:455-483)                       */

typedef struct wl_ctx wl_ctx;

/* ---- context -------------------------------------------------------------------- */
/* Create a context on HIP device `device` (>= 0).  Fails with WL_ENODEVICE when there is
 * no such device or it is not gfx950.  (replaces: nothing -- Julia allocates scratch
 * vectors per call, transforms_filter.jl:16-23,117-119)                                */
WL_API int wl_ctx_create(int device, wl_ctx **out);
WL_API int wl_ctx_destroy(wl_ctx *ctx);
/* Upper bound of the device workspace a transform of this shape needs; wl_ctx_reserve
 * grows the context's workspace once so that later calls never allocate.              */
WL_API size_t wl_workspace_bytes(int dtype, int ndims, const int64_t *dims, int L);
WL_API int wl_ctx_reserve(wl_ctx *ctx, size_t bytes);
/* hipStreamSynchronize for hosts without their own HIP binding.                        */
WL_API int wl_stream_sync(wl_ctx *ctx, void *stream);
WL_API const char *wl_strerror(int status);
WL_API int wl_last_hip_error(const wl_ctx *ctx);
WL_API int wl_version(void);

/* ---- helpers mirroring src/Util -------------------------------------------------- */
/* maxtransformlevels(n) (non_dyadic.jl:14-23)                                          */
WL_API int wl_maxtransformlevels(int64_t n);

/* ---- filter-bank DWT / IDWT ------------------------------------------------------- */
/* y = dwt(x, OrthoFilter(qmf), L) (fw != 0) or idwt (fw == 0); 1-D, 2-D or 3-D.
 * replaces _dwt!(y, x, filter::OrthoFilter, L, fw) -- transforms_filter.jl:13-62 (1-D),
 * :113-188 (2-D), :192-294 (3-D).  y and x must not alias; sizes need a 2^L factor in
 * every dimension; L == 0 copies.  Output layout [s_L ; d_L ; ... ; d_1] (1-D) /
 * Mallat quadrants (2-D/3-D) exactly as the reference.                                 */
WL_API int wl_dwt_filter(wl_ctx *ctx, int dtype, void *y, const void *x,
                  int ndims, const int64_t *dims,
                  const double *qmf, int flen, int L, int fw, void *stream);

/* ---- lifting DWT / IDWT ------------------------------------------------------------ */
/* In place on y: dwt!(y, scheme::GLS, L) / idwt!(y, scheme, L).
 * replaces _dwt!(y, scheme::GLS, L, fw) -- transforms_lifting.jl:30-76 (1-D), :128-194
 * (2-D, square only), :200-278 (3-D, cube only).  The scheme is passed flattened in table
 * order (wt_main.jl:451-480): step i has type step_is_update[i] (0 = WT.Predict: updates
 * the first/approximation half, 1 = WT.Update: updates the second/detail half), ncoef[i]
 * coefficients and shift step_shift[i]; coefs_flat holds the coefficients back to back.
 * Sign/order/reciprocal adjustments for the direction are done inside (makescheme).     */
WL_API int wl_dwt_lifting(wl_ctx *ctx, int dtype, void *y,
                   int ndims, const int64_t *dims,
                   int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                   const int32_t *step_shift, const double *coefs_flat,
                   double norm1, double norm2, int L, int fw, void *stream);
/* Out-of-place variant: y = dwt(x, scheme, L) without the reference's copyto!(y, x)
 * (transforms_main.jl:119-124); same result as copy + in place.  y == x is allowed.    */
WL_API int wl_dwt_lifting_oop(wl_ctx *ctx, int dtype, void *y, const void *x,
                       int ndims, const int64_t *dims,
                       int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                       const int32_t *step_shift, const double *coefs_flat,
                       double norm1, double norm2, int L, int fw, void *stream);

/* ---- batched column-wise DWT ("dwtc") ------------------------------------------------ */
/* 1-D transform of every column of a len x nsignals column-major matrix with leading
 * dimension ld >= len.  The reference names dwtc/idwtc (transforms_main.jl:179-181,188)
 * but never implements them; this build defines them as the 1-D _dwt! per column.       */
WL_API int wl_dwtc_filter(wl_ctx *ctx, int dtype, void *y, const void *x,
                   int64_t len, int64_t nsignals, int64_t ld,
                   const double *qmf, int flen, int L, int fw, void *stream);
WL_API int wl_dwtc_lifting(wl_ctx *ctx, int dtype, void *y,
                    int64_t len, int64_t nsignals, int64_t ld,
                    int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                    const int32_t *step_shift, const double *coefs_flat,
                    double norm1, double norm2, int L, int fw, void *stream);

/* ---- wavelet packet transform (1-D) --------------------------------------------------- */
/* y = wpt(x, filter, tree) / iwpt.  tree: one byte per node of the BitVector
 * (length 2^maxtransformlevels(n) - 1, util_main.jl:301-344), HOST pointer.
 * replaces _wpt!(y, x, filter, tree, fw) -- transforms_filter.jl:301-359.               */
WL_API int wl_wpt_filter(wl_ctx *ctx, int dtype, void *y, const void *x, int64_t n,
                  const double *qmf, int flen,
                  const uint8_t *tree, int64_t ntree, int fw, void *stream);
/* In place: wpt!(y, scheme, tree).  replaces _wpt!(y, scheme::GLS, tree, fw) --
 * transforms_lifting.jl:283-319.                                                        */
WL_API int wl_wpt_lifting(wl_ctx *ctx, int dtype, void *y, int64_t n,
                   int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                   const int32_t *step_shift, const double *coefs_flat,
                   double norm1, double norm2,
                   const uint8_t *tree, int64_t ntree, int fw, void *stream);

/* ---- introspection (tests / bench) ---------------------------------------------------- */
/* Select the kernel family: 0 = auto (fast paths where they apply), 1 = generic kernels
 * only.  Both produce bit-identical results; the switch exists so tests can prove it.   */
WL_API int wl_ctx_set_path(wl_ctx *ctx, int path);
/* Name of the dominant kernel used by the last transform call on this context.          */
WL_API const char *wl_last_kernel(const wl_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* WAVELETS_MI355X_H */
