/*
 * wavelets_mi355x.h -- C ABI of libwavelets_mi355x.so
 *
 * MI355X-native (CDNA4 / gfx950, hand-written HIP) backend for the one data-parallel
 * hot path of JuliaDSP/Wavelets.jl v0.10.1: the periodic orthogonal filter-bank
 * DWT/IDWT and the lifting DWT/IDWT behind dwt / idwt / dwt! / idwt! / wpt! / iwpt!,
 * plus the callers on either side of that path: modwt / imodwt,
 * threshold!, the median / mad! noise estimate and the array helpers of denoise.
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
 *    device workspace and is NOT thread-safe: use one context per stream.  Calls run on
 *    the context's device and leave the caller's current HIP device unchanged.
 *  - Return value: 0 (WL_OK) or a negative wl_status.  Nothing throws or aborts.
 *    The Julia glue maps the codes to the exceptions the reference throws
 *    (transforms_filter.jl:25-34): WL_EDIMS -> DimensionMismatch, WL_EINVAL_* /
 *    WL_EALIAS -> ArgumentError.
 *  - Arithmetic: sums are evaluated in the reference's order with separate multiply
 *    and add roundings (no FMA contraction), so Float32/Float64 results are
 *    bit-identical to the reference CPU loops on the same taps.  A second library
 *    with the same ABI, libwavelets_mi355x_fma.so (the same sources built with FMA
 *    contraction allowed, `make FMA=1`), is the opt-in "fused" arithmetic mode: it
 *    agrees with the reference to ||y - ref||_2 / ||ref||_2 <= 1e-6 sqrt(L) (Float32),
 *    1e-13 sqrt(L) (Float64) instead of bit for bit.  A host selects it by loading
 *    that file; neither library ever falls back on the other.
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

/* ---- sharding a batch of independent units over GPUs (dwtc columns, images) --------------------------------- */
/* Contiguous block partition: rank r of `world` owns units [*lo, *hi).  The path has no exchange step: one process
 * (or Julia task) per GPU calls wl_dwtc_* on its own column block with its own context; only the wavelet description travels
 * between ranks.  Host-side arithmetic only: usable without a device.  (replaces: nothing -- the reference has no dwtc,
 * transforms_main.jl:179-181; north_star defines the multi-GPU batch)                                               */
WL_API int wl_shard_range(int64_t nunits, int rank, int world, int64_t *lo, int64_t *hi);

/* ---- context -------------------------------------------------------------------- */
/* Create a context on HIP device `device` (>= 0).  Fails with WL_ENODEVICE when there is
 * no such device or it is not gfx950.  (replaces: nothing -- Julia allocates scratch
 * vectors per call, transforms_filter.jl:16-23,117-119)                                */
WL_API int wl_ctx_create(int device, wl_ctx **out);
WL_API int wl_ctx_destroy(wl_ctx *ctx);
/* Device workspace of the fast filter-bank paths for dwt / idwt of this shape: the
 * approximation ping-pong, 2 * (N / 2^ndims) elements (N = number of samples; for dwtc pass
 * ndims = 1 and dims[0] = len * nsignals).  wl_ctx_reserve grows the context's workspace
 * once so that later calls of that kind never allocate.  Lifting transforms, long (> 10 taps
 * in 2-D) / odd-length filters, 3-D boxes and the generic kernel family use up to 4 N more
 * elements: the context grows to that on their first call.  Growth inside a transform call is STREAM-ORDERED (hipFreeAsync /
 * hipMallocAsync on the call's stream: the old block is released behind the work already queued, nothing waits on the host and
 * the device is not synchronised); wl_ctx_reserve itself, which takes no stream, synchronises the device when it has to grow.
 * wl_ctx_workspace_held reports the current size.                                                                          */
WL_API size_t wl_workspace_bytes(int dtype, int ndims, const int64_t *dims, int L);
/* Upper bound for every TRANSFORM entry point on this shape (wl_dwt_*, wl_dwtc_*, wl_wpt_*: lifting, long / odd filters,
 * 3-D, the generic kernel family): reserve this much and no later transform of that shape allocates or synchronises,
 * whatever path it takes.  NOT covered: wl_denoise_ti_filter, whose batch of shifted copies needs about
 * (6.5 * prod(nspin) + nspin[0]) N elements (capped, see WL_TI_WS_CAP_MB) -- a plain one-spin denoise about 5.5 N -- and grows
 * the workspace (synchronising) on its first call like any other path; wl_modwt / wl_imodwt allocate nothing here.      */
WL_API size_t wl_workspace_bytes_full(int dtype, int ndims, const int64_t *dims, int L);
WL_API int wl_ctx_reserve(wl_ctx *ctx, size_t bytes);
WL_API size_t wl_ctx_workspace_held(const wl_ctx *ctx);
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
/* Out-of-place variant (x -> y, same shapes and ld): saves the copy a caller would otherwise make before the in-place call and
 * the staging copy the in-place first level needs (y == x is allowed and is the in-place call).                              */
WL_API int wl_dwtc_lifting_oop(wl_ctx *ctx, int dtype, void *y, const void *x,
                        int64_t len, int64_t nsignals, int64_t ld,
                        int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                        const int32_t *step_shift, const double *coefs_flat,
                        double norm1, double norm2, int L, int fw, void *stream);

/* ---- a batch of independent 2-D transforms ----------------------------------------------- */
/* y[:, :, i] = dwt(x[:, :, i], filter, L) (fw = 0: idwt) for i = 0 .. nimages-1: nimages images of dims[0] x dims[1]
 * (column-major, dense), image i at element offset i * image_stride of x and of y (image_stride >= dims[0] * dims[1]).
 * The same results as nimages calls of wl_dwt_filter with ndims = 2 -- the reference has no batched form
 * (transforms_filter.jl:113-188 is the per-image loop) -- but every level is ONE launch over all images: mid-size images,
 * whose single transform is launch latency rather than bandwidth, fill the chip together.  Workspace:
 * wl_workspace_bytes(dtype, 1, {nimages * image_stride}, L) is an upper bound for the fast paths.                          */
WL_API int wl_dwt_filter_batch(wl_ctx *ctx, int dtype, void *y, const void *x, const int64_t *dims, int64_t nimages,
                        int64_t image_stride, const double *qmf, int flen, int L, int fw, void *stream);

/* ---- wavelet packet transform (1-D) --------------------------------------------------- */
/* y = wpt(x, filter, tree) / iwpt.  tree: one byte per node of the BitVector
 * (length 2^maxtransformlevels(n) - 1, util_main.jl:301-344), HOST pointer; it is copied before the call returns (the node
 * bits of partially split depths travel through a pinned staging buffer of the context) and the stream is NOT synchronised.
 * A call with a partially split tree is not capturable in a hipGraph (its staging copy would be replayed with stale bits).
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

/* wpt(x, wt, L::Integer) / iwpt(x, wt, L) / wpt!(y, x, filter, L) / wpt!(y, scheme, L): the FULL tree of depth L
 * (= maketree(length(x), L, :full), transforms_main.jl:134-176) without a tree vector.  A tree of an n-sample signal has
 * n - 1 nodes; building, validating and scanning it on the host costs more than the transform itself from a few 10^5
 * samples on, and these entry points never look at one.  0 <= L <= maxtransformlevels(n), else WL_EINVAL_L.
 * Same results as the tree forms with the full tree; no stream synchronisation; capturable in a hipGraph.              */
WL_API int wl_wpt_filter_full(wl_ctx *ctx, int dtype, void *y, const void *x, int64_t n,
                       const double *qmf, int flen, int L, int fw, void *stream);
WL_API int wl_wpt_lifting_full(wl_ctx *ctx, int dtype, void *y, int64_t n,
                        int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                        const int32_t *step_shift, const double *coefs_flat,
                        double norm1, double norm2, int L, int fw, void *stream);

/* ---- maximal-overlap DWT, 1-D -------------------------- */
/* floor(log2(n)) -- replaces maxmodwttransformlevels, src/Util/non_dyadic.jl:24-25.        */
WL_API int wl_maxmodwttransformlevels(int64_t n);
/* W = modwt(x, filter, L): `out` is the n x (L+1) matrix (column-major, leading dimension
 * ldo >= n): column j-1 = level-j detail coefficients, column L = level-L scaling
 * coefficients.  Taps are used in Float64 (divided by sqrt 2) and every partial sum is
 * rounded to the element type, exactly as the reference's `w1[t] += h[n] * v[k]` does.
 * Errors: L > floor(log2 n) -> WL_EINVAL_SIZE ("Too many transform levels"), L < 1 ->
 * WL_EINVAL_L.  replaces modwt + modwt_step, src/Transforms/transforms_maximal_overlap.jl:10-63. */
WL_API int wl_modwt(wl_ctx *ctx, int dtype, void *out, int64_t ldo, const void *x, int64_t n,
             const double *qmf, int flen, int L, void *stream);
/* x = imodwt(xw, filter): xw is n x ncols (leading dimension ldw), x must not alias it.
 * replaces imodwt + imodwt_step, transforms_maximal_overlap.jl:72-107.                     */
WL_API int wl_imodwt(wl_ctx *ctx, int dtype, void *x, const void *xw, int64_t ldw, int64_t n, int ncols,
              const double *qmf, int flen, void *stream);

/* ---- thresholding and noise estimate -------------------- */
/* THType of src/Threshold/threshold_main.jl:8-15 (BiggestTH has its own entry point).      */
enum wl_thtype { WL_TH_HARD = 0, WL_TH_SOFT = 1, WL_TH_SEMISOFT = 2, WL_TH_STEIN = 3, WL_TH_POS = 4, WL_TH_NEG = 5 };
/* threshold!(x, TH, t) in place on n elements.  t_is_f64 selects the arithmetic type Julia's
 * promotion gives `x[i] op t`: 0 = the element type (t is an Integer or has the element
 * type), 1 = Float64 (t::Float64, e.g. sigma*dnt.t in denoise), result rounded to the
 * element type on store.  t is ignored for WL_TH_POS / WL_TH_NEG; t < 0 -> WL_EINVAL_ARG
 * (the reference's @assert).  replaces threshold!, threshold_main.jl:37-122.               */
WL_API int wl_threshold(wl_ctx *ctx, int dtype, void *x, int64_t n, int th, double t, int t_is_f64, void *stream);
/* threshold!(x, BiggestTH(), m): keep the m entries of largest magnitude, zero the rest
 * (exact order-statistic selection on device; ties at the cut are cleared in index order --
 * the reference's QuickSort leaves that order unspecified).  Synchronises `stream` once.
 * replaces threshold_main.jl:22-34.                                                        */
WL_API int wl_threshold_biggest(wl_ctx *ctx, int dtype, void *x, int64_t n, int64_t m, void *stream);
/* *result = median(v) (Statistics.median!: middle order statistic, or a/2 + b/2 of the two
 * middle ones; NaN if any NaN), computed in the element type and widened to double.  `v` is
 * not modified.  Synchronises `stream` (the result is a host scalar, as in the reference).  */
WL_API int wl_median(wl_ctx *ctx, int dtype, const void *v, int64_t n, double *result, void *stream);
/* *result = mad!(y): m = median!(y); y[i] = abs(y[i] - m); median!(y).  y is overwritten by
 * the absolute deviations.  replaces mad!, src/Threshold/denoising.jl:103-110 (noisest :92-101
 * = dwt level 1 + this on the level-1 detail range, divided by 0.6745 on the host).        */
WL_API int wl_mad(wl_ctx *ctx, int dtype, void *y, int64_t n, double *result, void *stream);
/* b[i] = a[i - shift] (periodic, every dimension; dims/shift have ndims entries, b != a).
 * replaces Util.circshift! (src/Util/util_main.jl:105-130) and Base.circshift as used by the
 * translation-invariant branch of denoise (denoising.jl:44-64).                            */
WL_API int wl_circshift(wl_ctx *ctx, int dtype, void *b, const void *a, int ndims, const int64_t *dims,
                 const int64_t *shift, void *stream);
/* y[i] += z[i] -- arrayadd!, denoising.jl:82-88.                                           */
WL_API int wl_arrayadd(wl_ctx *ctx, int dtype, void *y, const void *z, int64_t n, void *stream);
/* y[i] = T(y[i] * s) with s::Float64 -- rmul!(y, 1/pns), denoising.jl:66.                  */
WL_API int wl_rmul(wl_ctx *ctx, int dtype, void *y, int64_t n, double s, void *stream);

/* denoise(x, wt::OrthoFilter; L, dnt, TI = true, nspin) fused on the device (denoising.jl:21-67), 1-D vectors and square
 * matrices: all prod(nspin) circularly shifted copies are transformed, thresholded and transformed back as ONE batch
 * (in groups when the buffers would exceed the context's cap), then un-shifted and summed in spin order -- the summation
 * order of the reference, so the result carries the same roundings -- and scaled by 1/prod(nspin).  The noise estimate
 * sigma = noisest(x, wt) = mad!(level-1 detail range) / 0.6745 is computed on the device and consumed there (no host
 * round trip) unless sigma_host >= 0 supplies it (a custom estnoise).  th: wl_thtype, t_unit: dnt.t (threshold =
 * sigma * t_unit in Float64).  y must not alias x.  Nothing is allocated per spin; the call only enqueues.            */
WL_API int wl_denoise_ti_filter(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims,
                         const double *qmf, int flen, int L, int th, double t_unit, const int64_t *nspin,
                         double sigma_host, void *stream);
/* The same for a lifting scheme (wt::GLS; scheme arguments as wl_dwt_lifting): shifted signals run as one batched-lines
 * transform, shifted images one 2-D lifting transform per plane; sigma and everything else stay on the device.
 * replaces the translation-invariant branch of denoise(x, wt::GLS; TI=true), denoising.jl:36-67 (round 4).               */
WL_API int wl_denoise_ti_lifting(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims,
                          int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                          const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                          int L, int th, double t_unit, const int64_t *nspin, double sigma_host, void *stream);

/* ---- introspection (tests / bench) ---------------------------------------------------- */
/* Select the kernel family: 0 = auto (fast paths where they apply), 1 = generic kernels
 * only.  Both produce bit-identical results; the switch exists so tests can prove it.   */
WL_API int wl_ctx_set_path(wl_ctx *ctx, int path);
/* Name of the dominant kernel used by the last transform call on this context.          */
WL_API const char *wl_last_kernel(const wl_ctx *ctx);
/* Tuning / test switches of this context (the library reads no environment variable).
 * key: e.g. "WL_FUSE2_MIN", "WL_TJ", "WL_NO_INV2D" (DESIGN.md section 5 lists them with their
 * defaults); keys shorter than 32 characters, at most 32 per context; unknown keys are stored
 * and ignored.  No option changes a result: they pick between kernel families that are
 * bit-identical by construction, which is what the tests use them to prove.                */
WL_API int wl_ctx_set_option(wl_ctx *ctx, const char *key, int64_t value);
WL_API int wl_ctx_clear_options(wl_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* WAVELETS_MI355X_H */
