/*
 * oracle/wl_oracle_ext.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU oracle for the SURVEY.md section 8(f) rows 3 and 4 ("next" rows around the hot path):
 *   - modwt / imodwt            src/Transforms/transforms_maximal_overlap.jl:10-107
 *   - threshold! (all THTypes)  src/Threshold/threshold_main.jl:21-117
 *   - noisest / mad!            src/Threshold/denoising.jl:92-110
 *   - circshift!, arrayadd!     src/Util/util_main.jl:105-130, src/Threshold/denoising.jl:82-88
 * Literal restatements of the reference's Julia loops, with Julia's promotion rules written out
 * (a Float32 array combined with Float64 taps / a Float64 threshold is computed in Float64 and
 * rounded to Float32 on every store, exactly as `w1[t] += h[n] * v[k]` does).
 *
 * PARITY UNPINNED: the reference's tests hold no golden vectors for these functions
 * (test/transforms.jl:325-344 checks only imodwt(modwt(x)) ~ x and sizes; test/threshold.jl:1-21
 * only calls the functions).  The oracle is pinned by those properties only (tests/test_oracle_ext.py).
 * Third-party arithmetic: `median!` comes from the Statistics stdlib (Project.toml compat
 * Statistics = "1"), absent from /root/reference; its published algorithm is restated here:
 * NaN anywhere -> NaN; odd n -> the middle order statistic; even n -> middle(a, b) = a/2 + b/2 of the
 * two middle order statistics.  `sortperm(x, alg=QuickSort, by=abs)` (BiggestTH) is not stable in
 * Julia; ties at the cut are resolved here by index order (lower index is "smaller").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define WLO_API __attribute__((visibility("default")))
#define WLO_MAXF 128

enum { WLO_OK = 0, WLO_EINVAL_SIZE = -1, WLO_EINVAL_L = -2, WLO_EINVAL_DIMS = -4, WLO_EINVAL_DTYPE = -8, WLO_EINVAL_FILTER = -9,
       WLO_EINVAL_TH = -10 };

static long mod1l(long a, long n) { long r = (a - 1) % n; if (r < 0) r += n; return r + 1; }

/* WT.makereverseqmfpair(wt) with fw = true, T = Float64 (wt_main.jl:172-183), then g /= sqrt(2); h /= sqrt(2)
 * (transforms_maximal_overlap.jl:50-52): g = reverse(qmf)/sqrt(2) (scaling), h = mirror(qmf)/sqrt(2) (detail). */
static void modwt_filters(const double *qmf, int F, double *g, double *h)
{
    const double r2 = sqrt(2.0);
    for (int i = 0; i < F; ++i) {
        g[i] = qmf[F - 1 - i] / r2;
        h[i] = ((i & 1) ? -qmf[i] : qmf[i]) / r2;      /* mirror: (-1)^(i-1) in 1-based = (-1)^i 0-based */
    }
}

#define DEF_MODWT(T, SUF)                                                                                       \
    /* modwt_step, transforms_maximal_overlap.jl:10-31 (1-based t, k; v1/w1 are Vector{T}) */                   \
    static void modwt_step_##SUF(const T *v, long N, int j, const double *h, const double *g, int F, T *v1, T *w1) \
    {                                                                                                           \
        const long stride = 1L << (j - 1);                                                                      \
        for (long t = 1; t <= N; ++t) {                                                                         \
            long k = t;                                                                                         \
            w1[t - 1] = (T)(h[0] * (double)v[k - 1]);                                                           \
            v1[t - 1] = (T)(g[0] * (double)v[k - 1]);                                                           \
            for (int n = 2; n <= F; ++n) {                                                                      \
                k -= stride;                                                                                    \
                if (k <= 0) k = mod1l(k, N);                                                                    \
                w1[t - 1] = (T)((double)w1[t - 1] + h[n - 1] * (double)v[k - 1]);                               \
                v1[t - 1] = (T)((double)v1[t - 1] + g[n - 1] * (double)v[k - 1]);                               \
            }                                                                                                   \
        }                                                                                                       \
    }                                                                                                           \
    /* modwt, :47-63: returns [W V], N x (L+1), column j = level-j detail, last column = scaling */             \
    static void modwt_##SUF(T *out, const T *x, long N, const double *qmf, int F, int L)                        \
    {                                                                                                           \
        double g[WLO_MAXF], h[WLO_MAXF];                                                                        \
        modwt_filters(qmf, F, g, h);                                                                            \
        T *V = (T *)malloc((size_t)N * sizeof(T)), *v1 = (T *)malloc((size_t)N * sizeof(T));                    \
        memcpy(V, x, (size_t)N * sizeof(T));                                                                    \
        for (int j = 1; j <= L; ++j) {                                                                          \
            modwt_step_##SUF(V, N, j, h, g, F, v1, out + (size_t)(j - 1) * N);                                  \
            memcpy(V, v1, (size_t)N * sizeof(T));                                                               \
        }                                                                                                       \
        memcpy(out + (size_t)L * N, V, (size_t)N * sizeof(T));                                                  \
        free(V); free(v1);                                                                                      \
    }                                                                                                           \
    /* imodwt_step, :72-93 */                                                                                   \
    static void imodwt_step_##SUF(const T *v, const T *w, long N, int j, const double *h, const double *g, int F, T *v0) \
    {                                                                                                           \
        const long stride = 1L << (j - 1);                                                                      \
        for (long t = 1; t <= N; ++t) {                                                                         \
            long k = t;                                                                                         \
            v0[t - 1] = (T)(h[0] * (double)w[k - 1] + g[0] * (double)v[k - 1]);                                 \
            for (int n = 2; n <= F; ++n) {                                                                      \
                k += stride;                                                                                    \
                if (k > N) k = mod1l(k, N);                                                                     \
                v0[t - 1] = (T)((double)v0[t - 1] + (h[n - 1] * (double)w[k - 1] + g[n - 1] * (double)v[k - 1])); \
            }                                                                                                   \
        }                                                                                                       \
    }                                                                                                           \
    /* imodwt, :99-107: xw is N x ncols */                                                                      \
    static void imodwt_##SUF(T *x, const T *xw, long N, int ncols, const double *qmf, int F)                    \
    {                                                                                                           \
        double g[WLO_MAXF], h[WLO_MAXF];                                                                        \
        modwt_filters(qmf, F, g, h);                                                                            \
        T *tmp = (T *)malloc((size_t)N * sizeof(T));                                                            \
        memcpy(x, xw + (size_t)(ncols - 1) * N, (size_t)N * sizeof(T));                                         \
        for (int j = ncols - 1; j >= 1; --j) {                                                                  \
            imodwt_step_##SUF(x, xw + (size_t)(j - 1) * N, N, j, h, g, F, tmp);                                 \
            memcpy(x, tmp, (size_t)N * sizeof(T));                                                              \
        }                                                                                                       \
        free(tmp);                                                                                              \
    }

DEF_MODWT(float, f32)
DEF_MODWT(double, f64)

/* maxmodwttransformlevels(n) = floor(Int, log2(n)), non_dyadic.jl:24-25 */
WLO_API int wlo_maxmodwttransformlevels(int64_t n)
{
    int l = 0;
    while (n > 1) { n >>= 1; ++l; }
    return l;
}
WLO_API int wlo_modwt(int dtype, void *out, const void *x, int64_t N, const double *qmf, int flen, int L)
{
    if (flen < 1 || flen > WLO_MAXF) return WLO_EINVAL_FILTER;
    if (N < 1) return WLO_EINVAL_DIMS;
    if (L > wlo_maxmodwttransformlevels(N)) return WLO_EINVAL_SIZE;     /* "Too many transform levels (length(x) < 2^L)" */
    if (L < 1) return WLO_EINVAL_L;                                      /* "L must be >= 1" */
    if (dtype == 0) modwt_f32((float *)out, (const float *)x, (long)N, qmf, flen, L);
    else if (dtype == 1) modwt_f64((double *)out, (const double *)x, (long)N, qmf, flen, L);
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}
WLO_API int wlo_imodwt(int dtype, void *x, const void *xw, int64_t N, int ncols, const double *qmf, int flen)
{
    if (flen < 1 || flen > WLO_MAXF) return WLO_EINVAL_FILTER;
    if (N < 1 || ncols < 1) return WLO_EINVAL_DIMS;
    if (dtype == 0) imodwt_f32((float *)x, (const float *)xw, (long)N, ncols, qmf, flen);
    else if (dtype == 1) imodwt_f64((double *)x, (const double *)xw, (long)N, ncols, qmf, flen);
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* threshold!, threshold_main.jl:21-117.  th: 0 Hard, 1 Soft, 2 SemiSoft, 3 Stein, 4 Pos, 5 Neg.
 * C = the type Julia's promotion gives `x[i] op t`: the element type when t is an Integer or has the
 * element type (t_is_f64 = 0), Float64 when t is a Float64 (t_is_f64 = 1). */
#define DEF_TH(T, C, SUF)                                                                                       \
    static void threshold_##SUF(T *x, long n, int th, double t_)                                                \
    {                                                                                                           \
        const C t = (C)t_;                                                                                      \
        for (long i = 0; i < n; ++i) {                                                                          \
            const C xi = (C)x[i];                                                                               \
            const C ax = xi < 0 ? -xi : xi;                                                                     \
            const C sg = (C)((xi > 0) - (xi < 0));                                                              \
            switch (th) {                                                                                       \
            case 0: if (ax <= t) x[i] = 0; break;                                     /* :37-47 */              \
            case 1: { C sh = ax - t; if (sh < 0) x[i] = 0; else x[i] = (T)(sg * sh); } break;   /* :50-63 */    \
            case 2:                                                                   /* :66-82 */              \
                if (xi <= 2 * t) {                                                                              \
                    C sh = ax - t;                                                                              \
                    if (sh < 0) x[i] = 0;                                                                       \
                    else if (sh - t < 0) x[i] = (T)(sg * sh * 2);                                               \
                }                                                                                               \
                break;                                                                                          \
            case 3: { C sh = 1 - t * t / (xi * xi); if (sh < 0) x[i] = 0; else x[i] = (T)(xi * sh); } break;    /* :85-98 */ \
            case 4: if (xi > 0) x[i] = 0; break;                                      /* PosTH :113-122 */      \
            case 5: if (xi < 0) x[i] = 0; break;                                      /* NegTH :101-110 */      \
            }                                                                                                   \
        }                                                                                                       \
    }
DEF_TH(float, float, f32n)
DEF_TH(float, double, f32w)
DEF_TH(double, double, f64)

WLO_API int wlo_threshold(int dtype, void *x, int64_t n, int th, double t, int t_is_f64)
{
    if (th < 0 || th > 5) return WLO_EINVAL_TH;
    if (th <= 3 && !(t >= 0)) return WLO_EINVAL_TH;          /* @assert t >= 0 */
    if (dtype == 0) { if (t_is_f64) threshold_f32w((float *)x, (long)n, th, t); else threshold_f32n((float *)x, (long)n, th, t); }
    else if (dtype == 1) threshold_f64((double *)x, (long)n, th, t);
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}

/* BiggestTH, threshold_main.jl:22-34: zero the n-m entries of smallest magnitude (ties: lower index first) */
typedef struct { double a; long i; } absidx;
static int cmp_absidx(const void *p, const void *q)
{
    const absidx *x = (const absidx *)p, *y = (const absidx *)q;
    if (x->a < y->a) return -1;
    if (x->a > y->a) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
WLO_API int wlo_threshold_biggest(int dtype, void *x, int64_t n, int64_t m)
{
    if (m < 0) return WLO_EINVAL_TH;
    if (dtype != 0 && dtype != 1) return WLO_EINVAL_DTYPE;
    if (m > n) m = n;
    absidx *v = (absidx *)malloc((size_t)(n > 0 ? n : 1) * sizeof(absidx));
    for (long i = 0; i < n; ++i) { v[i].a = fabs(dtype == 0 ? (double)((float *)x)[i] : ((double *)x)[i]); v[i].i = i; }
    qsort(v, (size_t)n, sizeof(absidx), cmp_absidx);
    for (long i = 0; i < n - m; ++i) { if (dtype == 0) ((float *)x)[v[i].i] = 0; else ((double *)x)[v[i].i] = 0; }
    free(v);
    return WLO_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Statistics.median! restated (see header) and mad!, denoising.jl:103-110 */
static int cmp_f32(const void *p, const void *q) { float a = *(const float *)p, b = *(const float *)q; return (a > b) - (a < b); }
static int cmp_f64(const void *p, const void *q) { double a = *(const double *)p, b = *(const double *)q; return (a > b) - (a < b); }
#define DEF_MEDIAN(T, SUF, CMP)                                                                                 \
    static T median_##SUF(T *v, long n)                                                                         \
    {                                                                                                           \
        for (long i = 0; i < n; ++i) if (v[i] != v[i]) return (T)NAN;                                           \
        qsort(v, (size_t)n, sizeof(T), CMP);                                                                    \
        if (n & 1) return v[n / 2];                                                                             \
        return v[n / 2 - 1] / 2 + v[n / 2] / 2;                                                                 \
    }                                                                                                           \
    static T mad_##SUF(T *y, long n)                                                                            \
    {                                                                                                           \
        const T m = median_##SUF(y, n);                                                                         \
        for (long i = 0; i < n; ++i) { T d = y[i] - m; y[i] = d < 0 ? -d : d; }                                 \
        return median_##SUF(y, n);                                                                              \
    }
DEF_MEDIAN(float, f32, cmp_f32)
DEF_MEDIAN(double, f64, cmp_f64)

/* the arrays are permuted (sorted), as partialsort! is allowed to do */
WLO_API int wlo_median(int dtype, void *v, int64_t n, double *out)
{
    if (n < 1) return WLO_EINVAL_DIMS;
    if (dtype == 0) *out = (double)median_f32((float *)v, (long)n);
    else if (dtype == 1) *out = median_f64((double *)v, (long)n);
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}
WLO_API int wlo_mad(int dtype, void *y, int64_t n, double *out)
{
    if (n < 1) return WLO_EINVAL_DIMS;
    if (dtype == 0) *out = (double)mad_f32((float *)y, (long)n);
    else if (dtype == 1) *out = mad_f64((double *)y, (long)n);
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* b[i] = a[i - shift] (mod dims) along every dimension: Util.circshift! (util_main.jl:105-130) for
 * vectors, Base.circshift for arrays (denoising.jl:56,62); column-major dims[0] fastest. */
WLO_API int wlo_circshift(int dtype, void *b, const void *a, int ndims, const int64_t *dims, const int64_t *shift)
{
    if (ndims < 1 || ndims > 3) return WLO_EINVAL_DIMS;
    long d[3] = {1, 1, 1}, s[3] = {0, 0, 0};
    for (int k = 0; k < ndims; ++k) { d[k] = (long)dims[k]; s[k] = (long)(((shift[k] % dims[k]) + dims[k]) % dims[k]); }
    const size_t es = dtype == 0 ? 4 : 8;
    if (dtype != 0 && dtype != 1) return WLO_EINVAL_DTYPE;
    for (long i2 = 0; i2 < d[2]; ++i2)
        for (long i1 = 0; i1 < d[1]; ++i1)
            for (long i0 = 0; i0 < d[0]; ++i0) {
                const long j0 = (i0 - s[0] + d[0]) % d[0], j1 = (i1 - s[1] + d[1]) % d[1], j2 = (i2 - s[2] + d[2]) % d[2];
                memcpy((char *)b + es * (size_t)(i0 + d[0] * (i1 + d[1] * i2)), (const char *)a + es * (size_t)(j0 + d[0] * (j1 + d[1] * j2)), es);
            }
    return WLO_OK;
}
/* arrayadd!(y, z): y[i] += z[i], denoising.jl:82-88;  rmul!(y, s) with s::Float64: y[i] = T(y[i] * s) */
WLO_API int wlo_arrayadd(int dtype, void *y, const void *z, int64_t n)
{
    if (dtype == 0) for (long i = 0; i < n; ++i) ((float *)y)[i] = ((float *)y)[i] + ((const float *)z)[i];
    else if (dtype == 1) for (long i = 0; i < n; ++i) ((double *)y)[i] = ((double *)y)[i] + ((const double *)z)[i];
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}
WLO_API int wlo_rmul(int dtype, void *y, int64_t n, double s)
{
    if (dtype == 0) for (long i = 0; i < n; ++i) ((float *)y)[i] = (float)((double)((float *)y)[i] * s);
    else if (dtype == 1) for (long i = 0; i < n; ++i) ((double *)y)[i] = ((double *)y)[i] * s;
    else return WLO_EINVAL_DTYPE;
    return WLO_OK;
}
