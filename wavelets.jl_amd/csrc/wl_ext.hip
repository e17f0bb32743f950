// wl_ext.hip -- the callers around the hot path:
//   modwt / imodwt                       transforms_maximal_overlap.jl:10-107
//   threshold! (all THTypes)             threshold_main.jl:21-117
//   median! / mad! (noise estimate)      denoising.jl:92-110 (+ Statistics.median!)
//   circshift / arrayadd! / rmul!        util_main.jl:105-130, denoising.jl:82-88 (translation-invariant denoising)
// All of it is HBM-bound element-wise or gather work; the arithmetic follows Julia's promotion rules
// literally (Float32 data with Float64 taps / threshold: compute in Float64, round on every store) so the
// results are bit-identical to the reference loops.
#include "wl_ctx.h"
#include "wl_fast.h"
#include "wl_dev.h"

#include <cmath>
#include <cstring>

using namespace wl;

namespace {

constexpr int EXT_THREADS = 256;
inline unsigned ext_blocks(int64_t n, int per_thread, int cu_count)
{
    int64_t b = (n + (int64_t)EXT_THREADS * per_thread - 1) / ((int64_t)EXT_THREADS * per_thread);
    const int64_t cap = (int64_t)cu_count * 16;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (unsigned)b;
}


// element-wise in-place pass: 16-byte vectors when the pointer is aligned, scalar tail
template <typename T, typename F>
__device__ __forceinline__ void ew_inplace(T *__restrict__ x, int64_t n, int vec_ok, F f)
{
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    constexpr int U = 4;                      // independent 16-byte accesses in flight per lane
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nv = vec_ok ? n / V : 0;
    VT *xv = reinterpret_cast<VT *>(x);
    const int64_t tile = (int64_t)U * blockDim.x, ntiles = nv / tile;
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int64_t base = tl * tile + threadIdx.x;
        VT v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xv[base + (int64_t)u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int e = 0; e < V; ++e) v[u][e] = f(v[u][e], (base + (int64_t)u * blockDim.x) * V + e);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[base + (int64_t)u * blockDim.x] = v[u];
    }
    for (int64_t i = ntiles * tile + gid; i < nv; i += nthr) {
        VT v = xv[i];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = f(v[e], i * V + e);
        xv[i] = v;
    }
    for (int64_t i = nv * V + gid; i < n; i += nthr) x[i] = f(x[i], i);
}
inline int vec_ok16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0 ? 1 : 0; }

// ---- MODWT -----------------------------------------------------------------------------------------
struct ModwtTaps { int F; double h[WL_MAX_FLEN]; double g[WL_MAX_FLEN]; };   // h: detail, g: scaling (both / sqrt 2)

// WT.makereverseqmfpair(wt) (fw, Float64: g = reverse(qmf), h = mirror(qmf)) then `/= sqrt(2)`, :50-52
void make_modwt_taps(const double *qmf, int F, ModwtTaps &t)
{
    const double r2 = std::sqrt(2.0);
    t.F = F;
    for (int i = 0; i < WL_MAX_FLEN; ++i) { t.h[i] = 0.0; t.g[i] = 0.0; }
    for (int i = 0; i < F; ++i) {
        t.g[i] = qmf[F - 1 - i] / r2;
        t.h[i] = ((i & 1) ? -qmf[i] : qmf[i]) / r2;
    }
}

// modwt_step (:10-31): w1[t] = sum_n h[n] v[t - n*stride], v1[t] = sum_n g[n] v[...], accumulated in tap order,
// every partial sum rounded to T (Julia: `w1[t] += h[n] * v[k]` with w1::Vector{T}, h::Vector{Float64}).
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_modwt_step(const T *__restrict__ v, T *__restrict__ v1, T *__restrict__ w1,
                                                            int64_t N, int64_t stride, ModwtTaps tp)
{
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += nthr) {
        int64_t k = t;
        double x = (double)v[k];
        T w = (T)(tp.h[0] * x), s = (T)(tp.g[0] * x);
        for (int n = 1; n < tp.F; ++n) {
            k -= stride;
            if (k < 0) { k += N; if (k < 0) { k %= N; if (k < 0) k += N; } }   // one add unless the stride exceeds N (tiny signals)
            x = (double)v[k];
            w = (T)((double)w + tp.h[n] * x);
            s = (T)((double)s + tp.g[n] * x);
        }
        w1[t] = w;
        v1[t] = s;
    }
}
// imodwt_step (:72-93): v0[t] = sum_n (h[n] w[t + n*stride] + g[n] v[t + n*stride])
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_imodwt_step(const T *__restrict__ v, const T *__restrict__ w, T *__restrict__ v0,
                                                             int64_t N, int64_t stride, ModwtTaps tp)
{
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N; t += nthr) {
        int64_t k = t;
        T acc = (T)(tp.h[0] * (double)w[k] + tp.g[0] * (double)v[k]);
        for (int n = 1; n < tp.F; ++n) {
            k += stride;
            if (k >= N) { k -= N; if (k >= N) k %= N; }
            acc = (T)((double)acc + (tp.h[n] * (double)w[k] + tp.g[n] * (double)v[k]));
        }
        v0[t] = acc;
    }
}

// the same two kernels for strides that are multiples of the 16-byte vector width (levels >= 3 for Float32,
// >= 2 for Float64) and N a multiple of it: every tap of V consecutive outputs is one aligned vector load
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_modwt_step_v(const T *__restrict__ v, T *__restrict__ v1, T *__restrict__ w1,
                                                              int64_t NV, int64_t strideV, ModwtTaps tp)
{
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    const VT *vv = reinterpret_cast<const VT *>(v);
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < NV; t += nthr) {
        int64_t k = t;
        VT x = vv[k], w, s;
#pragma unroll
        for (int e = 0; e < V; ++e) { const double xd = (double)x[e]; w[e] = (T)(tp.h[0] * xd); s[e] = (T)(tp.g[0] * xd); }
        for (int n = 1; n < tp.F; ++n) {
            k -= strideV;
            if (k < 0) { k += NV; if (k < 0) { k %= NV; if (k < 0) k += NV; } }
            x = vv[k];
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const double xd = (double)x[e];
                w[e] = (T)((double)w[e] + tp.h[n] * xd);
                s[e] = (T)((double)s[e] + tp.g[n] * xd);
            }
        }
        reinterpret_cast<VT *>(w1)[t] = w;
        reinterpret_cast<VT *>(v1)[t] = s;
    }
}
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_imodwt_step_v(const T *__restrict__ v, const T *__restrict__ w, T *__restrict__ v0,
                                                               int64_t NV, int64_t strideV, ModwtTaps tp)
{
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    const VT *vv = reinterpret_cast<const VT *>(v), *wv = reinterpret_cast<const VT *>(w);
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < NV; t += nthr) {
        int64_t k = t;
        VT a = vv[k], b = wv[k], acc;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = (T)(tp.h[0] * (double)b[e] + tp.g[0] * (double)a[e]);
        for (int n = 1; n < tp.F; ++n) {
            k += strideV;
            if (k >= NV) { k -= NV; if (k >= NV) k %= NV; }
            a = vv[k];
            b = wv[k];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = (T)((double)acc[e] + (tp.h[n] * (double)b[e] + tp.g[n] * (double)a[e]));
        }
        reinterpret_cast<VT *>(v0)[t] = acc;
    }
}

// strides below the vector width (levels 1-2 for Float32, level 1 for Float64; S divides V): a thread still produces V
// consecutive outputs from aligned 16-byte loads -- the V/S taps that fall into one vector step are served from the register
// pair [previous chunk | current chunk] with compile-time offsets, then the pair slides by one chunk.  Same tap order and
// per-tap rounding as k_modwt_step (the scalar kernel issued one 4-byte load per lane and tap: 1.2 TB/s of level traffic).
template <typename T, int S>
__global__ void __launch_bounds__(EXT_THREADS) k_modwt_step_s(const T *__restrict__ v, T *__restrict__ v1, T *__restrict__ w1,
                                                              int64_t NV, ModwtTaps tp)
{
    constexpr int V = 16 / sizeof(T), G = V / S;
    typedef T VT __attribute__((ext_vector_type(V)));
    const VT *vv = reinterpret_cast<const VT *>(v);
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < NV; t += nthr) {
        int64_t k = t;
        VT c = vv[k];
        int64_t kp = (k == 0) ? NV - 1 : k - 1;
        VT p = vv[kp];
        VT w, s;
        for (int n0 = 0; n0 < tp.F; n0 += G) {
            T win[2 * V];
#pragma unroll
            for (int e = 0; e < V; ++e) { win[e] = p[e]; win[V + e] = c[e]; }
            c = p;                                         // slide: the next group of taps starts one chunk back
            kp = (kp == 0) ? NV - 1 : kp - 1;
            p = vv[kp];
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const int n = n0 + r;
                if (n < tp.F) {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const double xd = (double)win[V + e - r * S];
                        if (n == 0) { w[e] = (T)(tp.h[0] * xd); s[e] = (T)(tp.g[0] * xd); }
                        else { w[e] = (T)((double)w[e] + tp.h[n] * xd); s[e] = (T)((double)s[e] + tp.g[n] * xd); }
                    }
                }
            }
        }
        reinterpret_cast<VT *>(w1)[t] = w;
        reinterpret_cast<VT *>(v1)[t] = s;
    }
}
template <typename T, int S>
__global__ void __launch_bounds__(EXT_THREADS) k_imodwt_step_s(const T *__restrict__ v, const T *__restrict__ w, T *__restrict__ v0,
                                                               int64_t NV, ModwtTaps tp)
{
    constexpr int V = 16 / sizeof(T), G = V / S;
    typedef T VT __attribute__((ext_vector_type(V)));
    const VT *vv = reinterpret_cast<const VT *>(v), *wv = reinterpret_cast<const VT *>(w);
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < NV; t += nthr) {
        int64_t kn = (t + 1 == NV) ? 0 : t + 1;
        VT ca = vv[t], cb = wv[t], na = vv[kn], nb = wv[kn], acc;
        for (int n0 = 0; n0 < tp.F; n0 += G) {
            T wa[2 * V], wb[2 * V];
#pragma unroll
            for (int e = 0; e < V; ++e) { wa[e] = ca[e]; wa[V + e] = na[e]; wb[e] = cb[e]; wb[V + e] = nb[e]; }
            ca = na; cb = nb;
            kn = (kn + 1 == NV) ? 0 : kn + 1;
            na = vv[kn]; nb = wv[kn];
#pragma unroll
            for (int r = 0; r < G; ++r) {
                const int n = n0 + r;
                if (n < tp.F) {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const double term = tp.h[n] * (double)wb[e + r * S] + tp.g[n] * (double)wa[e + r * S];
                        acc[e] = (n == 0) ? (T)term : (T)((double)acc[e] + term);
                    }
                }
            }
        }
        reinterpret_cast<VT *>(v0)[t] = acc;
    }
}

template <typename T>
int modwt_impl(wl_ctx *ctx, hipStream_t st, T *out, int64_t ldo, const T *x, int64_t N, const double *qmf, int flen, int L)
{
    constexpr int V = 16 / sizeof(T);
    int rc = wl_ensure_ws(ctx, (size_t)2 * N * sizeof(T), st, true);
    if (rc != WL_OK) return rc;
    ModwtTaps tp;
    make_modwt_taps(qmf, flen, tp);
    T *A = (T *)ctx->ws, *B = A + N;
    const T *cur = x;
    const bool vec_base = (N % V) == 0 && (ldo % V) == 0 && vec_ok16(x) && vec_ok16(out);
    for (int j = 1; j <= L; ++j) {
        T *vdst = (j == L) ? out + (int64_t)L * ldo : ((j & 1) ? A : B);
        const int64_t stride = (int64_t)1 << (j - 1);
        if (vec_base && (stride % V) == 0) {
            hipLaunchKernelGGL((k_modwt_step_v<T>), dim3(ext_blocks(N / V, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, cur, vdst,
                               out + (int64_t)(j - 1) * ldo, N / V, stride / V, tp);
        } else if (vec_base && stride == 1 && N >= 2 * V && opt("WL_MODWT_SMALL", 1) != 0) {
            hipLaunchKernelGGL((k_modwt_step_s<T, 1>), dim3(ext_blocks(N / V, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, cur, vdst,
                               out + (int64_t)(j - 1) * ldo, N / V, tp);
        } else if (vec_base && stride == 2 && V == 4 && N >= 2 * V && opt("WL_MODWT_SMALL", 1) != 0) {
            hipLaunchKernelGGL((k_modwt_step_s<T, (sizeof(T) == 4 ? 2 : 1)>), dim3(ext_blocks(N / V, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st,
                               cur, vdst, out + (int64_t)(j - 1) * ldo, N / V, tp);
        } else {
            hipLaunchKernelGGL((k_modwt_step<T>), dim3(ext_blocks(N, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, cur, vdst,
                               out + (int64_t)(j - 1) * ldo, N, stride, tp);
        }
        WL_HIP(ctx, hipGetLastError());
        cur = vdst;
    }
    ctx->last_kernel = "k_modwt_step";
    return WL_OK;
}
template <typename T>
int imodwt_impl(wl_ctx *ctx, hipStream_t st, T *x, const T *xw, int64_t ldw, int64_t N, int ncols, const double *qmf, int flen)
{
    constexpr int V = 16 / sizeof(T);
    int rc = wl_ensure_ws(ctx, (size_t)2 * N * sizeof(T), st, true);
    if (rc != WL_OK) return rc;
    ModwtTaps tp;
    make_modwt_taps(qmf, flen, tp);
    T *A = (T *)ctx->ws, *B = A + N;
    const T *cur = xw + (int64_t)(ncols - 1) * ldw;
    if (ncols == 1) { WL_HIP(ctx, hipMemcpyAsync(x, cur, (size_t)N * sizeof(T), hipMemcpyDeviceToDevice, st)); return WL_OK; }
    const bool vec_base = (N % V) == 0 && (ldw % V) == 0 && vec_ok16(x) && vec_ok16(xw);
    for (int j = ncols - 1; j >= 1; --j) {
        if (j - 1 >= 62) return WL_EINVAL_L;
        T *dst = (j == 1) ? x : ((j & 1) ? A : B);
        const int64_t stride = (int64_t)1 << (j - 1);
        if (vec_base && (stride % V) == 0) {
            hipLaunchKernelGGL((k_imodwt_step_v<T>), dim3(ext_blocks(N / V, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, cur,
                               xw + (int64_t)(j - 1) * ldw, dst, N / V, stride / V, tp);
        } else if (vec_base && stride == 1 && N >= 2 * V && opt("WL_MODWT_SMALL", 1) != 0) {
            hipLaunchKernelGGL((k_imodwt_step_s<T, 1>), dim3(ext_blocks(N / V, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, cur,
                               xw + (int64_t)(j - 1) * ldw, dst, N / V, tp);
        } else if (vec_base && stride == 2 && V == 4 && N >= 2 * V && opt("WL_MODWT_SMALL", 1) != 0) {
            hipLaunchKernelGGL((k_imodwt_step_s<T, (sizeof(T) == 4 ? 2 : 1)>), dim3(ext_blocks(N / V, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st,
                               cur, xw + (int64_t)(j - 1) * ldw, dst, N / V, tp);
        } else {
            hipLaunchKernelGGL((k_imodwt_step<T>), dim3(ext_blocks(N, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, cur,
                               xw + (int64_t)(j - 1) * ldw, dst, N, stride, tp);
        }
        WL_HIP(ctx, hipGetLastError());
        cur = dst;
    }
    ctx->last_kernel = "k_imodwt_step";
    return WL_OK;
}

// ---- threshold! ------------------------------------------------------------------------------------
// (threshold_one: wl_dev.h -- the level-1 kernel of the translation-invariant batch applies it at its stores)
template <typename T, typename C>
__global__ void __launch_bounds__(EXT_THREADS) k_threshold(T *__restrict__ x, int64_t n, int th, C t, int vec_ok)
{
    ew_inplace<T>(x, n, vec_ok, [=](T v, int64_t) { return threshold_one<T, C>(v, th, t); });
}

// ---- order statistics: MSB-first radix select on monotone integer keys -------------------------------
// Two ranks are resolved together (the two middle order statistics of an even-length median).
struct SelState {
    unsigned long long prefix[2];
    unsigned long long k[2];            // remaining 0-based rank inside the current prefix class
    unsigned int hist[2][256];
    unsigned int nan_count;
    unsigned int pad;
    double result;                      // median / order statistic as Float64
    double value[2];                    // the two selected values
    unsigned long long less, equal;     // counts relative to value[0] (threshold_biggest)
};

template <typename T> struct KeyOf;
template <> struct KeyOf<float> {
    typedef unsigned int U;
    static constexpr int BYTES = 4;
    __device__ static U key(float v, int absmode)
    {
        U b = __float_as_uint(v);
        if (absmode) return b & 0x7fffffffu;
        return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    __device__ static float val(U k, int absmode)
    {
        if (absmode) return __uint_as_float(k);
        return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
    }
};
template <> struct KeyOf<double> {
    typedef unsigned long long U;
    static constexpr int BYTES = 8;
    __device__ static U key(double v, int absmode)
    {
        U b = (U)__double_as_longlong(v);
        if (absmode) return b & 0x7fffffffffffffffull;
        return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    }
    __device__ static double val(U k, int absmode)
    {
        if (absmode) return __longlong_as_double((long long)k);
        return __longlong_as_double((long long)((k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k));
    }
};

__global__ void k_sel_init(SelState *s, unsigned long long k0, unsigned long long k1)
{
    const int t = threadIdx.x;
    if (t < 256) { s->hist[0][t] = 0; s->hist[1][t] = 0; }
    if (t == 0) {
        s->prefix[0] = s->prefix[1] = 0; s->k[0] = k0; s->k[1] = k1; s->nan_count = 0; s->result = 0.0;
        s->value[0] = s->value[1] = 0.0; s->less = s->equal = 0;
    }
}
// histogram of byte `pass` (0 = most significant) among keys whose higher bytes equal the rank's prefix
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_sel_hist(const T *__restrict__ v, int64_t n, int pass, int absmode, SelState *s)
{
    typedef typename KeyOf<T>::U U;
    constexpr int NB = KeyOf<T>::BYTES;
    __shared__ unsigned int lh[2][256];
    lh[0][threadIdx.x] = 0;
    lh[1][threadIdx.x] = 0;
    __syncthreads();
    const int shift = 8 * (NB - 1 - pass);
    const U p0 = (U)s->prefix[0], p1 = (U)s->prefix[1];
    const bool same = (p0 == p1);
    unsigned int nans = 0;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthr) {
        const T x = v[i];
        if (pass == 0 && x != x) ++nans;
        const U key = KeyOf<T>::key(x, absmode);
        const U hi = (pass == 0) ? (U)0 : (key >> (shift + 8));
        const unsigned b = (unsigned)((key >> shift) & 0xff);
        if (pass == 0 || hi == (p0 >> (shift + 8))) atomicAdd(&lh[0][b], 1u);
        if (!same && hi == (p1 >> (shift + 8))) atomicAdd(&lh[1][b], 1u);
    }
    __syncthreads();
    if (lh[0][threadIdx.x]) atomicAdd(&s->hist[0][threadIdx.x], lh[0][threadIdx.x]);
    if (!same && lh[1][threadIdx.x]) atomicAdd(&s->hist[1][threadIdx.x], lh[1][threadIdx.x]);
    if (nans) atomicAdd(&s->nan_count, nans);
}
// pick the bucket holding each rank, extend the prefix, clear the histograms
template <typename T>
__global__ void k_sel_scan(int pass, SelState *s)
{
    constexpr int NB = KeyOf<T>::BYTES;
    const int shift = 8 * (NB - 1 - pass);
    if (threadIdx.x == 0) {
        const bool same = (s->prefix[0] == s->prefix[1]);
        for (int r = 0; r < 2; ++r) {
            const unsigned int *h = s->hist[(same && r == 1) ? 0 : r];
            unsigned long long k = s->k[r], cum = 0;
            int b = 0;
            for (; b < 255; ++b) {
                if (k < cum + h[b]) break;
                cum += h[b];
            }
            s->k[r] = k - cum;
            s->prefix[r] |= ((unsigned long long)b) << shift;
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) { s->hist[0][threadIdx.x] = 0; s->hist[1][threadIdx.x] = 0; }
}
// median!: odd n -> the middle order statistic, even n -> middle(a, b) = a/2 + b/2; NaN anywhere -> NaN
template <typename T>
__global__ void k_sel_finish(SelState *s, int absmode, int is_median, T *result_t)
{
    typedef typename KeyOf<T>::U U;
    const T a = KeyOf<T>::val((U)s->prefix[0], absmode), b = KeyOf<T>::val((U)s->prefix[1], absmode);
    s->value[0] = (double)a;
    s->value[1] = (double)b;
    T m = a;
    if (is_median) {
        if (s->prefix[0] != s->prefix[1]) m = a / 2 + b / 2;
        if (s->nan_count) m = (T)NAN;
    }
    s->result = (double)m;
    if (result_t) *result_t = m;
}
// mad!: y[i] = abs(y[i] - m)
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_absdev(T *__restrict__ y, int64_t n, const T *m, int vec_ok)
{
    const T mm = *m;
    ew_inplace<T>(y, n, vec_ok, [=](T v, int64_t) { const T d = v - mm; return d < 0 ? -d : d; });
}
// BiggestTH: counts of |x| below / equal to the cut, then the zeroing passes
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_count_cut(const T *__restrict__ x, int64_t n, SelState *s)
{
    const T a = (T)s->value[0];
    unsigned long long less = 0, eq = 0;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthr) {
        const T v = x[i] < 0 ? -x[i] : x[i];
        less += (v < a);
        eq += (v == a);
    }
    if (less) atomicAdd(&s->less, less);
    if (eq) atomicAdd(&s->equal, eq);
}
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_zero_below(T *__restrict__ x, int64_t n, const SelState *s, int inclusive)
{
    const T a = (T)s->value[0];
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += nthr) {
        const T v = x[i] < 0 ? -x[i] : x[i];
        if (v < a || (inclusive && v == a)) x[i] = (T)0;
    }
}
// ties at the cut: zero the first `need` entries (index order) whose magnitude equals the cut; one wave, ordered
template <typename T>
__global__ void __launch_bounds__(64) k_zero_ties(T *__restrict__ x, int64_t n, const SelState *s, unsigned long long need)
{
    const T a = (T)s->value[0];
    const int lane = threadIdx.x;
    unsigned long long done = 0;
    for (int64_t base = 0; base < n && done < need; base += 64) {
        const int64_t i = base + lane;
        const T v = (i < n) ? (x[i] < 0 ? -x[i] : x[i]) : (T)-1;
        const bool tie = (i < n) && (v == a);
        const unsigned long long mask = __ballot(tie);
        const unsigned long long before = mask & ((1ull << lane) - 1ull);
        if (tie && done + (unsigned long long)__popcll(before) < need) x[i] = (T)0;
        done += (unsigned long long)__popcll(mask);
    }
}

int ensure_aux(wl_ctx *ctx)
{
    if (ctx->aux) return WL_OK;
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, 8192);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return WL_ENOMEM; }
    ctx->aux = p;
    return WL_OK;
}

// Small arrays (<= 8192 elements, e.g. the noise estimate of a 2-D array: the lower half of one column): the whole
// median / mad! in ONE workgroup with the keys in LDS -- the multi-launch radix select above is launch-bound there.
template <typename T>
__device__ typename KeyOf<T>::U lds_select(const typename KeyOf<T>::U *keys, int n, unsigned long long k, unsigned int *hist,
                                             unsigned long long *sh)
{
    typedef typename KeyOf<T>::U U;
    constexpr int NB = KeyOf<T>::BYTES;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) { sh[0] = 0; sh[1] = k; }
    for (int pass = 0; pass < NB; ++pass) {
        const int shift = 8 * (NB - 1 - pass);
        for (int b = tid; b < 256; b += nthr) hist[b] = 0;
        __syncthreads();
        const U prefix = (U)sh[0];
        for (int i = tid; i < n; i += nthr) {
            const U key = keys[i];
            if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(unsigned)((key >> shift) & 0xff)], 1u);
        }
        __syncthreads();
        // the bin that holds rank kk: first b with kk < hist[0] + ... + hist[b] (255 when none).  One wave scans the 256 bins, four
        // per lane (thread 0 walking them one dependent LDS read at a time cost ~8 us per pass: 16 passes per mad!)
        if (tid < 64) {
            const unsigned int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const unsigned int mine = h0 + h1 + h2 + h3;
            unsigned int incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned int o = __shfl_up(incl, d, 64);
                if (tid >= d) incl += o;
            }
            const unsigned long long kk = sh[1];
            const unsigned long long excl = incl - mine;
            const bool hit = kk < (unsigned long long)incl;
            const unsigned long long m = __ballot(hit);
            const int first = m ? (__ffsll((long long)m) - 1) : 64;
            if (tid == first) {
                int b = 4 * tid;
                unsigned long long cum = excl;
                if (kk >= cum + h0) { cum += h0; ++b; if (kk >= cum + h1) { cum += h1; ++b; if (kk >= cum + h2) { cum += h2; ++b; } } }
                sh[1] = kk - cum;
                sh[0] = sh[0] | (((unsigned long long)b) << shift);
            } else if (first == 64 && tid == 63) {       // (cannot happen for kk < n; mirrors the serial walk: bin 255, all lower bins skipped)
                sh[1] = kk - (excl + h0 + h1 + h2);
                sh[0] = sh[0] | (255ull << shift);
            }
        }
        __syncthreads();
    }
    const U r = (U)sh[0];
    __syncthreads();                // the next call re-initialises sh
    return r;
}
template <typename T>
__device__ T lds_median(typename KeyOf<T>::U *keys, int n, unsigned int *hist, unsigned long long *sh, unsigned int nan_count)
{
    typedef typename KeyOf<T>::U U;
    const unsigned long long k1 = (unsigned long long)(n / 2), k0 = (n & 1) ? k1 : k1 - 1;
    const U p0 = lds_select<T>(keys, n, k0, hist, sh);
    const U p1 = (k1 == k0) ? p0 : lds_select<T>(keys, n, k1, hist, sh);
    const T a = KeyOf<T>::val(p0, 0), b = KeyOf<T>::val(p1, 0);
    T m = (p0 != p1) ? (a / 2 + b / 2) : a;
    if (nan_count) m = (T)NAN;
    return m;
}
// do_mad = 0: result = median(v) (v untouched);  1: mad!(v): v[i] = abs(v[i] - median(v)), result = median of that
template <typename T>
__global__ void __launch_bounds__(1024) k_mad_lds(T *v, int n, int do_mad, SelState *s)
{
    typedef typename KeyOf<T>::U U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    U *keys = reinterpret_cast<U *>(smem_raw);
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long sh[2];
    __shared__ unsigned int nans;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) nans = 0;
    __syncthreads();
    unsigned int mynan = 0;
    for (int i = tid; i < n; i += nthr) {
        const T x = v[i];
        if (x != x) ++mynan;
        keys[i] = KeyOf<T>::key(x, 0);
    }
    if (mynan) atomicAdd(&nans, mynan);
    __syncthreads();
    T m = lds_median<T>(keys, n, hist, sh, nans);
    if (do_mad) {
        __syncthreads();
        for (int i = tid; i < n; i += nthr) {
            const T d = KeyOf<T>::val(keys[i], 0) - m;
            const T ad = d < 0 ? -d : d;
            v[i] = ad;
            keys[i] = KeyOf<T>::key(ad, 0);
        }
        __syncthreads();
        m = lds_median<T>(keys, n, hist, sh, nans);
    }
    if (tid == 0) s->result = (double)m;
}
// keys of n elements in dynamic LDS next to ~1 KiB of static LDS: stay under the 64 KiB a kernel gets without
// hipFuncSetAttribute (Float64 keys are 8 bytes)
template <typename T>
constexpr int64_t mad_lds_max() { return sizeof(T) == 4 ? 8192 : 4096; }

template <typename T>
int mad_small(wl_ctx *ctx, hipStream_t st, T *v, int64_t n, int do_mad, double *result_host)
{
    int rc = ensure_aux(ctx);
    if (rc != WL_OK) return rc;
    SelState *s = (SelState *)ctx->aux;
    const int threads = n >= 2048 ? 1024 : (n >= 256 ? 256 : 64);
    hipLaunchKernelGGL((k_mad_lds<T>), dim3(1), dim3(threads), (size_t)n * sizeof(typename KeyOf<T>::U), st, v, (int)n, do_mad, s);
    WL_HIP(ctx, hipGetLastError());
    if (result_host) {              // (nullptr: the result stays in the selection state on the device, nobody waits)
        WL_HIP(ctx, hipMemcpyAsync(result_host, &s->result, sizeof(double), hipMemcpyDeviceToHost, st));
        WL_HIP(ctx, hipStreamSynchronize(st));
    }
    return WL_OK;
}

// enqueue the selection of ranks k0 <= k1 (0-based) of v[0..n); leaves prefix[] resolved in the state
template <typename T>
int select_ranks(wl_ctx *ctx, hipStream_t st, const T *v, int64_t n, unsigned long long k0, unsigned long long k1, int absmode)
{
    SelState *s = (SelState *)ctx->aux;
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(256), 0, st, s, k0, k1);
    const unsigned nb = ext_blocks(n, 8, ctx->cu_count);
    for (int pass = 0; pass < KeyOf<T>::BYTES; ++pass) {
        hipLaunchKernelGGL((k_sel_hist<T>), dim3(nb), dim3(EXT_THREADS), 0, st, v, n, pass, absmode, s);
        hipLaunchKernelGGL((k_sel_scan<T>), dim3(1), dim3(256), 0, st, pass, s);
    }
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

template <typename T>
int median_impl(wl_ctx *ctx, hipStream_t st, const T *v, int64_t n, double *result_host, T *result_dev)
{
    int rc = ensure_aux(ctx);
    if (rc != WL_OK) return rc;
    SelState *s = (SelState *)ctx->aux;
    const unsigned long long k1 = (unsigned long long)(n / 2), k0 = (n & 1) ? k1 : k1 - 1;
    rc = select_ranks<T>(ctx, st, v, n, k0, k1, 0);
    if (rc != WL_OK) return rc;
    hipLaunchKernelGGL((k_sel_finish<T>), dim3(1), dim3(1), 0, st, s, 0, 1, result_dev);
    WL_HIP(ctx, hipGetLastError());
    if (result_host) {
        WL_HIP(ctx, hipMemcpyAsync(result_host, &s->result, sizeof(double), hipMemcpyDeviceToHost, st));
        WL_HIP(ctx, hipStreamSynchronize(st));
    }
    return WL_OK;
}

template <typename T>
int biggest_impl(wl_ctx *ctx, hipStream_t st, T *x, int64_t n, int64_t m)
{
    if (m > n) m = n;
    const int64_t nz = n - m;                      // entries to clear
    if (nz <= 0) return WL_OK;
    int rc = ensure_aux(ctx);
    if (rc != WL_OK) return rc;
    SelState *s = (SelState *)ctx->aux;
    rc = select_ranks<T>(ctx, st, x, n, (unsigned long long)(nz - 1), (unsigned long long)(nz - 1), 1);
    if (rc != WL_OK) return rc;
    hipLaunchKernelGGL((k_sel_finish<T>), dim3(1), dim3(1), 0, st, s, 1, 0, (T *)nullptr);
    const unsigned nb = ext_blocks(n, 4, ctx->cu_count);
    hipLaunchKernelGGL((k_count_cut<T>), dim3(nb), dim3(EXT_THREADS), 0, st, x, n, s);
    WL_HIP(ctx, hipGetLastError());
    unsigned long long cnt[2] = {0, 0};
    WL_HIP(ctx, hipMemcpyAsync(cnt, &s->less, sizeof(cnt), hipMemcpyDeviceToHost, st));
    WL_HIP(ctx, hipStreamSynchronize(st));
    const unsigned long long need = (unsigned long long)nz - cnt[0];       // ties at the cut that must go
    if (need >= cnt[1]) {
        hipLaunchKernelGGL((k_zero_below<T>), dim3(nb), dim3(EXT_THREADS), 0, st, x, n, s, 1);
    } else {
        hipLaunchKernelGGL((k_zero_ties<T>), dim3(1), dim3(64), 0, st, x, n, s, need);
        hipLaunchKernelGGL((k_zero_below<T>), dim3(nb), dim3(EXT_THREADS), 0, st, x, n, s, 0);
    }
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

// ---- circshift / arrayadd! / rmul! -------------------------------------------------------------------
struct Shift3 { int64_t d[3]; int64_t s[3]; };
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_circshift(T *__restrict__ b, const T *__restrict__ a, int64_t n, Shift3 p)
{
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += nthr) {
        const int64_t i0 = e % p.d[0], r = e / p.d[0], i1 = r % p.d[1], i2 = r / p.d[1];
        int64_t j0 = i0 - p.s[0], j1 = i1 - p.s[1], j2 = i2 - p.s[2];
        if (j0 < 0) j0 += p.d[0];
        if (j1 < 0) j1 += p.d[1];
        if (j2 < 0) j2 += p.d[2];
        b[e] = a[j0 + p.d[0] * (j1 + p.d[1] * j2)];
    }
}
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_arrayadd(T *__restrict__ y, const T *__restrict__ z, int64_t n, int vec_ok)
{
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    const VT *zv = reinterpret_cast<const VT *>(z);
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x, gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nv = vec_ok ? n / V : 0;
    VT *yv = reinterpret_cast<VT *>(y);
    for (int64_t i = gid; i < nv; i += nthr) {
        VT a = yv[i];
        const VT b = zv[i];
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] = a[e] + b[e];
        yv[i] = a;
    }
    for (int64_t i = nv * V + gid; i < n; i += nthr) y[i] = y[i] + z[i];
}
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_rmul(T *__restrict__ y, int64_t n, double s, int vec_ok)
{
    ew_inplace<T>(y, n, vec_ok, [=](T v, int64_t) { return (T)((double)v * s); });
}


// ---- translation-invariant denoising as ONE batch (denoising.jl:36-67) ---------------------------------------------------
// spin i (1-based) shifts dimension d by nspin2circ(nspin, i)[d] (denoising.jl:112-121: first dimension fastest)
struct TiGeom { int64_t n0, n1, N; int64_t nsp0, nsp1; int64_t b0; };
__device__ __forceinline__ void ti_shift_of(const TiGeom &g, int64_t spin0, int64_t &s0, int64_t &s1)
{
    s0 = (spin0 % g.nsp0) % g.n0;
    s1 = ((spin0 / g.nsp0) % g.nsp1) % g.n1;
}
// Z[b] = circshift(x, +shift(b0 + b)): z[i] = x[i - shift] (Util.circshift!, util_main.jl:105-130).
// grid: x = groups of 4 rows, y = column, z = spin (no integer division per element; 16-byte stores, the shifted reads are
// four scalar loads from a contiguous run)
template <typename T>
__global__ void __launch_bounds__(256) k_ti_shift(T *__restrict__ Z, const T *__restrict__ x, TiGeom g)
{
    typedef T V4 __attribute__((ext_vector_type(4)));
    const int64_t b = blockIdx.z;
    int64_t s0, s1;
    ti_shift_of(g, g.b0 + b, s0, s1);
    const int n0 = (int)g.n0, sh = (int)s0;
    const bool vec = (n0 & 3) == 0 && (reinterpret_cast<uintptr_t>(Z) % (4 * sizeof(T))) == 0;
    // eight columns per workgroup (a workgroup per column is a single 16-byte store per thread: launch-rate bound)
    for (int64_t i1 = (int64_t)blockIdx.y * 8; i1 < g.n1 && i1 < (int64_t)blockIdx.y * 8 + 8; ++i1) {
        int64_t j1 = i1 - s1;
        if (j1 < 0) j1 += g.n1;
        const T *src = x + g.n0 * j1;
        T *dst = Z + b * g.N + g.n0 * i1;
        for (int i0 = 4 * (int)(blockIdx.x * blockDim.x + threadIdx.x); i0 < n0; i0 += 4 * (int)(gridDim.x * blockDim.x)) {
            T v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int j0 = i0 + e - sh;
                if (j0 < 0) j0 += n0;
                v[e] = (i0 + e < n0) ? src[j0] : (T)0;
            }
            if (vec) {
                *reinterpret_cast<V4 *>(dst + i0) = V4{v[0], v[1], v[2], v[3]};
            } else {
                for (int e = 0; e < 4 && i0 + e < n0; ++e) dst[i0 + e] = v[e];
            }
        }
    }
}
// y += circshift(Z[b], -shift(b0 + b)) for b = 0 .. nb-1 IN THAT ORDER (arrayadd! once per spin: the summation order
// of the reference, so the sums carry the same roundings).  grid: x = groups of 4 rows, y = column.  The shifts of the
// batch sit in LDS (no integer division per spin and lane); the spins are read eight at a time, loads before the adds.
template <typename T>
__global__ void __launch_bounds__(256) k_ti_accumulate(T *__restrict__ y, const T *__restrict__ Z, TiGeom g, int64_t nb, int first)
{
    __shared__ int sh0[256], sh1[256];
    const int n0 = (int)g.n0, n1 = (int)g.n1, i1 = (int)blockIdx.y;
    for (int64_t bb0 = 0; bb0 < nb; bb0 += 256) {
        const int nbb = (int)((nb - bb0 < 256) ? (nb - bb0) : 256);
        __syncthreads();
        if ((int)threadIdx.x < nbb) {
            int64_t s0, s1;
            ti_shift_of(g, g.b0 + bb0 + threadIdx.x, s0, s1);
            sh0[threadIdx.x] = (int)s0;
            sh1[threadIdx.x] = (int)s1;
        }
        __syncthreads();
        // a thread owns rows ib + tid + 256 e (e < 4): consecutive lanes read consecutive addresses of every shifted plane (the
        // shifts are arbitrary, so 16-byte loads are out; four rows per lane side by side cost four partial lines per load)
        for (int ib = 4 * (int)(blockIdx.x * blockDim.x); ib < n0; ib += 4 * (int)(gridDim.x * blockDim.x)) {
            const int ibt = ib + (int)threadIdx.x;
            T acc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = ((first && bb0 == 0) || ibt + 256 * e >= n0) ? (T)0 : y[ibt + 256 * e + (int64_t)n0 * i1];
            for (int b8 = 0; b8 < nbb; b8 += 8) {
                T v[8][4];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (b8 + u < nbb) {
                        int j1 = i1 + sh1[b8 + u];
                        if (j1 >= n1) j1 -= n1;
                        const T *zp = Z + (bb0 + b8 + u) * g.N + (int64_t)n0 * j1;
                        const int s = sh0[b8 + u];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int j0 = ibt + 256 * e + s;
                            if (j0 >= n0) j0 -= n0;
                            v[u][e] = (ibt + 256 * e < n0) ? zp[j0] : (T)0;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (b8 + u < nbb) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + v[u][e];
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ibt + 256 * e < n0) y[ibt + 256 * e + (int64_t)n0 * i1] = acc[e];
        }
    }
}
// threshold!(x, TH, sigma * t_unit) with sigma = mad / 0.6745 read from the device (noisest, denoising.jl:92-101): the
// product is formed in Float64 exactly as Julia does for a Float64 dnt.t, whatever the element type
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_threshold_dev(T *__restrict__ x, int64_t n, int th, const double *__restrict__ mad_dev,
                                                               double t_unit, double sigma_host, int vec_ok)
{
    const double sigma = (sigma_host >= 0) ? sigma_host : (*mad_dev / 0.6745);
    const double t = sigma * t_unit;
    ew_inplace<T>(x, n, vec_ok, [=](T v, int64_t) { return threshold_one<T, double>(v, th, t); });
}
// the same on the approximation quadrant [0, h0) x [0, h1) of every plane of a batch (leading dimension n0, plane stride N): the
// details of level 1 were thresholded by the kernel that produced them (SrcView::th)
template <typename T>
__global__ void __launch_bounds__(256) k_threshold_dev_quadrant(T *__restrict__ x, int64_t n0, int64_t N, int64_t h0, int th,
                                                                const double *__restrict__ mad_dev, double t_unit, double sigma_host)
{
    const double sigma = (sigma_host >= 0) ? sigma_host : (*mad_dev / 0.6745);
    const double t = sigma * t_unit;
    T *col = x + (int64_t)blockIdx.z * N + (int64_t)blockIdx.y * n0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < h0; i += (int64_t)gridDim.x * blockDim.x)
        col[i] = threshold_one<T, double>(col[i], th, t);
}
template <typename T>
__global__ void __launch_bounds__(EXT_THREADS) k_copy_range(T *__restrict__ dst, const T *__restrict__ src, int64_t n)
{
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += nthr) dst[e] = src[e];
}

template <typename T>
int denoise_ti_impl(wl_ctx *ctx, hipStream_t st, T *y, const T *x, int ndims, const int64_t *dims, const double *qmf, int flen, int L,
                    int th, double t_unit, const int64_t *nspin, double sigma_host)
{
    const int64_t n0 = dims[0], n1 = (ndims == 2) ? dims[1] : 1, N = n0 * n1;
    const int64_t nsp0 = nspin[0], nsp1 = (ndims == 2) ? nspin[1] : 1, pns = nsp0 * nsp1;
    Taps<T> taps;
    make_taps<T>(qmf, flen, taps);
    int rc = ensure_aux(ctx);
    if (rc != WL_OK) return rc;
    SelState *sel = (SelState *)ctx->aux;
    // spins per batch: the whole set unless the buffers (2 N B for the shifted copies and their coefficients, plus the
    // transform workspace of the batch box) would pass the cap
    // Virtual shifts (Float32 square images on the LDS-exchange level kernel): a circular shift along dim 2 is an offset of the
    // column index inside the level-1 kernel, so only the nsp0 row-shifted copies of x are materialised (ZR) instead of
    // nsp0 * nsp1 shifted planes -- for 8 x 8 spins of a 2048^2 image 128 MiB written instead of 1 GiB (265 us of 2.4 ms).
    bool virt = false;
    if constexpr (sizeof(T) == 4) {
        virt = ndims == 2 && L >= 1 && ctx->path == 0 && (flen % 2) == 0 && flen <= 10 && opt("WL_TI_VIRTSHIFT", 1) != 0 &&
               fwd2d_lds_ok(flen, 1, n0, n1) && (n0 % 4) == 0 && nsp1 <= n1 && nsp0 <= n0 && nsp0 < (1 << 20);
    }
    const size_t zr_elems = virt ? (size_t)N * (size_t)nsp0 : 0;
    const size_t cap = (size_t)opt("WL_TI_WS_CAP_MB", 8192) << 20;
    // (one spin: no shifted copy Z, see below)
    const size_t ncopies = (pns == 1) ? 1 : 2;
    auto need = [&](int64_t B) { return (ws_elems(N * B, ndims) + ncopies * (size_t)N * B + zr_elems + (size_t)n0 + 64) * sizeof(T); };
    int64_t B = pns;
    while (B > 1 && need(B) > cap) B = (B + 1) / 2;
    if (B > 65535) B = 65535;
    rc = wl_ensure_ws(ctx, need(B), st, true);
    while (rc == WL_ENOMEM && B > 1) {                      // (another allocator may own most of the HBM: smaller groups of spins)
        B = (B + 1) / 2;
        rc = wl_ensure_ws(ctx, need(B), st, true);
    }
    if (rc != WL_OK) return rc;
    T *tw = (T *)ctx->ws;                                   // transform workspace of the batch box (with the generic buffers)
    T *Z = tw + ws_elems(N * B, ndims);
    T *XT = Z + (pns == 1 ? 0 : N * B);
    T *ZR = XT + N * B;                                     // row-shifted copies (virtual shifts only)
    T *dr = ZR + zr_elems;                                  // detail range of the noise estimate (n0/2 samples)
    const unsigned nbk = ext_blocks(N * B, 4, ctx->cu_count);

    // ---- sigma = noisest(x, wt): level-1 transform, MAD of y1[detailrange(y1, 1)] (linear indexing: first column) ----
    if (!(sigma_host >= 0)) {
        BoxSpec b1;
        b1.nd = ndims; b1.nt = ndims;
        b1.dims[0] = n0; b1.dims[1] = n1; b1.dims[2] = 1;
        b1.full = dense_strides(b1.dims);
        if (n0 < 2 || (n0 % 2) != 0 || (ndims == 2 && (n1 % 2) != 0)) return WL_EINVAL_SIZE;
        rc = filter_fwd_levels<T>(tw, true, ctx->cu_count, ctx->path, st, b1, XT, x, taps, 1, &ctx->last_kernel, &ctx->last_hip);
        if (rc != WL_OK) return rc;
        const int64_t lo = (int64_t)llround((double)n0 / 2 + 1) - 1, hi = n0;        // detailrange(n0, 1), 0-based half open
        const int64_t nd = hi - lo;
        hipLaunchKernelGGL((k_copy_range<T>), dim3(ext_blocks(nd, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, dr, XT + lo, nd);
        if (nd <= mad_lds_max<T>()) {
            rc = mad_small<T>(ctx, st, dr, nd, 1, nullptr);
            if (rc != WL_OK) return rc;
        } else {
            void *mdev = (char *)ctx->aux + 4096;
            rc = median_impl<T>(ctx, st, dr, nd, nullptr, (T *)mdev);
            if (rc != WL_OK) return rc;
            hipLaunchKernelGGL((k_absdev<T>), dim3(ext_blocks(nd, 4, ctx->cu_count)), dim3(EXT_THREADS), 0, st, dr, nd, (const T *)mdev, vec_ok16(dr));
            rc = median_impl<T>(ctx, st, dr, nd, nullptr, (T *)nullptr);
            if (rc != WL_OK) return rc;
        }
    }
    // ---- one spin of shift zero (the plain, not translation-invariant denoise routed here so that sigma stays on the device):
    //      y = idwt(threshold!(dwt(x))) with no shifted copy, no accumulation and no scaling -- the reference's own sequence
    //      (denoising.jl:69-80), signed zeros included, and a quarter of the batch path's memory ----
    if (pns == 1) {
        BoxSpec b1;
        b1.nd = ndims; b1.nt = ndims;
        b1.dims[0] = n0; b1.dims[1] = n1; b1.dims[2] = 1;
        b1.full = dense_strides(b1.dims);
        const T *coef_src = x;
        if (L > 0) {
            rc = filter_fwd_levels<T>(tw, true, ctx->cu_count, ctx->path, st, b1, XT, x, taps, L, &ctx->last_kernel, &ctx->last_hip);
            if (rc != WL_OK) return rc;
            coef_src = XT;
        } else {
            hipLaunchKernelGGL((k_copy_range<T>), dim3(ext_blocks(N, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, XT, x, N);
            coef_src = XT;
        }
        hipLaunchKernelGGL((k_threshold_dev<T>), dim3(ext_blocks(N, 4, ctx->cu_count)), dim3(EXT_THREADS), 0, st, XT, N, th, &sel->result, t_unit,
                           sigma_host, vec_ok16(XT));
        if (L > 0) {
            const char *kn = nullptr;
            rc = filter_inv_levels<T>(tw, true, ctx->cu_count, ctx->path, st, b1, y, coef_src, taps, L, &kn, &ctx->last_hip);
            if (rc != WL_OK) return rc;
        } else {
            hipLaunchKernelGGL((k_copy_range<T>), dim3(ext_blocks(N, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, y, coef_src, N);
        }
        ctx->last_kernel = "denoise_one_spin";
        WL_HIP(ctx, hipGetLastError());
        return WL_OK;
    }
    // ---- the spins, B at a time ----
    BoxSpec bb;
    bb.nd = ndims + 1; bb.nt = ndims;
    bb.dims[0] = n0; bb.dims[1] = (ndims == 2) ? n1 : 0; bb.dims[2] = 1;
    TiGeom g;
    g.n0 = n0; g.n1 = n1; g.N = N; g.nsp0 = nsp0; g.nsp1 = nsp1;
    const unsigned gxs = (unsigned)((n0 / 4 + 255) / 256 > 0 ? ((n0 / 4 + 255) / 256 > 64 ? 64 : (n0 / 4 + 255) / 256) : 1);
    if (virt) {
        TiGeom gr = g;                                      // spins 0 .. nsp0-1 shift dim 1 only
        gr.nsp1 = 1; gr.b0 = 0;
        hipLaunchKernelGGL((k_ti_shift<T>), dim3(gxs, (unsigned)((n1 + 7) / 8), (unsigned)nsp0), dim3(256), 0, st, ZR, x, gr);
    }
    for (int64_t b0 = 0; b0 < pns; b0 += B) {
        const int64_t nb = (pns - b0 < B) ? (pns - b0) : B;
        g.b0 = b0;
        if (ndims == 2) { bb.dims[2] = nb; }
        else { bb.dims[1] = nb; bb.dims[2] = 1; }
        bb.full = dense_strides(bb.dims);
        bool shifted = false, thresholded_l1 = false;
        int64_t th_c0 = n0, th_c1 = n1;                      // what the level kernels left unthresholded: the low corner of every plane
        if (virt) {
            // plane p of this group = copy (b0 + p) % nsp0 of ZR with its columns rotated by (b0 + p) / nsp0
            tl_srcview.mod = (int)nsp0; tl_srcview.spin0 = b0; tl_srcview.used = 0; tl_srcview.corner0 = n0; tl_srcview.corner1 = n1;
            // ... and that launch thresholds the level-1 details as it stores them (3/4 of all coefficients)
            // (hard: one Float32 compare; soft / semisoft / Stein: the same cut, Float64 only for the survivors -- ThCut, wl_dev.h)
            const bool fuse_th = opt("WL_TI_FUSE_TH", 1) != 0 && th >= WL_TH_HARD && th <= WL_TH_STEIN &&
                                 (th == WL_TH_HARD || opt("WL_TI_FUSE_SOFT", 1) != 0);
            tl_srcview.th = fuse_th ? th : -1; tl_srcview.t_unit = t_unit; tl_srcview.sigma_host = sigma_host; tl_srcview.mad_dev = &sel->result;
            rc = filter_fwd_levels<T>(tw, true, ctx->cu_count, ctx->path, st, bb, XT, ZR, taps, L, &ctx->last_kernel, &ctx->last_hip);
            shifted = tl_srcview.used != 0;
            th_c0 = tl_srcview.corner0; th_c1 = tl_srcview.corner1;
            tl_srcview.mod = 0; tl_srcview.th = -1;
            // WL_RETRY_NOVIEW: no view-aware tier took level 1 -- nothing was enqueued (and nothing read from ZR): materialise below
            if (rc == WL_RETRY_NOVIEW) { rc = WL_OK; shifted = false; }
            if (rc != WL_OK) return rc;
            thresholded_l1 = shifted && fuse_th;
        }
        if (!shifted) {                                     // (materialised shifted planes: every other case)
            hipLaunchKernelGGL((k_ti_shift<T>), dim3(gxs, (unsigned)((n1 + 7) / 8), (unsigned)nb), dim3(256), 0, st, Z, x, g);
            if (L > 0) {
                rc = filter_fwd_levels<T>(tw, true, ctx->cu_count, ctx->path, st, bb, XT, Z, taps, L, &ctx->last_kernel, &ctx->last_hip);
                if (rc != WL_OK) return rc;
            }
        }
        // L == 0: dwt / idwt are copies (transforms_filter.jl:36-38), so the shifted signal itself is thresholded
        T *const coef = (L == 0) ? Z : XT;
        if (thresholded_l1) {
            const int64_t h0 = th_c0, h1 = th_c1;
            if (h0 > 0 && h1 > 0)
                hipLaunchKernelGGL((k_threshold_dev_quadrant<T>), dim3((unsigned)((h0 + 255) / 256 > 8 ? 8 : (h0 + 255) / 256), (unsigned)h1, (unsigned)nb),
                                   dim3(256), 0, st, coef, n0, N, h0, th, &sel->result, t_unit, sigma_host);
        } else {
            hipLaunchKernelGGL((k_threshold_dev<T>), dim3(nbk), dim3(EXT_THREADS), 0, st, coef, N * nb, th, &sel->result, t_unit, sigma_host,
                               vec_ok16(coef));
        }
        if (L > 0) {
            const char *kn = nullptr;
            rc = filter_inv_levels<T>(tw, true, ctx->cu_count, ctx->path, st, bb, Z, XT, taps, L, &kn, &ctx->last_hip);
            if (rc != WL_OK) return rc;
        }
        hipLaunchKernelGGL((k_ti_accumulate<T>), dim3(gxs, (unsigned)n1), dim3(256), 0, st, y, Z, g, nb, b0 == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL((k_rmul<T>), dim3(ext_blocks(N, 4, ctx->cu_count)), dim3(EXT_THREADS), 0, st, y, N, 1.0 / (double)pns, vec_ok16(y));
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

// ---- translation-invariant denoise with a lifting scheme (denoising.jl:36-67 with wt::GLS) --------------------------------
// The same device-resident sequence as denoise_ti_impl -- sigma from the level-1 transform without a host round trip, the spins
// shifted / transformed / thresholded / inverted / un-shifted / accumulated B at a time -- with the lifting transforms of the
// library: a batch of shifted SIGNALS is one batched-lines call (the fused line kernels over all spins), a batch of shifted
// IMAGES is one 2-D lifting transform per plane (the 2-D lifting kernels are not plane-batched).
template <typename T>
int denoise_ti_lifting_impl(wl_ctx *ctx, hipStream_t st, T *y, const T *x, int ndims, const int64_t *dims, const LiftScheme<T> &scf,
                            const LiftScheme<T> &sci, int L, int th, double t_unit, const int64_t *nspin, double sigma_host)
{
    const int64_t n0 = dims[0], n1 = (ndims == 2) ? dims[1] : 1, N = n0 * n1;
    const int64_t nsp0 = nspin[0], nsp1 = (ndims == 2) ? nspin[1] : 1, pns = nsp0 * nsp1;
    int rc = ensure_aux(ctx);
    if (rc != WL_OK) return rc;
    SelState *sel = (SelState *)ctx->aux;
    const size_t cap = (size_t)opt("WL_TI_WS_CAP_MB", 8192) << 20;
    // transform workspace: the batched-lines box (1-D) or one plane (2-D), then the shifted copies Z and their coefficients XT
    auto tws = [&](int64_t B) { return ws_elems(ndims == 1 ? N * B : N, 1); };
    auto need = [&](int64_t B) { return (tws(B) + (size_t)2 * N * B + (size_t)n0 + 64) * sizeof(T); };
    int64_t B = pns;
    while (B > 1 && need(B) > cap) B = (B + 1) / 2;
    if (B > 65535) B = 65535;
    rc = wl_ensure_ws(ctx, need(B), st, true);
    while (rc == WL_ENOMEM && B > 1) { B = (B + 1) / 2; rc = wl_ensure_ws(ctx, need(B), st, true); }
    if (rc != WL_OK) return rc;
    T *tw = (T *)ctx->ws;
    T *Z = tw + tws(B);
    T *XT = Z + N * B;
    T *dr = XT + N * B;
    BoxSpec b1;                                              // one signal / image
    b1.nd = ndims; b1.nt = ndims;
    b1.dims[0] = n0; b1.dims[1] = n1; b1.dims[2] = 1;
    b1.full = dense_strides(b1.dims);
    // ---- sigma = noisest(x, wt): level-1 transform, MAD of y1[detailrange(y1, 1)] ----
    if (!(sigma_host >= 0)) {
        if (n0 < 2 || (n0 % 2) != 0 || (ndims == 2 && (n1 % 2) != 0)) return WL_EINVAL_SIZE;
        rc = wl_lifting_box<T>(ctx, st, b1, XT, x, scf, 1, 1);
        if (rc != WL_OK) return rc;
        const int64_t lo = (int64_t)llround((double)n0 / 2 + 1) - 1, hi = n0;
        const int64_t nd = hi - lo;
        hipLaunchKernelGGL((k_copy_range<T>), dim3(ext_blocks(nd, 1, ctx->cu_count)), dim3(EXT_THREADS), 0, st, dr, XT + lo, nd);
        if (nd <= mad_lds_max<T>()) {
            rc = mad_small<T>(ctx, st, dr, nd, 1, nullptr);
            if (rc != WL_OK) return rc;
        } else {
            void *mdev = (char *)ctx->aux + 4096;
            rc = median_impl<T>(ctx, st, dr, nd, nullptr, (T *)mdev);
            if (rc != WL_OK) return rc;
            hipLaunchKernelGGL((k_absdev<T>), dim3(ext_blocks(nd, 4, ctx->cu_count)), dim3(EXT_THREADS), 0, st, dr, nd, (const T *)mdev, vec_ok16(dr));
            rc = median_impl<T>(ctx, st, dr, nd, nullptr, (T *)nullptr);
            if (rc != WL_OK) return rc;
        }
    }
    TiGeom g;
    g.n0 = n0; g.n1 = n1; g.N = N; g.nsp0 = nsp0; g.nsp1 = nsp1;
    const unsigned gxs = (unsigned)((n0 / 4 + 255) / 256 > 0 ? ((n0 / 4 + 255) / 256 > 64 ? 64 : (n0 / 4 + 255) / 256) : 1);
    auto transform = [&](T *dst, const T *src, int64_t nb, const LiftScheme<T> &sc, int fw) -> int {
        if (ndims == 1) {                                    // nb signals = nb lines of one batched call
            BoxSpec bb;
            bb.nd = 2; bb.nt = 1;
            bb.dims[0] = n0; bb.dims[1] = nb; bb.dims[2] = 1;
            bb.full = dense_strides(bb.dims);
            return wl_lifting_box<T>(ctx, st, bb, dst, src, sc, L, fw);
        }
        for (int64_t p = 0; p < nb; ++p) {
            int r = wl_lifting_box<T>(ctx, st, b1, dst + p * N, src + p * N, sc, L, fw);
            if (r != WL_OK) return r;
        }
        return WL_OK;
    };
    for (int64_t b0 = 0; b0 < pns; b0 += B) {
        const int64_t nb = (pns - b0 < B) ? (pns - b0) : B;
        g.b0 = b0;
        hipLaunchKernelGGL((k_ti_shift<T>), dim3(gxs, (unsigned)((n1 + 7) / 8), (unsigned)nb), dim3(256), 0, st, Z, x, g);
        rc = transform(XT, Z, nb, scf, 1);                   // (L = 0: a copy, as the reference's dwt is)
        if (rc != WL_OK) return rc;
        hipLaunchKernelGGL((k_threshold_dev<T>), dim3(ext_blocks(N * nb, 4, ctx->cu_count)), dim3(EXT_THREADS), 0, st, XT, N * nb, th, &sel->result,
                           t_unit, sigma_host, vec_ok16(XT));
        rc = transform(Z, XT, nb, sci, 0);
        if (rc != WL_OK) return rc;
        hipLaunchKernelGGL((k_ti_accumulate<T>), dim3(gxs, (unsigned)n1), dim3(256), 0, st, y, Z, g, nb, b0 == 0 ? 1 : 0);
    }
    hipLaunchKernelGGL((k_rmul<T>), dim3(ext_blocks(N, 4, ctx->cu_count)), dim3(EXT_THREADS), 0, st, y, N, 1.0 / (double)pns, vec_ok16(y));
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

inline int ext_enter(wl_ctx *ctx, int dtype)
{
    if (!ctx) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    return WL_OK;
}

}  // namespace

extern "C" {

int wl_maxmodwttransformlevels(int64_t n)
{
    int l = 0;
    while (n > 1) { n >>= 1; ++l; }
    return l;
}

int wl_modwt(wl_ctx *ctx, int dtype, void *out, int64_t ldo, const void *x, int64_t n, const double *qmf, int flen, int L,
             void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!out || !x || !qmf) return WL_EINVAL_ARG;
    if (flen < 1 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    if (n < 1 || ldo < n) return WL_EDIMS;
    if (L > wl_maxmodwttransformlevels(n)) return WL_EINVAL_SIZE;      // "Too many transform levels (length(x) < 2^L)"
    if (L < 1) return WL_EINVAL_L;                                       // "L must be >= 1"
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32 ? modwt_impl<float>(ctx, st, (float *)out, ldo, (const float *)x, n, qmf, flen, L)
                           : modwt_impl<double>(ctx, st, (double *)out, ldo, (const double *)x, n, qmf, flen, L);
}

int wl_imodwt(wl_ctx *ctx, int dtype, void *x, const void *xw, int64_t ldw, int64_t n, int ncols, const double *qmf, int flen,
              void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!x || !xw || !qmf) return WL_EINVAL_ARG;
    if (flen < 1 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    if (n < 1 || ncols < 1 || ldw < n) return WL_EDIMS;
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32 ? imodwt_impl<float>(ctx, st, (float *)x, (const float *)xw, ldw, n, ncols, qmf, flen)
                           : imodwt_impl<double>(ctx, st, (double *)x, (const double *)xw, ldw, n, ncols, qmf, flen);
}

int wl_threshold(wl_ctx *ctx, int dtype, void *x, int64_t n, int th, double t, int t_is_f64, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!x && n > 0) return WL_EINVAL_ARG;
    if (th < WL_TH_HARD || th > WL_TH_NEG) return WL_EINVAL_ARG;
    if (th <= WL_TH_STEIN && !(t >= 0)) return WL_EINVAL_ARG;           // @assert t >= 0
    if (n <= 0) return WL_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = ext_blocks(n, 4, ctx->cu_count);
    if (dtype == WL_F64) hipLaunchKernelGGL((k_threshold<double, double>), dim3(nb), dim3(EXT_THREADS), 0, st, (double *)x, n, th, t, vec_ok16(x));
    else if (t_is_f64) hipLaunchKernelGGL((k_threshold<float, double>), dim3(nb), dim3(EXT_THREADS), 0, st, (float *)x, n, th, t, vec_ok16(x));
    else hipLaunchKernelGGL((k_threshold<float, float>), dim3(nb), dim3(EXT_THREADS), 0, st, (float *)x, n, th, (float)t, vec_ok16(x));
    WL_HIP(ctx, hipGetLastError());
    ctx->last_kernel = "k_threshold";
    return WL_OK;
}

int wl_threshold_biggest(wl_ctx *ctx, int dtype, void *x, int64_t n, int64_t m, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if ((!x && n > 0) || m < 0) return WL_EINVAL_ARG;
    if (n <= 0) return WL_OK;
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32 ? biggest_impl<float>(ctx, st, (float *)x, n, m) : biggest_impl<double>(ctx, st, (double *)x, n, m);
}

int wl_median(wl_ctx *ctx, int dtype, const void *v, int64_t n, double *result, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!v || !result) return WL_EINVAL_ARG;
    if (n < 1) return WL_EDIMS;
    hipStream_t st = (hipStream_t)stream;
    if (n <= (dtype == WL_F32 ? mad_lds_max<float>() : mad_lds_max<double>()))        // (the kernel does not write v when do_mad == 0)
        return dtype == WL_F32 ? mad_small<float>(ctx, st, (float *)const_cast<void *>(v), n, 0, result)
                               : mad_small<double>(ctx, st, (double *)const_cast<void *>(v), n, 0, result);
    return dtype == WL_F32 ? median_impl<float>(ctx, st, (const float *)v, n, result, (float *)nullptr)
                           : median_impl<double>(ctx, st, (const double *)v, n, result, (double *)nullptr);
}

int wl_mad(wl_ctx *ctx, int dtype, void *y, int64_t n, double *result, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!y || !result) return WL_EINVAL_ARG;
    if (n < 1) return WL_EDIMS;
    rc = ensure_aux(ctx);
    if (rc != WL_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (n <= (dtype == WL_F32 ? mad_lds_max<float>() : mad_lds_max<double>()))
        return dtype == WL_F32 ? mad_small<float>(ctx, st, (float *)y, n, 1, result) : mad_small<double>(ctx, st, (double *)y, n, 1, result);
    void *mdev = (char *)ctx->aux + 4096;          // the first median, in the element type
    const unsigned nb = ext_blocks(n, 4, ctx->cu_count);
    if (dtype == WL_F32) {
        rc = median_impl<float>(ctx, st, (const float *)y, n, nullptr, (float *)mdev);
        if (rc != WL_OK) return rc;
        hipLaunchKernelGGL((k_absdev<float>), dim3(nb), dim3(EXT_THREADS), 0, st, (float *)y, n, (const float *)mdev, vec_ok16(y));
        return median_impl<float>(ctx, st, (const float *)y, n, result, (float *)nullptr);
    }
    rc = median_impl<double>(ctx, st, (const double *)y, n, nullptr, (double *)mdev);
    if (rc != WL_OK) return rc;
    hipLaunchKernelGGL((k_absdev<double>), dim3(nb), dim3(EXT_THREADS), 0, st, (double *)y, n, (const double *)mdev, vec_ok16(y));
    return median_impl<double>(ctx, st, (const double *)y, n, result, (double *)nullptr);
}

int wl_denoise_ti_filter(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims, const double *qmf, int flen,
                         int L, int th, double t_unit, const int64_t *nspin, double sigma_host, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!y || !x || !dims || !qmf || !nspin) return WL_EINVAL_ARG;
    if (ndims < 1 || ndims > 2) return WL_EDIMS;
    if (flen < 2 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    // threshold!(xt, dnt.th, sigma*t) (denoising.jl:58) has methods for Hard / Soft / Semisoft / Stein only (threshold_main.jl:21-80)
    if (th < WL_TH_HARD || th > WL_TH_STEIN) return WL_EINVAL_ARG;
    // a custom estimator's value: NaN trips the reference's `@assert t >= 0` (threshold_main.jl:24) and so does Inf * 0; +Inf
    // alone passes it -- every coefficient is thresholded -- and is accepted here as well (negative = "estimate on the device")
    if (sigma_host != sigma_host || (sigma_host >= 0 && !(sigma_host * t_unit >= 0))) return WL_EINVAL_ARG;
    for (int d = 0; d < ndims; ++d)
        if (dims[d] < 1 || nspin[d] < 1) return WL_EDIMS;
    if (ndims == 2 && dims[0] != dims[1]) return WL_EINVAL_CUBE;            // iscube(x) (denoising.jl:29)
    if (ndims == 2 && dims[1] > 65535) return WL_EINVAL_SIZE;               // (one grid row per column in the shift kernels)
    if (L < 0) return WL_EINVAL_L;
    for (int d = 0; d < ndims; ++d)
        if (L >= 62 || (dims[d] % ((int64_t)1 << L)) != 0) return WL_EINVAL_SIZE;
    if (y == x) return WL_EALIAS;
    if (th <= WL_TH_STEIN && !(t_unit >= 0)) return WL_EINVAL_ARG;
    hipStream_t st = (hipStream_t)stream;
    rc = dtype == WL_F32 ? denoise_ti_impl<float>(ctx, st, (float *)y, (const float *)x, ndims, dims, qmf, flen, L, th, t_unit, nspin, sigma_host)
                         : denoise_ti_impl<double>(ctx, st, (double *)y, (const double *)x, ndims, dims, qmf, flen, L, th, t_unit, nspin, sigma_host);
    if (rc == WL_OK) ctx->last_kernel = "denoise_ti_batch";
    return rc;
}

int wl_denoise_ti_lifting(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims,
                          int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef, const int32_t *step_shift,
                          const double *coefs_flat, double norm1, double norm2,
                          int L, int th, double t_unit, const int64_t *nspin, double sigma_host, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!y || !x || !dims || !nspin) return WL_EINVAL_ARG;
    if (ndims < 1 || ndims > 2) return WL_EDIMS;
    if (th < WL_TH_HARD || th > WL_TH_STEIN) return WL_EINVAL_ARG;
    if (sigma_host != sigma_host || (sigma_host >= 0 && !(sigma_host * t_unit >= 0))) return WL_EINVAL_ARG;
    for (int d = 0; d < ndims; ++d)
        if (dims[d] < 1 || nspin[d] < 1) return WL_EDIMS;
    if (ndims == 2 && dims[0] != dims[1]) return WL_EINVAL_CUBE;
    if (ndims == 2 && dims[1] > 65535) return WL_EINVAL_SIZE;
    if (L < 0) return WL_EINVAL_L;
    for (int d = 0; d < ndims; ++d)
        if (L >= 62 || (dims[d] % ((int64_t)1 << L)) != 0) return WL_EINVAL_SIZE;
    if (y == x) return WL_EALIAS;
    if (!(t_unit >= 0)) return WL_EINVAL_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WL_F32) {
        LiftScheme<float> f, i;
        rc = wl_make_scheme<float>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, 1, f);
        if (rc == WL_OK) rc = wl_make_scheme<float>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, 0, i);
        if (rc == WL_OK) rc = denoise_ti_lifting_impl<float>(ctx, st, (float *)y, (const float *)x, ndims, dims, f, i, L, th, t_unit, nspin, sigma_host);
    } else {
        LiftScheme<double> f, i;
        rc = wl_make_scheme<double>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, 1, f);
        if (rc == WL_OK) rc = wl_make_scheme<double>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, 0, i);
        if (rc == WL_OK) rc = denoise_ti_lifting_impl<double>(ctx, st, (double *)y, (const double *)x, ndims, dims, f, i, L, th, t_unit, nspin, sigma_host);
    }
    if (rc == WL_OK) ctx->last_kernel = "denoise_ti_lifting";
    return rc;
}

int wl_circshift(wl_ctx *ctx, int dtype, void *b, const void *a, int ndims, const int64_t *dims, const int64_t *shift, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!b || !a || !dims || !shift) return WL_EINVAL_ARG;
    if (ndims < 1 || ndims > 3) return WL_EDIMS;
    if (b == a) return WL_EALIAS;
    Shift3 p = {{1, 1, 1}, {0, 0, 0}};
    int64_t n = 1;
    for (int k = 0; k < ndims; ++k) {
        if (dims[k] < 1) return (dims[k] == 0) ? WL_OK : WL_EDIMS;
        p.d[k] = dims[k];
        p.s[k] = ((shift[k] % dims[k]) + dims[k]) % dims[k];
        n *= dims[k];
    }
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = ext_blocks(n, 4, ctx->cu_count);
    if (dtype == WL_F32) hipLaunchKernelGGL((k_circshift<float>), dim3(nb), dim3(EXT_THREADS), 0, st, (float *)b, (const float *)a, n, p);
    else hipLaunchKernelGGL((k_circshift<double>), dim3(nb), dim3(EXT_THREADS), 0, st, (double *)b, (const double *)a, n, p);
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

int wl_arrayadd(wl_ctx *ctx, int dtype, void *y, const void *z, int64_t n, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if ((!y || !z) && n > 0) return WL_EINVAL_ARG;
    if (n <= 0) return WL_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = ext_blocks(n, 4, ctx->cu_count);
    if (dtype == WL_F32) hipLaunchKernelGGL((k_arrayadd<float>), dim3(nb), dim3(EXT_THREADS), 0, st, (float *)y, (const float *)z, n, vec_ok16(y) & vec_ok16(z));
    else hipLaunchKernelGGL((k_arrayadd<double>), dim3(nb), dim3(EXT_THREADS), 0, st, (double *)y, (const double *)z, n, vec_ok16(y) & vec_ok16(z));
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

int wl_rmul(wl_ctx *ctx, int dtype, void *y, int64_t n, double s, void *stream)
{
    int rc = ext_enter(ctx, dtype);
    if (rc != WL_OK) return rc;
    WL_SCOPE(ctx);
    if (!y && n > 0) return WL_EINVAL_ARG;
    if (n <= 0) return WL_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = ext_blocks(n, 4, ctx->cu_count);
    if (dtype == WL_F32) hipLaunchKernelGGL((k_rmul<float>), dim3(nb), dim3(EXT_THREADS), 0, st, (float *)y, n, s, vec_ok16(y));
    else hipLaunchKernelGGL((k_rmul<double>), dim3(nb), dim3(EXT_THREADS), 0, st, (double *)y, n, s, vec_ok16(y));
    WL_HIP(ctx, hipGetLastError());
    return WL_OK;
}

}  // extern "C"
