// wl_gtile.hip -- one 2-D filter-bank level (forward or inverse) of a block of ANY even extents, Float32 / Float64, F <= 10:
// the fast path of the shapes the streaming / tile / tail kernels decline (rows not a multiple of 8, columns not a multiple
// of 16, extents that are not powers of two -- image sizes such as 1080 x 1920 and their 540 x 960, 270 x 480 levels).
//
// Before this kernel those levels ran as two one-thread-per-output passes (wl_generic.hip): ~19 us each whatever the size
// (a chain of dependent scalar-tap loads, emulated 64-bit index divisions and global loads on a lone wave), two per level.
// Here a workgroup owns a 32 x 32 piece of every sub-band (64 x 64 samples of the block): the piece with its halo of F-2
// samples on every side is staged to LDS (periodic wrap by compare-and-subtract, all loads in flight), both passes run
// LDS -> LDS with compile-time taps, edges are clipped at the store.  One launch per level, ~5 us for the small levels and
// bandwidth-bound (with a 1.3-1.6x read amplification served by L2) for big odd-sized ones.
// Arithmetic: the closed forms of wl_internal.h in the reference's order, no FMA -- bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <typename T, int F>
struct GTileArgs {
    const T *src; int64_t lds;      // fw: block M x N                       inv: coefficient array
    T *y; int64_t ldy;              // fw: coefficient array                 inv: result block M x N
    T *ll; int64_t ldll;            // fw: approximation destination (hm x hn) or nullptr (-> y)
                                    // inv: approximation source (hm x hn) or nullptr (-> the corner of src)
    int M, N;                       // block extents (even)
    TapsF<T, F> tp;
};

__device__ __forceinline__ int gt_wrap(int i, int n)
{
    while (i < 0) i += n;
    while (i >= n) i -= n;
    return i;
}

// (s, d) of a pair from its 2F-2 window values xv[e] = x[2k - (F-2) + e]  (F = 2: the pair itself)
template <typename T, int F>
__device__ __forceinline__ void gt_window_sd(const T (&xv)[2 * F - 2 > 0 ? 2 * F - 2 : 2], const TapsF<T, F> &tp, T &s, T &d)
{
    s = tp.h[0] * xv[F - 2];
#pragma unroll
    for (int m = 1; m < F; ++m) s = s + tp.h[m] * xv[F - 2 + m];
    d = tp.g[F - 1] * xv[0];
#pragma unroll
    for (int m = F - 2; m >= 0; --m) d = d + tp.g[m] * xv[F - 1 - m];
}

template <typename T, int F>
__global__ void __launch_bounds__(512) k_fwd2d_gtile(GTileArgs<T, F> a)
{
    constexpr int NW = (F == 2) ? 2 : 2 * F - 2, H = F - 2;
    constexpr int E = 64 + 2 * H;                // staged samples per dimension
    constexpr int LDX = E | 1, LDT = E | 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *X = reinterpret_cast<T *>(smem_raw);      // [E rows x E cols], column j at j*LDX
    T *Tt = X + LDX * E;                         // [E rows x (32 s-cols | 32 d-cols)]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int M = a.M, N = a.N, hm = M >> 1, hn = N >> 1;
    const int r0h = blockIdx.x * 32, c0h = blockIdx.y * 32;
    const int R0 = 2 * r0h - H, C0 = 2 * c0h - H;
    for (int it = tid; it < E * E; it += nthr) {
        const int i = it % E, j = it / E;
        const int gr = gt_wrap(R0 + i, M), gc = gt_wrap(C0 + j, N);
        X[i + j * LDX] = a.src[gr + (int64_t)gc * a.lds];
    }
    lds_barrier_vm();
    // dim 2: row i, column pair kc -> Tt[i][kc] = s, Tt[i][32 + kc] = d
    for (int it = tid; it < E * 32; it += nthr) {
        const int i = it % E, kc = it / E;
        if (c0h + kc < hn) {
            T xv[NW];
#pragma unroll
            for (int e = 0; e < NW; ++e) xv[e] = X[i + (2 * kc + e) * LDX];
            T s, d;
            gt_window_sd<T, F>(xv, a.tp, s, d);
            Tt[i + kc * LDT] = s;
            Tt[i + (32 + kc) * LDT] = d;
        }
    }
    lds_barrier();
    // dim 1: column c of Tt (s-cols then d-cols), row pair kr -> the four sub-bands
    T *const lld = a.ll ? a.ll : a.y;
    const int64_t ldl = a.ll ? a.ldll : a.ldy;
    for (int it = tid; it < 32 * 64; it += nthr) {
        const int kr = it & 31, c = it >> 5;
        const int kc = c & 31;
        if (r0h + kr < hm && c0h + kc < hn) {
            T xv[NW];
#pragma unroll
            for (int e = 0; e < NW; ++e) xv[e] = Tt[2 * kr + e + c * LDT];
            T s, d;
            gt_window_sd<T, F>(xv, a.tp, s, d);
            const int gk = r0h + kr;
            if (c < 32) {                                        // s along dim 2: ss (approximation) and ds
                lld[gk + (int64_t)(c0h + kc) * ldl] = s;
                a.y[hm + gk + (int64_t)(c0h + kc) * a.ldy] = d;
            } else {                                             // d along dim 2: sd and dd
                a.y[gk + (int64_t)(hn + c0h + kc) * a.ldy] = s;
                a.y[hm + gk + (int64_t)(hn + c0h + kc) * a.ldy] = d;
            }
        }
    }
}

// inverse: (x[2p], x[2p+1]) from sw[q] = s[p - SH + q], dw[q] = d[p + q] (window_inv, wl_dev.h)
template <typename T, int F>
__global__ void __launch_bounds__(512) k_inv2d_gtile(GTileArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2, E = 32 + SH, LD = E | 1, LT = 65;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *SS = reinterpret_cast<T *>(smem_raw);     // [E s-rows x E s-cols]
    T *DS = SS + LD * E, *SD = DS + LD * E, *DD = SD + LD * E;
    T *Tt = DD + LD * E;                         // [64 rows x (E s-cols | E d-cols)]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int M = a.M, N = a.N, hm = M >> 1, hn = N >> 1;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, Pr = r0 >> 1, Pc = c0 >> 1;
    const T *lls = a.ll ? a.ll : a.src;
    const int64_t ldl = a.ll ? a.ldll : a.lds;
    for (int it = tid; it < E * E; it += nthr) {
        const int ai = it % E, bi = it / E;
        const int rs = gt_wrap(Pr - SH + ai, hm), rd = gt_wrap(Pr + ai, hm), cs = gt_wrap(Pc - SH + bi, hn), cd = gt_wrap(Pc + bi, hn);
        SS[ai + bi * LD] = lls[rs + (int64_t)cs * ldl];
        DS[ai + bi * LD] = a.src[hm + rd + (int64_t)cs * a.lds];
        SD[ai + bi * LD] = a.src[rs + (int64_t)(hn + cd) * a.lds];
        DD[ai + bi * LD] = a.src[hm + rd + (int64_t)(hn + cd) * a.lds];
    }
    lds_barrier_vm();
    // dim 1: columns b of (s-cols | d-cols), pairs p -> Tt rows 2p, 2p+1
    for (int it = tid; it < 32 * 2 * E; it += nthr) {
        const int p = it & 31, b = it >> 5;
        const T *sp = (b < E) ? SS + b * LD : SD + (b - E) * LD;
        const T *dp = (b < E) ? DS + b * LD : DD + (b - E) * LD;
        T sw[SH + 1], dw[SH + 1];
#pragma unroll
        for (int q = 0; q <= SH; ++q) { sw[q] = sp[p + q]; dw[q] = dp[p + q]; }
        T xe, xo;
        window_inv<T, F>(sw, dw, a.tp, xe, xo);
        Tt[2 * p + b * LT] = xe;
        Tt[2 * p + 1 + b * LT] = xo;
    }
    lds_barrier();
    // dim 2: rows i, column pairs p -> the result tile (clipped at the block's edges)
    for (int it = tid; it < 64 * 32; it += nthr) {
        const int i = it & 63, p = it >> 6;
        if (r0 + i < M && c0 + 2 * p < N) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int q = 0; q <= SH; ++q) { sw[q] = Tt[i + (p + q) * LT]; dw[q] = Tt[i + (E + p + q) * LT]; }
            T xe, xo;
            window_inv<T, F>(sw, dw, a.tp, xe, xo);
            a.y[r0 + i + (int64_t)(c0 + 2 * p) * a.ldy] = xe;
            a.y[r0 + i + (int64_t)(c0 + 2 * p + 1) * a.ldy] = xo;
        }
    }
}

bool gtile_ok(int F, int64_t M, int64_t N)
{
    if (F < 2 || F > 10 || (F & 1)) return false;
    return M >= 2 && N >= 2 && (M % 2) == 0 && (N % 2) == 0 && M < ((int64_t)1 << 30) && N < ((int64_t)1 << 30) &&
           (N + 63) / 64 <= 65535;
}

template <typename T, int F, int FW>
static hipError_t launch_gtile_f(hipStream_t st, const Taps<T> &taps, const T *src, int64_t lds, T *y, int64_t ldy, T *ll, int64_t ldll,
                                 int M, int N)
{
    GTileArgs<T, F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.M = M; a.N = N;
    a.tp = shrink<T, F>(taps);
    size_t elems;
    if (FW) {
        constexpr int E = 64 + 2 * (F - 2), LDX = E | 1;
        elems = (size_t)LDX * E + (size_t)LDX * 64 + 16;
    } else {
        constexpr int SH = (F - 2) / 2, E = 32 + SH, LD = E | 1;
        elems = (size_t)4 * LD * E + (size_t)65 * 2 * E + 16;
    }
    const size_t shmem = elems * sizeof(T);
    const void *fn = FW ? reinterpret_cast<const void *>(&k_fwd2d_gtile<T, F>) : reinterpret_cast<const void *>(&k_inv2d_gtile<T, F>);
    if (shmem > 48 * 1024) {
        static thread_local const void *done_fn[8];
        static thread_local int done_dev[8];
        static thread_local int ndone = 0;
        int dev = 0;
        (void)hipGetDevice(&dev);
        bool done = false;
        for (int i = 0; i < ndone; ++i) done = done || (done_fn[i] == fn && done_dev[i] == dev);
        if (!done) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (ndone < 8) { done_fn[ndone] = fn; done_dev[ndone] = dev; ++ndone; }
        }
    }
    const dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64));
    if (FW) hipLaunchKernelGGL((k_fwd2d_gtile<T, F>), grid, dim3(512), shmem, st, a);
    else hipLaunchKernelGGL((k_inv2d_gtile<T, F>), grid, dim3(512), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t gtile_launch(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t lds, T *y, int64_t ldy, T *ll, int64_t ldll,
                        int M, int N)
{
#define WL_GT(FF_)                                                                                                   \
    case FF_: return fw ? launch_gtile_f<T, FF_, 1>(st, taps, src, lds, y, ldy, ll, ldll, M, N)                      \
                        : launch_gtile_f<T, FF_, 0>(st, taps, src, lds, y, ldy, ll, ldll, M, N);
    switch (taps.F) {
        WL_GT(2) WL_GT(4) WL_GT(6) WL_GT(8) WL_GT(10)
    default: return hipErrorInvalidValue;
    }
#undef WL_GT
}
template hipError_t gtile_launch<float>(hipStream_t, const Taps<float> &, int, const float *, int64_t, float *, int64_t, float *, int64_t, int, int);
template hipError_t gtile_launch<double>(hipStream_t, const Taps<double> &, int, const double *, int64_t, double *, int64_t, double *, int64_t, int,
                                         int);

}  // namespace wl
