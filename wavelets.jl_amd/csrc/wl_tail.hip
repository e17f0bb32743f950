// wl_tail.hip -- the deep end of a forward filter-bank transform: ALL remaining levels of a small power-of-two block
// (2-D: m0 x m1 <= 4096 elements; 1-D: a line of <= 4096 samples, one workgroup per line for batches) in one launch.
//
// At these sizes a level is pure latency: a 64 x 64 block is 16 KiB and the 32 x 32 ... 2 x 2 levels below it hold less
// data than one wave's registers, yet every level is two dependent passes (reference transforms_filter.jl:161-172:
// dim 2 then dim 1) with an all-to-all between them.  k_tail_fwd (wl_fwd.hip, any size/filter) spends ~0.7 us per
// pass on index arithmetic with run-time wrap loops; here the block extents are powers of two, so
//   * periodic wrap is a bit mask, every LDS address of a window is known before the first read is issued
//     (2F-2 independent ds_read per output pair, issued back to back);
//   * the dim-1 pass (contiguous lines) reads its window as F-1 aligned 8-byte pairs;
//   * F is a compile-time constant, the taps live in SGPRs;
//   * barriers order LDS traffic only (detail coefficients stream to HBM while the next pass runs).
// Same closed forms and summation order as everywhere else (wl_internal.h): bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

WL_STAMP_DECL(tail)

namespace wl {

template <typename T, int F>
struct Tail2Args {
    const T *src; int64_t s1;       // source block: element (i, j) at src[i + j*s1]
    T *y; int64_t ldy;              // destination array (same origin as the block)
    int64_t src_item, y_item;       // per-blockIdx.x offsets (batched lines)
    int lg0, lg1;                   // block extents 2^lg0 x 2^lg1 (lg1 = 0 with nt = 1: a line)
    int nt;                         // 1: transform dim 1 only; 2: both dims
    int nlev;
    TapsF<T, F> tp;
};

// (s, d) of pair k from the 2F-2 window values xv[e] = x[(2k - (F-2) + e) mod n]
template <typename T, int F>
__device__ __forceinline__ void window_sd(const T (&xv)[2 * F - 2 > 0 ? 2 * F - 2 : 2], const TapsF<T, F> &tp, T &s, T &d)
{
    s = tp.h[0] * xv[F - 2];
#pragma unroll
    for (int m = 1; m < F; ++m) s = s + tp.h[m] * xv[F - 2 + m];
    d = tp.g[F - 1] * xv[0];
#pragma unroll
    for (int m = F - 2; m >= 0; --m) d = d + tp.g[m] * xv[F - 1 - m];
}

template <typename T, int F>
__global__ void __launch_bounds__(1024) k_tail2_fwd(Tail2Args<T, F> a)
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int NW = (F == 2) ? 2 : 2 * F - 2;          // window length (F = 2: the pair itself)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const bool multi = nthr > 64;
    const int m0 = 1 << a.lg0, m1 = 1 << a.lg1;
    const int ld = (m1 > 1) ? (m0 + 2) : m0;              // even: the 8-byte window reads stay aligned
    T *A = reinterpret_cast<T *>(smem_raw);
    T *B = A + (size_t)ld * m1 + 8;
    const T *src = a.src + (int64_t)blockIdx.x * a.src_item;
    T *y = a.y + (int64_t)blockIdx.x * a.y_item;
    if (tid == 0) WL_STAMP_AT(tail, 0, 0);

    // ---- stage the block (16-byte loads when the layout allows, all of a thread's loads in flight together) ----
    {
        constexpr int VW = 16 / sizeof(T);
        const int total = m0 * m1;
        const bool vec_ok = (m0 % VW) == 0 && (a.s1 % VW) == 0 && (a.src_item % VW) == 0 &&
                            ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0);
        if (vec_ok) {
            const int totalv = total / VW, lgv = a.lg0 - (VW == 4 ? 2 : 1);
            for (int idx = tid; idx < totalv; idx += 4 * nthr) {
                T v[4][VW];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int id = idx + r * nthr;
                    if (id < totalv) vload<T, VW>(src + (id & ((1 << lgv) - 1)) * VW + (int64_t)(id >> lgv) * a.s1, v[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int id = idx + r * nthr;
                    if (id < totalv) {
                        T *dst = A + (id & ((1 << lgv) - 1)) * VW + (id >> lgv) * ld;      // ld even: 8-byte aligned at least
#pragma unroll
                        for (int e = 0; e < VW; e += 2) *reinterpret_cast<T2 *>(dst + e) = T2{v[r][e], v[r][e + 1]};
                    }
                }
            }
        } else {
            for (int idx = tid; idx < total; idx += nthr) A[(idx & (m0 - 1)) + (idx >> a.lg0) * ld] = src[(idx & (m0 - 1)) + (int64_t)(idx >> a.lg0) * a.s1];
        }
    }
    if (multi) lds_barrier_vm(); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (tid == 0) WL_STAMP_AT(tail, 0, 1);

    for (int lev = 0; lev < a.nlev; ++lev) {
        const bool last = (lev == a.nlev - 1);
        if (tid == 0 && lev >= 1 && lev <= 5) WL_STAMP_AT(tail, 0, 1 + lev);
        const int lgn0 = a.lg0 - lev, n0 = 1 << lgn0, h0 = n0 >> 1;
        if (a.nt == 2) {
            const int lgn1 = a.lg1 - lev, n1 = 1 << lgn1, h1 = n1 >> 1;
            // ---- dim-2 pass: A (n0 x n1) -> B, [s ; d] along j; lanes along i ----
            for (int idx = tid; idx < (h1 << lgn0); idx += nthr) {
                const int i = idx & (n0 - 1), k = idx >> lgn0;
                const T *p = A + i;
                T xv[NW];
#pragma unroll
                for (int e = 0; e < NW; ++e) xv[e] = p[((2 * k - (F - 2) + e) & (n1 - 1)) * ld];
                T s, d;
                window_sd<T, F>(xv, a.tp, s, d);
                B[i + k * ld] = s;
                B[i + (h1 + k) * ld] = d;
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- dim-1 pass: B -> details to y, approximation to A (h0 x h1) or y; lanes along k ----
            for (int idx = tid; idx < (n1 << (lgn0 - 1)); idx += nthr) {
                const int k = idx & (h0 - 1), j = idx >> (lgn0 - 1);
                const T *p = B + j * ld;
                T xv[NW];
#pragma unroll
                for (int e = 0; e < NW / 2; ++e) {
                    const T2 v = *reinterpret_cast<const T2 *>(p + ((2 * k - (F - 2) + 2 * e) & (n0 - 1)));
                    xv[2 * e] = v.x; xv[2 * e + 1] = v.y;
                }
                T s, d;
                window_sd<T, F>(xv, a.tp, s, d);
                T *yc = y + (int64_t)j * a.ldy;
                yc[h0 + k] = d;
                if (j < h1 && !last) A[k + j * ld] = s;
                else yc[k] = s;
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            for (int k = tid; k < h0; k += nthr) {
                T xv[NW];
#pragma unroll
                for (int e = 0; e < NW / 2; ++e) {
                    const T2 v = *reinterpret_cast<const T2 *>(A + ((2 * k - (F - 2) + 2 * e) & (n0 - 1)));
                    xv[2 * e] = v.x; xv[2 * e + 1] = v.y;
                }
                T s, d;
                window_sd<T, F>(xv, a.tp, s, d);
                y[h0 + k] = d;
                if (last) y[k] = s;
                else B[k] = s;
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            T *t = A; A = B; B = t;
        }
    }
    if (tid == 0) WL_STAMP_AT(tail, 0, 7);
}

template <typename T>
bool tail2_ok(int F, int nt, int64_t m0, int64_t m1, int nlev, int fmax)
{
    // fmax = 10: the default; 20: the forward and (round 4) the inverse tail are also instantiated for 12..20 taps (wl_fwd2d_long.hip's
    // filters: one workgroup finishes a 64 x 64 block instead of two chip-wide launches per level down to 16 x 16)
    if (F < 2 || F > fmax || F > 20 || (F & 1)) return false;
    auto pow2 = [](int64_t v) { return v >= 2 && (v & (v - 1)) == 0; };
    if (!pow2(m0) || (nt == 2 && !pow2(m1))) return false;
    if (nt == 1 && m1 != 1) return false;
    const int64_t cap = 4096;                                       // elements: 16 KiB of Float32, 32 KiB of Float64
    if (m0 * m1 > cap) return false;
    // every level needs both extents >= 2
    int lg0 = 0, lg1 = 0;
    while (((int64_t)1 << lg0) < m0) ++lg0;
    while (((int64_t)1 << lg1) < m1) ++lg1;
    if (nlev < 1 || nlev > lg0 || (nt == 2 && nlev > lg1)) return false;
    return true;
}

// blocks of Float64 need more than the default 64 KiB of dynamic LDS: raise the limit once per (kernel, device)
static hipError_t tail2_lds_attr(const void *fn, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    static thread_local const void *done_fn[16];
    static thread_local int done_dev[16];
    static thread_local int ndone = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < ndone; ++i)
        if (done_fn[i] == fn && done_dev[i] == dev) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess && ndone < 16) { done_fn[ndone] = fn; done_dev[ndone] = dev; ++ndone; }
    return e;
}

template <typename T, int F>
static hipError_t launch_tail2_f(hipStream_t st, const Taps<T> &taps, const T *src, int64_t s1, T *y, int64_t ldy,
                                 int64_t src_item, int64_t y_item, int nitems, int m0, int m1, int nt, int nlev)
{
    Tail2Args<T, F> a;
    a.src = src; a.s1 = s1; a.y = y; a.ldy = ldy; a.src_item = src_item; a.y_item = y_item; a.nt = nt; a.nlev = nlev;
    a.lg0 = 0; a.lg1 = 0;
    while ((1 << a.lg0) < m0) ++a.lg0;
    while ((1 << a.lg1) < m1) ++a.lg1;
    a.tp = shrink<T, F>(taps);
    const int ld = (m1 > 1) ? (m0 + 2) : m0;
    const size_t shmem = (2 * ((size_t)ld * m1 + 8)) * sizeof(T);
    const int pairs = m0 * m1 / 2;                                   // output pairs of the first pass
    // (r06: 12 waves for the 4096-element block -- 64^2 six levels 8.03 us with 512 threads, 7.50 with 768, 7.69 with 1024, 10.1 with 256)
    int threads = pairs >= 2048 ? 768 : (pairs >= 1024 ? 512 : (pairs >= 256 ? 256 : (pairs >= 128 ? 128 : 64)));
    const int to = (int)opt("WL_TAIL2_THREADS", 0);
    if (to >= 64 && to <= 1024 && (to % 64) == 0) threads = to;
    hipError_t ea = tail2_lds_attr(reinterpret_cast<const void *>(&k_tail2_fwd<T, F>), shmem);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((k_tail2_fwd<T, F>), dim3((unsigned)nitems), dim3(threads), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_tail2(hipStream_t st, const Taps<T> &taps, const T *src, int64_t s1, T *y, int64_t ldy,
                        int64_t src_item, int64_t y_item, int nitems, int m0, int m1, int nt, int nlev)
{
    switch (taps.F) {
    case 2: return launch_tail2_f<T, 2>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 4: return launch_tail2_f<T, 4>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 6: return launch_tail2_f<T, 6>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 8: return launch_tail2_f<T, 8>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 10: return launch_tail2_f<T, 10>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 12: return launch_tail2_f<T, 12>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 14: return launch_tail2_f<T, 14>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 16: return launch_tail2_f<T, 16>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 18: return launch_tail2_f<T, 18>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    case 20: return launch_tail2_f<T, 20>(st, taps, src, s1, y, ldy, src_item, y_item, nitems, m0, m1, nt, nlev);
    default: return hipErrorInvalidValue;
    }
}

// ---------------------------------------------------------------------------------------------------
// The inverse: the deepest `nlev` levels of a reconstruction whose last output here is a power-of-two block n0 x n1
// (<= 4096 f32 elements) / a line of n0 samples, in one launch.  The coefficient corner is staged to LDS once; a 2-D level
// is the dim-1 reconstruction A -> B (lanes along the output pairs of a column, 8-byte stores) followed by the dim-2
// reconstruction B -> A (lanes along the rows), in place over the quadrants it has just consumed (reference
// transforms_filter.jl:173-186: columns first, then rows).  Same masks / compile-time taps as k_tail2_fwd.
template <typename T, int F>
struct Tail2InvArgs {
    const T *x; int64_t ldx; int64_t x_item;
    T *out; int64_t ldo; int64_t out_item;
    int lg0, lg1;                   // OUTPUT extents 2^lg0 x 2^lg1 of the last level done here (lg1 = 0 with nt = 1)
    int nt, nlev;
    TapsF<T, F> tp;
};

template <typename T, int F>
__global__ void __launch_bounds__(512) k_tail2_inv(Tail2InvArgs<T, F> a)
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int SH = (F - 2) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const bool multi = nthr > 64;
    const int n0 = 1 << a.lg0, n1 = 1 << a.lg1;
    const int ld = (n1 > 1) ? (n0 + 2) : n0;
    T *A = reinterpret_cast<T *>(smem_raw);
    T *B = A + (size_t)ld * n1 + 8;
    const T *x = a.x + (int64_t)blockIdx.x * a.x_item;
    T *out = a.out + (int64_t)blockIdx.x * a.out_item;
    {
        constexpr int VW = 16 / sizeof(T);
        const int total = n0 * n1;
        const bool vec_ok = (n0 % VW) == 0 && (a.ldx % VW) == 0 && (a.x_item % VW) == 0 && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
        if (vec_ok) {
            const int totalv = total / VW, lgv = a.lg0 - (VW == 4 ? 2 : 1);
            for (int idx = tid; idx < totalv; idx += 4 * nthr) {
                T v[4][VW];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int id = idx + r * nthr;
                    if (id < totalv) vload<T, VW>(x + (id & ((1 << lgv) - 1)) * VW + (int64_t)(id >> lgv) * a.ldx, v[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int id = idx + r * nthr;
                    if (id < totalv) {
                        T *dst = A + (id & ((1 << lgv) - 1)) * VW + (id >> lgv) * ld;
#pragma unroll
                        for (int e = 0; e < VW; e += 2) *reinterpret_cast<T2 *>(dst + e) = T2{v[r][e], v[r][e + 1]};
                    }
                }
            }
        } else {
            for (int idx = tid; idx < total; idx += nthr) A[(idx & (n0 - 1)) + (idx >> a.lg0) * ld] = x[(idx & (n0 - 1)) + (int64_t)(idx >> a.lg0) * a.ldx];
        }
    }
    if (multi) lds_barrier_vm(); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    if (a.nt == 2) {
        for (int t = 0; t < a.nlev; ++t) {
            const bool last = (t == a.nlev - 1);
            const int lgo0 = a.lg0 - (a.nlev - 1 - t), lgo1 = a.lg1 - (a.nlev - 1 - t);
            const int o0 = 1 << lgo0, o1 = 1 << lgo1, h0 = o0 >> 1, h1 = o1 >> 1;
            // ---- dim-1 reconstruction: A -> B; lanes along the output pairs p of column j ----
            for (int idx = tid; idx < (o1 << (lgo0 - 1)); idx += nthr) {
                const int p = idx & (h0 - 1), j = idx >> (lgo0 - 1);
                const T *sp = A + j * ld, *dp = sp + h0;
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sp[(p - SH + q) & (h0 - 1)]; dw[q] = dp[(p + q) & (h0 - 1)]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                *reinterpret_cast<T2 *>(B + 2 * p + j * ld) = T2{xe, xo};
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // ---- dim-2 reconstruction: B -> A corner (or the result array); lanes along the rows ----
            for (int idx = tid; idx < (h1 << lgo0); idx += nthr) {
                const int i = idx & (o0 - 1), p = idx >> lgo0;
                const T *sp = B + i, *dp = sp + h1 * ld;
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sp[((p - SH + q) & (h1 - 1)) * ld]; dw[q] = dp[((p + q) & (h1 - 1)) * ld]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                if (last) {
                    out[i + (int64_t)(2 * p) * a.ldo] = xe;
                    out[i + (int64_t)(2 * p + 1) * a.ldo] = xo;
                } else {
                    A[i + (2 * p) * ld] = xe;
                    A[i + (2 * p + 1) * ld] = xo;
                }
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    } else {
        // a line: coefficients stay in A, reconstructions ping-pong between B and B + n0/2 (every non-final one is <= n0/2 long)
        const T *cur = A;
        T *P0 = B, *P1 = B + (n0 >> 1) + 8;
        for (int t = 0; t < a.nlev; ++t) {
            const bool last = (t == a.nlev - 1);
            const int lgo = a.lg0 - (a.nlev - 1 - t), o = 1 << lgo, h = o >> 1;
            const T *sp = cur, *dp = A + h;
            T *dst = (t & 1) ? P1 : P0;
            for (int p = tid; p < h; p += nthr) {
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sp[(p - SH + q) & (h - 1)]; dw[q] = dp[(p + q) & (h - 1)]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                if (last) *reinterpret_cast<T2 *>(out + 2 * p) = T2{xe, xo};
                else *reinterpret_cast<T2 *>(dst + 2 * p) = T2{xe, xo};
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cur = dst;
        }
    }
}

template <typename T>
bool tail2_inv_ok(int F, int nt, int64_t n0, int64_t n1, int nlev, const T *out, int64_t out_item)
{
    if (!tail2_ok<T>(F, nt, n0, n1, nlev, 20)) return false;       // (12 ... 20 taps since round 4, as the forward tail)
    // the last level of a line is stored as 8-byte pairs
    if (nt == 1 && ((reinterpret_cast<uintptr_t>(out) % (2 * sizeof(T))) != 0 || (out_item % 2) != 0)) return false;
    return true;
}

template <typename T, int F>
static hipError_t launch_tail2_inv_f(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, int64_t x_item, T *out, int64_t ldo,
                                     int64_t out_item, int nitems, int n0, int n1, int nt, int nlev)
{
    Tail2InvArgs<T, F> a;
    a.x = x; a.ldx = ldx; a.x_item = x_item; a.out = out; a.ldo = ldo; a.out_item = out_item; a.nt = nt; a.nlev = nlev;
    a.lg0 = 0; a.lg1 = 0;
    while ((1 << a.lg0) < n0) ++a.lg0;
    while ((1 << a.lg1) < n1) ++a.lg1;
    a.tp = shrink<T, F>(taps);
    const int ld = (n1 > 1) ? (n0 + 2) : n0;
    const size_t shmem = (2 * ((size_t)ld * n1 + 8) + 16) * sizeof(T);
    const int pairs = n0 * n1 / 2;
    int threads = pairs >= 1024 ? 512 : (pairs >= 256 ? 256 : (pairs >= 128 ? 128 : 64));
    const int to = (int)opt("WL_TAIL2_THREADS", 0);
    if (to >= 64 && to <= 512 && (to % 64) == 0) threads = to;
    hipError_t ea = tail2_lds_attr(reinterpret_cast<const void *>(&k_tail2_inv<T, F>), shmem);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((k_tail2_inv<T, F>), dim3((unsigned)nitems), dim3(threads), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_tail2_inv(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, int64_t x_item, T *out, int64_t ldo,
                            int64_t out_item, int nitems, int n0, int n1, int nt, int nlev)
{
    switch (taps.F) {
    case 2: return launch_tail2_inv_f<T, 2>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 4: return launch_tail2_inv_f<T, 4>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 6: return launch_tail2_inv_f<T, 6>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 8: return launch_tail2_inv_f<T, 8>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 10: return launch_tail2_inv_f<T, 10>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 12: return launch_tail2_inv_f<T, 12>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 14: return launch_tail2_inv_f<T, 14>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 16: return launch_tail2_inv_f<T, 16>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 18: return launch_tail2_inv_f<T, 18>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    case 20: return launch_tail2_inv_f<T, 20>(st, taps, x, ldx, x_item, out, ldo, out_item, nitems, n0, n1, nt, nlev);
    default: return hipErrorInvalidValue;
    }
}

template bool tail2_inv_ok<float>(int, int, int64_t, int64_t, int, const float *, int64_t);
template bool tail2_inv_ok<double>(int, int, int64_t, int64_t, int, const double *, int64_t);
template hipError_t launch_tail2_inv<float>(hipStream_t, const Taps<float> &, const float *, int64_t, int64_t, float *, int64_t, int64_t, int, int,
                                            int, int, int);
template hipError_t launch_tail2_inv<double>(hipStream_t, const Taps<double> &, const double *, int64_t, int64_t, double *, int64_t, int64_t, int,
                                             int, int, int, int);
// ---------------------------------------------------------------------------------------------------
// 3-D boxes: every remaining level of a power-of-two box of <= 4096 elements in one workgroup (forward), the deepest
// levels of a reconstruction whose output is such a box (inverse).  Three dependent passes per level in the reference's
// order (forward: planes, rows, columns -- transforms_filter.jl:246-287; inverse: the reverse), LDS <-> LDS with mask wrap;
// replaces three one-thread-per-output launches per level.
template <typename T, int F>
struct Tail3Args {
    const T *src; int64_t s1, s2;   // forward: source box; inverse: coefficient array
    T *y; int64_t y1, y2;           // forward: coefficient array; inverse: result box
    int lg0, lg1, lg2;              // extents 2^lg (inverse: OUTPUT extents of the last level done here)
    int nlev;
    TapsF<T, F> tp;
};

template <typename T, int F, int FW>
__global__ void __launch_bounds__(512) k_tail3(Tail3Args<T, F> a)
{
    constexpr int NW = (F == 2) ? 2 : 2 * F - 2, SH = (F - 2) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n0 = 1 << a.lg0, n1 = 1 << a.lg1, n2 = 1 << a.lg2;
    const int ld1 = n0, ld2 = n0 * n1, total = n0 * n1 * n2;
    T *A = reinterpret_cast<T *>(smem_raw);
    T *B = A + total + 8;
    for (int idx = tid; idx < total; idx += nthr) {
        const int i = idx & (n0 - 1), j = (idx >> a.lg0) & (n1 - 1), k = idx >> (a.lg0 + a.lg1);
        A[idx] = a.src[i + (int64_t)j * a.s1 + (int64_t)k * a.s2];
    }
    lds_barrier_vm();
    // one pass along the dimension with element stride `st` (m pairs... see callers): thread = (pair p, the two other indices)
    if (FW) {
        for (int lev = 0; lev < a.nlev; ++lev) {
            const bool last = (lev == a.nlev - 1);
            const int l0 = a.lg0 - lev, l1 = a.lg1 - lev, l2 = a.lg2 - lev;
            const int m0 = 1 << l0, m1 = 1 << l1, m2 = 1 << l2, h0 = m0 >> 1, h1 = m1 >> 1, h2 = m2 >> 1;
            // dim 3: A -> B
            for (int idx = tid; idx < m0 * m1 * h2; idx += nthr) {
                const int i = idx & (m0 - 1), j = (idx >> l0) & (m1 - 1), p = idx >> (l0 + l1);
                const T *q = A + i + j * ld1;
                T xv[NW];
#pragma unroll
                for (int e = 0; e < NW; ++e) xv[e] = q[((2 * p - (F - 2) + e) & (m2 - 1)) * ld2];
                T sv, dv;
                window_sd<T, F>(xv, a.tp, sv, dv);
                B[i + j * ld1 + p * ld2] = sv;
                B[i + j * ld1 + (h2 + p) * ld2] = dv;
            }
            lds_barrier();
            // dim 2: B -> A
            for (int idx = tid; idx < m0 * h1 * m2; idx += nthr) {
                const int i = idx & (m0 - 1), p = (idx >> l0) & (h1 - 1), k = idx >> (l0 + l1 - 1);
                const T *q = B + i + k * ld2;
                T xv[NW];
#pragma unroll
                for (int e = 0; e < NW; ++e) xv[e] = q[((2 * p - (F - 2) + e) & (m1 - 1)) * ld1];
                T sv, dv;
                window_sd<T, F>(xv, a.tp, sv, dv);
                A[i + p * ld1 + k * ld2] = sv;
                A[i + (h1 + p) * ld1 + k * ld2] = dv;
            }
            lds_barrier();
            // dim 1: A -> details to y, the approximation octant to B (next level's input) or y
            for (int idx = tid; idx < h0 * m1 * m2; idx += nthr) {
                const int p = idx & (h0 - 1), j = (idx >> (l0 - 1)) & (m1 - 1), k = idx >> (l0 - 1 + l1);
                const T *q = A + j * ld1 + k * ld2;
                T xv[NW];
#pragma unroll
                for (int e = 0; e < NW; ++e) xv[e] = q[(2 * p - (F - 2) + e) & (m0 - 1)];
                T sv, dv;
                window_sd<T, F>(xv, a.tp, sv, dv);
                T *yc = a.y + (int64_t)j * a.y1 + (int64_t)k * a.y2;
                yc[h0 + p] = dv;
                if (!last && j < h1 && k < h2) B[p + j * ld1 + k * ld2] = sv;
                else yc[p] = sv;
            }
            lds_barrier();
            T *t = A; A = B; B = t;                       // the approximation octant sits in the corner of the new A
        }
    } else {
        for (int t = 0; t < a.nlev; ++t) {
            const bool last = (t == a.nlev - 1);
            const int sft = a.nlev - 1 - t;
            const int l0 = a.lg0 - sft, l1 = a.lg1 - sft, l2 = a.lg2 - sft;
            const int o0 = 1 << l0, o1 = 1 << l1, o2 = 1 << l2, h0 = o0 >> 1, h1 = o1 >> 1, h2 = o2 >> 1;
            // dim 1: A -> B
            for (int idx = tid; idx < h0 * o1 * o2; idx += nthr) {
                const int p = idx & (h0 - 1), j = (idx >> (l0 - 1)) & (o1 - 1), k = idx >> (l0 - 1 + l1);
                const T *sp = A + j * ld1 + k * ld2, *dp = sp + h0;
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sp[(p - SH + q) & (h0 - 1)]; dw[q] = dp[(p + q) & (h0 - 1)]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                B[2 * p + j * ld1 + k * ld2] = xe;
                B[2 * p + 1 + j * ld1 + k * ld2] = xo;
            }
            lds_barrier();
            // dim 2: B -> A (in place over the octants just consumed)
            for (int idx = tid; idx < o0 * h1 * o2; idx += nthr) {
                const int i = idx & (o0 - 1), p = (idx >> l0) & (h1 - 1), k = idx >> (l0 + l1 - 1);
                const T *sp = B + i + k * ld2, *dp = sp + h1 * ld1;
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sp[((p - SH + q) & (h1 - 1)) * ld1]; dw[q] = dp[((p + q) & (h1 - 1)) * ld1]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                A[i + (2 * p) * ld1 + k * ld2] = xe;
                A[i + (2 * p + 1) * ld1 + k * ld2] = xo;
            }
            lds_barrier();
            // dim 3: A -> B (or the result array)
            for (int idx = tid; idx < o0 * o1 * h2; idx += nthr) {
                const int i = idx & (o0 - 1), j = (idx >> l0) & (o1 - 1), p = idx >> (l0 + l1);
                const T *sp = A + i + j * ld1, *dp = sp + h2 * ld2;
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sp[((p - SH + q) & (h2 - 1)) * ld2]; dw[q] = dp[((p + q) & (h2 - 1)) * ld2]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                if (last) {
                    a.y[i + (int64_t)j * a.y1 + (int64_t)(2 * p) * a.y2] = xe;
                    a.y[i + (int64_t)j * a.y1 + (int64_t)(2 * p + 1) * a.y2] = xo;
                } else {
                    B[i + j * ld1 + (2 * p) * ld2] = xe;
                    B[i + j * ld1 + (2 * p + 1) * ld2] = xo;
                }
            }
            lds_barrier();
            if (!last) {                                   // the reconstruction becomes the approximation octant of the next level
                for (int idx = tid; idx < o0 * o1 * o2; idx += nthr) {
                    const int i = idx & (o0 - 1), j = (idx >> l0) & (o1 - 1), k = idx >> (l0 + l1);
                    A[i + j * ld1 + k * ld2] = B[i + j * ld1 + k * ld2];
                }
                lds_barrier();
            }
        }
    }
}

template <typename T>
bool tail3_ok(int F, int64_t n0, int64_t n1, int64_t n2, int nlev)
{
    if (F < 2 || F > 10 || (F & 1)) return false;
    auto pow2 = [](int64_t v) { return v >= 2 && (v & (v - 1)) == 0; };
    if (!pow2(n0) || !pow2(n1) || !pow2(n2) || n0 * n1 * n2 > 4096) return false;
    int lg0 = 0, lg1 = 0, lg2 = 0;
    while (((int64_t)1 << lg0) < n0) ++lg0;
    while (((int64_t)1 << lg1) < n1) ++lg1;
    while (((int64_t)1 << lg2) < n2) ++lg2;
    return nlev >= 1 && nlev <= lg0 && nlev <= lg1 && nlev <= lg2;
}

template <typename T, int F, int FW>
static hipError_t launch_tail3_f(hipStream_t st, const Taps<T> &taps, const T *src, int64_t s1, int64_t s2, T *y, int64_t y1, int64_t y2,
                                 int n0, int n1, int n2, int nlev)
{
    Tail3Args<T, F> a;
    a.src = src; a.s1 = s1; a.s2 = s2; a.y = y; a.y1 = y1; a.y2 = y2; a.nlev = nlev;
    a.lg0 = a.lg1 = a.lg2 = 0;
    while ((1 << a.lg0) < n0) ++a.lg0;
    while ((1 << a.lg1) < n1) ++a.lg1;
    while ((1 << a.lg2) < n2) ++a.lg2;
    a.tp = shrink<T, F>(taps);
    const int total = n0 * n1 * n2;
    const size_t shmem = (2 * (size_t)total + 16) * sizeof(T);
    const int threads = total >= 2048 ? 512 : (total >= 512 ? 256 : 64);
    hipError_t ea = tail2_lds_attr(reinterpret_cast<const void *>(&k_tail3<T, F, FW>), shmem);
    if (ea != hipSuccess) return ea;
    hipLaunchKernelGGL((k_tail3<T, F, FW>), dim3(1), dim3(threads), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t launch_tail3(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t s1, int64_t s2, T *y, int64_t y1, int64_t y2,
                        int n0, int n1, int n2, int nlev)
{
#define WL_T3(FF_)                                                                                                        \
    case FF_: return fw ? launch_tail3_f<T, FF_, 1>(st, taps, src, s1, s2, y, y1, y2, n0, n1, n2, nlev)                   \
                        : launch_tail3_f<T, FF_, 0>(st, taps, src, s1, s2, y, y1, y2, n0, n1, n2, nlev);
    switch (taps.F) {
        WL_T3(2) WL_T3(4) WL_T3(6) WL_T3(8) WL_T3(10)
    default: return hipErrorInvalidValue;
    }
#undef WL_T3
}
template bool tail3_ok<float>(int, int64_t, int64_t, int64_t, int);
template bool tail3_ok<double>(int, int64_t, int64_t, int64_t, int);
template hipError_t launch_tail3<float>(hipStream_t, const Taps<float> &, int, const float *, int64_t, int64_t, float *, int64_t, int64_t, int, int,
                                        int, int);
template hipError_t launch_tail3<double>(hipStream_t, const Taps<double> &, int, const double *, int64_t, int64_t, double *, int64_t, int64_t, int,
                                         int, int, int);

template bool tail2_ok<float>(int, int, int64_t, int64_t, int, int);
template bool tail2_ok<double>(int, int, int64_t, int64_t, int, int);
template hipError_t launch_tail2<float>(hipStream_t, const Taps<float> &, const float *, int64_t, float *, int64_t, int64_t, int64_t, int, int, int,
                                        int, int);
template hipError_t launch_tail2<double>(hipStream_t, const Taps<double> &, const double *, int64_t, double *, int64_t, int64_t, int64_t, int, int,
                                         int, int, int);

}  // namespace wl
