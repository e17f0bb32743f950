// wl_vlong.hip -- filters the <= 24-tap kernel families do not cover: odd lengths and up to 64 taps.
// Instantiated for the Battle-Lemarie filters of the reference's table (23, 41, 59 taps, wt_main.jl:372-436).
//
// At these lengths a level is bound by VALU issue, not by HBM (59 taps: 236 multiply/add instructions per sample pair
// and pass against 16 bytes of traffic), so the design goal is "every instruction in the loop is a multiply or an add":
//
//   k_vl_axis_fwd / k_vl_axis_inv   one level along a STRIDED axis (dim 2 of a matrix with contiguous rows): a lane owns
//       one row, the wave marches along the axis with the whole filter window in registers (F + 6 columns, statically
//       indexed; shifted by 8 columns = 4 coefficient pairs per step), the next step's 8 columns are in flight while the
//       current 4 pairs are computed.  No LDS, no cross-lane traffic, no barrier.  Forward detail coefficients are
//       produced with a shift of (F-1)/2 pairs so that they share the scaling coefficients' window.
//   k_vl_lines_fwd / k_vl_lines_inv one level along CONTIGUOUS lines: a workgroup stages 2048 samples (+ halo, periodic
//       wrap resolved while staging) of one line in LDS, every thread reads its two windows back with constant offsets
//       and produces 4 coefficient pairs (forward) / 8 samples (inverse), stored as 16-byte vectors.
//
// Only the scaling taps h travel to the kernels: g[m] = (-1)^m h[m] exactly (make_taps, wl_internal.h), and
// fl(acc + fl((-h) x)) == fl(acc - fl(h x)), so a detail term is a multiply and an add or a subtract.  Summation order is
// the closed form of wl_internal.h (s: m ascending; d: m descending; inverse: S (m descending) + D (m ascending)),
// bit-identical to the generic kernels.
#include "wl_fast.h"

namespace wl {

template <typename T, int F>
struct TapsH { T h[F]; };
template <typename T, int F>
static TapsH<T, F> shrink_h(const Taps<T> &t)
{
    TapsH<T, F> r;
    for (int i = 0; i < F; ++i) r.h[i] = t.h[i];
    return r;
}

// s[q] = sum_{m = 0..F-1, ascending} h[m] * W[OFF + 2q + m]
template <typename T, int F, int NQ, int OFF, int WN>
__device__ __forceinline__ void vl_scaling(const T (&W)[WN], const TapsH<T, F> &tp, T (&s)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) s[q] = tp.h[0] * W[OFF + 2 * q];
#pragma unroll
    for (int m = 1; m < F; ++m)
#pragma unroll
        for (int q = 0; q < NQ; ++q) s[q] = s[q] + tp.h[m] * W[OFF + 2 * q + m];
}
// d[q] = sum_{m = F-1..0, descending} g[m] * W[OFF + 2q + (F-1-m)],  g[m] = (-1)^m h[m]
template <typename T, int F, int NQ, int OFF, int WN>
__device__ __forceinline__ void vl_detail(const T (&W)[WN], const TapsH<T, F> &tp, T (&d)[NQ])
{
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const T p = tp.h[F - 1] * W[OFF + 2 * q];
        d[q] = ((F - 1) & 1) ? -p : p;
    }
#pragma unroll
    for (int i = 1; i < F; ++i) {
        const int m = F - 1 - i;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T p = tp.h[m] * W[OFF + 2 * q + i];
            d[q] = (m & 1) ? (d[q] - p) : (d[q] + p);
        }
    }
}
// inverse: outputs x[2(p+q)] and x[2(p+q)+1], q < NQ, from SW[j] = s[p - HS + j] and DW[j] = d[p + j], HS = (F-1)/2
template <typename T, int F, int NQ, int SN, int DN>
__device__ __forceinline__ void vl_inverse(const T (&SW)[SN], const T (&DW)[DN], const TapsH<T, F> &tp, T (&out)[2 * NQ])
{
    constexpr int HS = (F - 1) / 2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int m0 = (((F - 1 - r) & 1) == 0) ? F - 1 : F - 2;        // largest tap with (o - m) even
        T S[NQ], D[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) S[q] = tp.h[m0] * SW[q + HS + (r - m0) / 2];
#pragma unroll
        for (int m = m0 - 2; m >= 0; m -= 2)
#pragma unroll
            for (int q = 0; q < NQ; ++q) S[q] = S[q] + tp.h[m] * SW[q + HS + (r - m) / 2];
        const int m1 = r ? 0 : 1;                                       // smallest tap with (o + m - 1) even
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T p = tp.h[m1] * DW[q + (r + m1 - 1) / 2];
            D[q] = (m1 & 1) ? -p : p;
        }
#pragma unroll
        for (int m = m1 + 2; m < F; m += 2)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const T p = tp.h[m] * DW[q + (r + m - 1) / 2];
                D[q] = (m & 1) ? (D[q] - p) : (D[q] + p);
            }
#pragma unroll
        for (int q = 0; q < NQ; ++q) out[2 * q + r] = S[q] + D[q];
    }
}

// ---------------------------------------------------------------------------------------------------
// strided axis
template <typename T, int F>
struct VlAxisArgs {
    const T *src; int64_t lds;
    T *dst; int64_t ldd;
    int64_t R, C;                   // rows (contiguous), axis length (multiple of 8)
    int CH;                         // coefficient pairs per wave task (multiple of 4)
    int nrg, nchunks;               // row groups of 64, chunks along the axis
    TapsH<T, F> tp;
};

// forward: dst[:, k] = s[k], dst[:, C/2 + k] = d[k]
// (column offsets are wave-uniform and advanced incrementally -- offset += stride, reset at the periodic wrap -- so the
//  whole address computation is a handful of scalar adds per column)
template <typename T, int F>
__global__ void __launch_bounds__(256) k_vl_axis_fwd(VlAxisArgs<T, F> a)
{
    constexpr int OD = F & 1, CS = (F - 2 + OD) / 2, WN = F + OD + 6;
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    const int rg = (int)(task % a.nrg), chunk = (int)(task / a.nrg);
    if (chunk >= a.nchunks) return;
    const int64_t row = (int64_t)rg * 64 + lane;
    const bool valid = row < a.R;
    const int C = (int)a.C, nx = C >> 1;
    const int k0 = chunk * a.CH;
    const int kend = (k0 + a.CH < nx) ? (k0 + a.CH) : nx;
    const T *base = a.src + (valid ? row : 0);
    T *out = a.dst + (valid ? row : 0);
    int jc = 2 * k0;                                  // next column to load (periodic)
    while (jc >= C) jc -= C;
    int64_t oj = (int64_t)jc * a.lds;
    T W[WN], N[8];
#pragma unroll
    for (int j = 0; j < WN - 8; ++j) {
        W[j] = base[oj];
        oj += a.lds;
        if (++jc == C) { jc = 0; oj = 0; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        N[e] = base[oj];
        oj += a.lds;
        if (++jc == C) { jc = 0; oj = 0; }
    }
    int kd = k0 + CS;                                 // detail pair index of the step's first output (periodic)
    while (kd >= nx) kd -= nx;
    int64_t os = (int64_t)k0 * a.ldd, od = (int64_t)(nx + kd) * a.ldd;
    const int64_t od0 = (int64_t)nx * a.ldd;
    for (int k = k0; k < kend; k += 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) W[WN - 8 + e] = N[e];
#pragma unroll
        for (int e = 0; e < 8; ++e) {                 // the next step's columns (harmless past the chunk's end: periodic)
            N[e] = base[oj];
            oj += a.lds;
            if (++jc == C) { jc = 0; oj = 0; }
        }
        T s[4], d[4];
        vl_scaling<T, F, 4, 0, WN>(W, a.tp, s);
        vl_detail<T, F, 4, OD, WN>(W, a.tp, d);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (valid) {
                out[os] = s[q];
                out[od] = d[q];
            }
            os += a.ldd;
            od += a.ldd;
            if (++kd == nx) { kd = 0; od = od0; }
        }
#pragma unroll
        for (int j = 0; j < WN - 8; ++j) W[j] = W[j + 8];
    }
}

// inverse: src[:, p] = s[p], src[:, C/2 + p] = d[p]; dst[:, o] = x[o]
template <typename T, int F>
__global__ void __launch_bounds__(256) k_vl_axis_inv(VlAxisArgs<T, F> a)
{
    constexpr int HS = (F - 1) / 2, HD = (F - 1) / 2, SN = HS + 4, DN = HD + 4;
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform
    const int rg = (int)(task % a.nrg), chunk = (int)(task / a.nrg);
    if (chunk >= a.nchunks) return;
    const int64_t row = (int64_t)rg * 64 + lane;
    const bool valid = row < a.R;
    const int C = (int)a.C, nx = C >> 1;
    const int p0 = chunk * a.CH;
    const int pend = (p0 + a.CH < nx) ? (p0 + a.CH) : nx;
    const T *bs = a.src + (valid ? row : 0);
    const T *bd = bs + (int64_t)nx * a.lds;
    T *out = a.dst + (valid ? row : 0);
    int js = p0 - HS, jd = p0;
    while (js < 0) js += nx;
    int64_t osx = (int64_t)js * a.lds, odx = (int64_t)jd * a.lds;
    T SW[SN], DW[DN], NS[4], ND[4];
#pragma unroll
    for (int j = 0; j < HS; ++j) {
        SW[j] = bs[osx];
        osx += a.lds;
        if (++js == nx) { js = 0; osx = 0; }
    }
#pragma unroll
    for (int j = 0; j < HD; ++j) {
        DW[j] = bd[odx];
        odx += a.lds;
        if (++jd == nx) { jd = 0; odx = 0; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        NS[e] = bs[osx];
        ND[e] = bd[odx];
        osx += a.lds;
        odx += a.lds;
        if (++js == nx) { js = 0; osx = 0; }
        if (++jd == nx) { jd = 0; odx = 0; }
    }
    int64_t oo = (int64_t)(2 * p0) * a.ldd;
    for (int p = p0; p < pend; p += 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { SW[HS + e] = NS[e]; DW[HD + e] = ND[e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            NS[e] = bs[osx];
            ND[e] = bd[odx];
            osx += a.lds;
            odx += a.lds;
            if (++js == nx) { js = 0; osx = 0; }
            if (++jd == nx) { jd = 0; odx = 0; }
        }
        T o[8];
        vl_inverse<T, F, 4, SN, DN>(SW, DW, a.tp, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (valid) out[oo] = o[e];
            oo += a.ldd;
        }
#pragma unroll
        for (int j = 0; j < HS; ++j) SW[j] = SW[j + 4];
#pragma unroll
        for (int j = 0; j < HD; ++j) DW[j] = DW[j + 4];
    }
}

template <typename T, int F, int FW>
static hipError_t launch_vl_axis(hipStream_t st, const Taps<T> &taps, const T *src, int64_t lds, T *dst, int64_t ldd, int64_t R,
                                 int64_t C, int cu_count)
{
    VlAxisArgs<T, F> a;
    a.src = src; a.lds = lds; a.dst = dst; a.ldd = ldd; a.R = R; a.C = C;
    a.nrg = (int)((R + 63) / 64);
    const int64_t nx = C >> 1;
    // one round of resident waves: chunks per row group = wave slots of the device / row groups (a second, partly filled
    // round of these long, compute-bound tasks would cost as much as the first)
    static int occ_blocks[2] = {0, 0};
    if (occ_blocks[FW] == 0) {
        int nb = 0;
        hipError_t oe = FW ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_vl_axis_fwd<T, F>, 256, 0)
                           : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_vl_axis_inv<T, F>, 256, 0);
        occ_blocks[FW] = (oe == hipSuccess && nb > 0) ? nb : 2;
    }
    const int64_t slots = (int64_t)cu_count * occ_blocks[FW] * 4;
    int64_t per_rg = slots / a.nrg;
    if (per_rg < 1) per_rg = 1;
    int64_t ch = (nx + per_rg - 1) / per_rg;
    ch = ((ch + 3) / 4) * 4;
    if (ch < 16) ch = 16;
    const int CH = (int)ch;
    a.CH = CH;
    a.nchunks = (int)((nx + CH - 1) / CH);
    a.tp = shrink_h<T, F>(taps);
    const int64_t tasks = (int64_t)a.nrg * a.nchunks;
    const dim3 grid((unsigned)((tasks + 3) / 4));
    if (FW) hipLaunchKernelGGL((k_vl_axis_fwd<T, F>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_vl_axis_inv<T, F>), grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// contiguous lines
template <typename T, int F>
struct VlLineArgs {
    const T *a; int64_t a_ls;       // fw: src lines          inv: approximation source
    const T *b; int64_t b_ls;       // fw: unused             inv: detail source
    T *o0; int64_t o0_ls;           // fw: s destination      inv: dst lines
    T *o1; int64_t o1_ls;           // fw: d destination      inv: unused
    int64_t n;                      // line length (multiple of 8)
    TapsH<T, F> tp;
};
// coefficient pairs per thread: a thread's windows start 2*NQ (forward) / NQ (inverse) elements apart, so the LDS image
// carries one pad element per such group -- the lanes of a wave then read with an odd element stride (no bank conflicts)
template <typename T> constexpr int vl_nq() { return sizeof(T) == 4 ? 8 : 4; }
constexpr int vl_log2(int v) { return v <= 1 ? 0 : 1 + vl_log2(v >> 1); }

template <typename T, int N>
__device__ __forceinline__ void v_ld(const T *p, T (&v)[N])
{
    constexpr int C = (16 / sizeof(T)) < N ? (16 / sizeof(T)) : N;
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t = *reinterpret_cast<const V *>(p + c * C);
#pragma unroll
        for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void v_st(T *p, const T (&v)[N])
{
    constexpr int C = (16 / sizeof(T)) < N ? (16 / sizeof(T)) : N;
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        *reinterpret_cast<V *>(p + c * C) = t;
    }
}

template <typename T, int F>
__global__ void __launch_bounds__(256) k_vl_lines_fwd(VlLineArgs<T, F> a)
{
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NQ = vl_nq<T>(), G = 2 * NQ, SHIFT = vl_log2(G), TP = 256 * NQ;
    constexpr int HB = ((F - 2 + G - 1) / G) * G;     // samples staged on either side of the tile (multiple of the pad group)
    constexpr int TL = 2 * TP + 2 * HB;
    __shared__ T lds[TL + TL / G + 1];
    const int tid = threadIdx.x;
    const int64_t n = a.n, nx = n >> 1;
    const int64_t k0 = (int64_t)blockIdx.x * TP;
    const int tp_ = (int)((nx - k0 < TP) ? (nx - k0) : TP);             // pairs of this tile (multiple of 4)
    const T *src = a.a + (int64_t)blockIdx.y * a.a_ls;
    const int staged = 2 * tp_ + 2 * HB;
    for (int u = tid * VEC; u < staged; u += 256 * VEC) {
        int64_t idx = 2 * k0 - HB + u;
        while (idx < 0) idx += n;
        while (idx >= n) idx -= n;
        T v[VEC];
        v_ld<T, VEC>(src + idx, v);
        const int pos = u + (u >> SHIFT);
#pragma unroll
        for (int e = 0; e < VEC; ++e) lds[pos + e] = v[e];
    }
    __syncthreads();
    if (NQ * tid >= tp_) return;
    const T *w0 = &lds[(G + 1) * tid];                // + constant: padded position of sample HB + G*tid + jj
    constexpr int WN = F + 2 * NQ - 2;
    T s[NQ], d[NQ];
    {
        T W[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) W[j] = w0[HB + HB / G + j + (j >> SHIFT)];
        vl_scaling<T, F, NQ, 0, WN>(W, a.tp, s);
    }
    {
        T W[WN];
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int jj = j + 2 - F;                 // may be negative: arithmetic shift = floor division
            W[j] = w0[HB + HB / G + jj + (jj >> SHIFT)];
        }
        vl_detail<T, F, NQ, 0, WN>(W, a.tp, d);
    }
    T *so = a.o0 + (int64_t)blockIdx.y * a.o0_ls + k0 + NQ * tid;
    T *dO = a.o1 + (int64_t)blockIdx.y * a.o1_ls + k0 + NQ * tid;
#pragma unroll
    for (int c = 0; c < NQ / 4; ++c) {
        if (NQ * tid + 4 * c < tp_) {
            T t4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = s[4 * c + e];
            v_st<T, 4>(so + 4 * c, t4);
#pragma unroll
            for (int e = 0; e < 4; ++e) t4[e] = d[4 * c + e];
            v_st<T, 4>(dO + 4 * c, t4);
        }
    }
}

template <typename T, int F>
__global__ void __launch_bounds__(256) k_vl_lines_inv(VlLineArgs<T, F> a)
{
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NQ = vl_nq<T>(), SHIFT = vl_log2(NQ), TP = 256 * NQ;
    constexpr int HS = (F - 1) / 2, HD = (F - 1) / 2;
    constexpr int HR = ((HS + NQ - 1) / NQ) * NQ;
    constexpr int TL = TP + HR;
    __shared__ T lds_s[TL + TL / NQ + 1];
    __shared__ T lds_d[TL + TL / NQ + 1];
    const int tid = threadIdx.x;
    const int64_t n = a.n, nx = n >> 1;
    const int64_t p0 = (int64_t)blockIdx.x * TP;
    const int tp_ = (int)((nx - p0 < TP) ? (nx - p0) : TP);
    const T *ss = a.a + (int64_t)blockIdx.y * a.a_ls;
    const T *ds = a.b + (int64_t)blockIdx.y * a.b_ls;
    const int staged = tp_ + HR;
    for (int u = tid * VEC; u < staged; u += 256 * VEC) {
        int64_t is = p0 - HR + u, id = p0 + u;
        while (is < 0) is += nx;
        while (is >= nx) is -= nx;
        while (id >= nx) id -= nx;
        T v[VEC], w[VEC];
        v_ld<T, VEC>(ss + is, v);
        v_ld<T, VEC>(ds + id, w);
        const int pos = u + (u >> SHIFT);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { lds_s[pos + e] = v[e]; lds_d[pos + e] = w[e]; }
    }
    __syncthreads();
    if (NQ * tid >= tp_) return;
    T SW[HS + NQ], DW[HD + NQ];
#pragma unroll
    for (int j = 0; j < HS + NQ; ++j) {
        const int jj = j - HS;                        // lds_s sample HR + NQ*tid + jj
        SW[j] = lds_s[(NQ + 1) * tid + HR + HR / NQ + jj + (jj >> SHIFT)];
    }
#pragma unroll
    for (int j = 0; j < HD + NQ; ++j) DW[j] = lds_d[(NQ + 1) * tid + j + (j >> SHIFT)];
    T o[2 * NQ];
    vl_inverse<T, F, NQ, HS + NQ, HD + NQ>(SW, DW, a.tp, o);
    T *op = a.o0 + (int64_t)blockIdx.y * a.o0_ls + 2 * (p0 + NQ * tid);
#pragma unroll
    for (int c = 0; c < NQ / 4; ++c) {
        if (NQ * tid + 4 * c < tp_) {
            T t8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t8[e] = o[8 * c + e];
            v_st<T, 8>(op + 8 * c, t8);
        }
    }
}

template <typename T, int F, int FW>
static hipError_t launch_vl_lines(hipStream_t st, const Taps<T> &taps, VlLineArgs<T, F> a, int64_t n, int64_t nlines)
{
    constexpr int TP = 256 * vl_nq<T>();
    a.n = n;
    a.tp = shrink_h<T, F>(taps);
    if (nlines <= 0) return hipSuccess;
    const int64_t ntiles = ((n >> 1) + TP - 1) / TP;
    for (int64_t l0 = 0; l0 < nlines; l0 += 32768) {
        const int64_t nb = (nlines - l0 < 32768) ? (nlines - l0) : 32768;
        VlLineArgs<T, F> b = a;
        b.a = a.a + l0 * a.a_ls; b.b = a.b ? a.b + l0 * a.b_ls : nullptr;
        b.o0 = a.o0 + l0 * a.o0_ls; b.o1 = a.o1 ? a.o1 + l0 * a.o1_ls : nullptr;
        if (FW) hipLaunchKernelGGL((k_vl_lines_fwd<T, F>), dim3((unsigned)ntiles, (unsigned)nb), dim3(256), 0, st, b);
        else hipLaunchKernelGGL((k_vl_lines_inv<T, F>), dim3((unsigned)ntiles, (unsigned)nb), dim3(256), 0, st, b);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
#define WL_DISPATCH_FV(F_, ...)                              \
    switch (F_) {                                            \
    case 12: { constexpr int FF = 12; __VA_ARGS__; } break;  \
    case 14: { constexpr int FF = 14; __VA_ARGS__; } break;  \
    case 16: { constexpr int FF = 16; __VA_ARGS__; } break;  \
    case 18: { constexpr int FF = 18; __VA_ARGS__; } break;  \
    case 20: { constexpr int FF = 20; __VA_ARGS__; } break;  \
    case 24: { constexpr int FF = 24; __VA_ARGS__; } break;  \
    case 23: { constexpr int FF = 23; __VA_ARGS__; } break;  \
    case 41: { constexpr int FF = 41; __VA_ARGS__; } break;  \
    case 59: { constexpr int FF = 59; __VA_ARGS__; } break;  \
    default: break;                                          \
    }

// instantiated lengths: the Battle-Lemarie filters (no other fast path), and the 12..24-tap families for the levels that
// are too small for their ring kernels (wl_axis.hip: lines >= 512, axis lengths in multiples of 32)
bool vlong_filter_ok(int F) { return F == 23 || F == 41 || F == 59 || F == 12 || F == 14 || F == 16 || F == 18 || F == 20 || F == 24; }
// lengths that never use the ring kernels: the Battle filters, and 24 taps (measured: 8192^2 level 262 us here, 309 us there)
bool vlong_only(int F) { return F == 23 || F == 41 || F == 59 || F == 24; }

template <typename T>
hipError_t vl_lines_fwd(hipStream_t st, const Taps<T> &taps, const T *src, int64_t src_ls, T *sdst, int64_t s_ls, T *ddst,
                        int64_t d_ls, int64_t n, int64_t nlines)
{
    hipError_t e = hipErrorInvalidValue;
    WL_DISPATCH_FV(taps.F, {
        VlLineArgs<T, FF> a;
        a.a = src; a.a_ls = src_ls; a.b = nullptr; a.b_ls = 0; a.o0 = sdst; a.o0_ls = s_ls; a.o1 = ddst; a.o1_ls = d_ls;
        e = launch_vl_lines<T, FF, 1>(st, taps, a, n, nlines);
    });
    return e;
}
template <typename T>
hipError_t vl_lines_inv(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls, T *dst,
                        int64_t o_ls, int64_t n, int64_t nlines)
{
    hipError_t e = hipErrorInvalidValue;
    WL_DISPATCH_FV(taps.F, {
        VlLineArgs<T, FF> a;
        a.a = ssrc; a.a_ls = s_ls; a.b = dsrc; a.b_ls = d_ls; a.o0 = dst; a.o0_ls = o_ls; a.o1 = nullptr; a.o1_ls = 0;
        e = launch_vl_lines<T, FF, 0>(st, taps, a, n, nlines);
    });
    return e;
}
template <typename T>
hipError_t vl_axis(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t lds, T *dst, int64_t ldd, int64_t R, int64_t C,
                   int cu_count)
{
    hipError_t e = hipErrorInvalidValue;
    WL_DISPATCH_FV(taps.F, {
        if (fw) e = launch_vl_axis<T, FF, 1>(st, taps, src, lds, dst, ldd, R, C, cu_count);
        else e = launch_vl_axis<T, FF, 0>(st, taps, src, lds, dst, ldd, R, C, cu_count);
    });
    return e;
}
#define WL_INST_VL(T)                                                                                                              \
    template hipError_t vl_lines_fwd<T>(hipStream_t, const Taps<T> &, const T *, int64_t, T *, int64_t, T *, int64_t, int64_t, int64_t); \
    template hipError_t vl_lines_inv<T>(hipStream_t, const Taps<T> &, const T *, int64_t, const T *, int64_t, T *, int64_t, int64_t,    \
                                        int64_t);                                                                                  \
    template hipError_t vl_axis<T>(hipStream_t, const Taps<T> &, int, const T *, int64_t, T *, int64_t, int64_t, int64_t, int);
WL_INST_VL(float)
WL_INST_VL(double)

}  // namespace wl
