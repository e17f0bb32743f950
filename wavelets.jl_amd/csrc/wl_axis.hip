// wl_axis.hip -- single-axis streaming passes used to assemble the 3-D filter-bank transform
// (reference: planes / rows / columns passes of _dwt! 3-D, transforms_filter.jl:246-287).
//
//   k_fwd_axis_stream / k_inv_axis_stream
//       one level along a STRIDED axis of a (rows x axis-length) matrix whose rows are contiguous:
//       lanes own 16 bytes of consecutive rows (coalesced 1 KiB per wave-instruction), the wave marches
//       along the axis with a register ring; no cross-lane traffic.  A 3-D box maps onto it twice:
//       axis 3 = matrix of n0*n1 rows x n2 columns; axis 2 = n2 independent n0 x n1 matrices (blockIdx.y).
//   k_fwd_short_lines / k_inv_short_lines
//       one level along the CONTIGUOUS axis for short lines (n0 in {8,...,512}: volumes are rarely longer):
//       n0/8 lanes hold one whole line (8 samples per lane), 64/(n0/8) lines per wave; the periodic halo
//       comes from the neighbouring lane of the same group by ds_bpermute, so there is no tile overlap.
// Arithmetic: the closed forms of wl_internal.h, bit-identical to the generic kernels.
#include "wl_fast.h"


namespace wl {

template <typename T, int F>
struct TapsA { T h[F]; T g[F]; };
template <typename T, int F>
static TapsA<T, F> shrink_a(const Taps<T> &t)
{
    TapsA<T, F> r;
    for (int i = 0; i < F; ++i) { r.h[i] = t.h[i]; r.g[i] = t.g[i]; }
    return r;
}
template <typename T, int N>
__device__ __forceinline__ void a_ld(const T *p, T (&v)[N])
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
__device__ __forceinline__ void a_st(T *p, const T (&v)[N])
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

// ---------------------------------------------------------------------------------------------------
// column ring of the forward axis kernel: 16 slots up to 10 taps, 32 beyond (loads run (R-F)/2 steps ahead)
constexpr int axis_ring(int F) { return F <= 10 ? 16 : 32; }
// coefficient rings of the inverse axis kernel: (F-2)/2 + 1 live columns + the prefetch distance
constexpr int axis_ring_inv(int F) { return F <= 10 ? 8 : 16; }

template <typename T, int F>
struct AxisArgs {
    const T *src; int64_t lds; int64_t bs_src;     // column stride, batch stride
    T *dst; int64_t ldd; int64_t bs_dst;
    int64_t R, C;                                  // rows (contiguous), axis length
    int TJ;                                        // forward: input columns per chunk (multiple of 16); inverse: output pairs per chunk (multiple of 8)
    int nstrips, nchunks;
    int chunk0;                                    // first chunk of this launch (slab launches of the 3-D level cover a chunk range)
    TapsA<T, F> tp;
};

// forward: dst[:, p] = s[p], dst[:, C/2 + p] = d[p]
template <typename T, int F, int RPL>
__global__ void __launch_bounds__(64) k_fwd_axis_stream(AxisArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2, R = axis_ring(F), U = R / 2, PFD = (R - F) / 2;
    const int lane = threadIdx.x;
    const int strip = (int)(blockIdx.x % (unsigned)a.nstrips);
    const int chunk = a.chunk0 + (int)(blockIdx.x / (unsigned)a.nstrips);
    const int64_t row = ((int64_t)strip * 64 + lane) * RPL;
    const bool valid = row < a.R;
    const int64_t rr = valid ? row : 0;
    const int64_t C = a.C, nx = C >> 1;
    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < C) ? (j0 + a.TJ) : C;
    const int S = (int)((jend - j0) >> 1);          // multiple of U
    const T *base = a.src + (int64_t)blockIdx.y * a.bs_src + rr;
    T *out = a.dst + (int64_t)blockIdx.y * a.bs_dst + rr;
    T ring[R][RPL];
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= C) jc -= C;
        a_ld<T, RPL>(base + jc * a.lds, ring[c]);
    }
    const int64_t kbase = j0 >> 1;
    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= C) jc -= C;
                a_ld<T, RPL>(base + jc * a.lds, ring[(2 * u + R - 2 + e) % R]);
            }
        }
        T sv[RPL], dv[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            T s = a.tp.h[0] * ring[(2 * u) % R][q];
#pragma unroll
            for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * ring[(2 * u + m) % R][q];
            T d = a.tp.g[F - 1] * ring[(2 * u) % R][q];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + a.tp.g[m] * ring[(2 * u + F - 1 - m) % R][q];
            sv[q] = s;
            dv[q] = d;
        }
        if (valid) {
            const int64_t k = kbase + t;
            int64_t kd = k + SH;
            if (kd >= nx) kd -= nx;
            a_st<T, RPL>(out + k * a.ldd, sv);
            a_st<T, RPL>(out + (nx + kd) * a.ldd, dv);
        }
    };
    int t0 = 0;
    for (; t0 < S - U; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u, true);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(t0 + u, u, u < U - PFD);
}

// inverse: src[:, 0..C/2) = s, src[:, C/2..C) = d  ->  dst[:, 0..C)
template <typename T, int F>
__device__ __forceinline__ void a_inv_pair(const T *sw, const T *dw, const TapsA<T, F> &tp, T &xe, T &xo)
{
    constexpr int SH = (F - 2) / 2;
    T Se = tp.h[F - 2] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Se = Se + tp.h[F - 2 - 2 * q] * sw[q];
    T De = tp.g[1] * dw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) De = De + tp.g[1 + 2 * q] * dw[q];
    xe = Se + De;
    T So = tp.h[F - 1] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) So = So + tp.h[F - 1 - 2 * q] * sw[q];
    T Do = tp.g[0] * dw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Do = Do + tp.g[2 * q] * dw[q];
    xo = So + Do;
}

template <typename T, int F, int RPL>
__global__ void __launch_bounds__(64) k_inv_axis_stream(AxisArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2, R = axis_ring_inv(F);
    const int lane = threadIdx.x;
    const int strip = (int)(blockIdx.x % (unsigned)a.nstrips);
    const int chunk = (int)(blockIdx.x / (unsigned)a.nstrips);
    const int64_t row = ((int64_t)strip * 64 + lane) * RPL;
    const bool valid = row < a.R;
    const int64_t rr = valid ? row : 0;
    const int64_t nx = a.C >> 1;
    const int64_t p0 = (int64_t)chunk * a.TJ;
    const int64_t pend = (p0 + a.TJ < nx) ? (p0 + a.TJ) : nx;
    const int S = (int)(pend - p0);
    const T *sbase = a.src + (int64_t)blockIdx.y * a.bs_src + rr;
    const T *dbase = sbase + nx * a.lds;
    T *out = a.dst + (int64_t)blockIdx.y * a.bs_dst + rr;
    T rs[R][RPL], rd[R][RPL];
#pragma unroll
    for (int c = 0; c < R - 1; ++c) {
        int64_t js = p0 - SH + c;
        if (js < 0) js += nx;
        if (js >= nx) js -= nx;
        int64_t jd = p0 + c;
        if (jd >= nx) jd -= nx;
        a_ld<T, RPL>(sbase + js * a.lds, rs[c]);
        a_ld<T, RPL>(dbase + jd * a.lds, rd[c]);
    }
    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {
            int64_t js = p0 - SH + t + (R - 1);
            if (js >= nx) js -= nx;
            int64_t jd = p0 + t + (R - 1);
            if (jd >= nx) jd -= nx;
            a_ld<T, RPL>(sbase + js * a.lds, rs[(u + R - 1) % R]);
            a_ld<T, RPL>(dbase + jd * a.lds, rd[(u + R - 1) % R]);
        }
        T xe[RPL], xo[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int i = 0; i <= SH; ++i) { sw[i] = rs[(u + i) % R][q]; dw[i] = rd[(u + i) % R][q]; }
            a_inv_pair<T, F>(sw, dw, a.tp, xe[q], xo[q]);
        }
        if (valid) {
            const int64_t p = p0 + t;
            a_st<T, RPL>(out + (2 * p) * a.ldd, xe);
            a_st<T, RPL>(out + (2 * p + 1) * a.ldd, xo);
        }
    };
    int t0 = 0;
    for (; t0 < S - R; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u, true);
    }
#pragma unroll
    for (int u = 0; u < R; ++u) step(t0 + u, u, u <= SH);
}

// ---------------------------------------------------------------------------------------------------
// short contiguous lines; lines are addressed as a 2-D grid (i2 < c2, i3 < c3)
template <typename T, int F>
struct ShortArgs {
    const T *a; int64_t a2, a3;      // fw: src          inv: approximation source
    const T *b; int64_t b2, b3;      // fw: unused       inv: detail source
    T *o0; int64_t o02, o03;         // fw: s dest       inv: dst
    T *o1; int64_t o12, o13;         // fw: d dest       inv: unused
    // low-low corner (lines with i2 < l2 and i3 < l3): fw: their s goes to ll instead of o0; inv: their approximation
    // comes from ll instead of a.  ll == nullptr: no redirection.
    T *ll; int64_t ll2, ll3; int l2; int64_t l3;
    int n;                           // line length: 8*G, G | 64
    int G, c2;
    int64_t nlines;                  // c2 * c3
    TapsA<T, F> tp;
};

template <typename T, int F, int FW>
__global__ void __launch_bounds__(256) k_short_lines(ShortArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2;
    const int lane = threadIdx.x & 63;
    const int G = a.G, lpw = 64 / G;
    const int g = lane / G, r = lane - g * G;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t li = wave * lpw + g;
    const bool valid = li < a.nlines;
    const int64_t lc = valid ? li : 0;
    const int64_t i3 = lc / a.c2, i2 = lc - i3 * a.c2;
    const int nxt = g * G + (r + 1 == G ? 0 : r + 1), prv = g * G + (r == 0 ? G - 1 : r - 1);
    const bool corner = a.ll != nullptr && i2 < a.l2 && i3 < a.l3;
    if (FW) {
        T v[8];
        a_ld<T, 8>(a.a + i2 * a.a2 + i3 * a.a3 + 8 * r, v);
        constexpr int LO = -(F - 2), HI = 8 + F - 2;
        T ext[HI - LO];
#pragma unroll
        for (int e = 0; e < 8; ++e) ext[e - LO] = v[e];
#pragma unroll
        for (int e = 0; e < F - 2; ++e) {
            ext[8 + e - LO] = __shfl(v[e], nxt, 64);                 // next lane's first F-2 samples
            ext[e] = __shfl(v[8 - (F - 2) + e], prv, 64);            // previous lane's last F-2 samples
        }
        T so[4], dO[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            T s = a.tp.h[0] * ext[2 * q - LO];
#pragma unroll
            for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * ext[2 * q + m - LO];
            T d = a.tp.g[F - 1] * ext[2 * q + 1 - (F - 1) - LO];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + a.tp.g[m] * ext[2 * q + 1 - m - LO];
            so[q] = s;
            dO[q] = d;
        }
        if (valid) {
            a_st<T, 4>((corner ? a.ll + i2 * a.ll2 + i3 * a.ll3 : a.o0 + i2 * a.o02 + i3 * a.o03) + 4 * r, so);
            a_st<T, 4>(a.o1 + i2 * a.o12 + i3 * a.o13 + 4 * r, dO);
        }
    } else {
        T s[4], d[4];
        a_ld<T, 4>((corner ? a.ll + i2 * a.ll2 + i3 * a.ll3 : a.a + i2 * a.a2 + i3 * a.a3) + 4 * r, s);
        a_ld<T, 4>(a.b + i2 * a.b2 + i3 * a.b3 + 4 * r, d);
        T sx[4 + SH], dx[4 + SH];
#pragma unroll
        for (int i = 0; i < 4; ++i) { sx[SH + i] = s[i]; dx[i] = d[i]; }
#pragma unroll
        for (int i = 0; i < SH; ++i) {
            sx[i] = __shfl(s[4 - SH + i], prv, 64);
            dx[4 + i] = __shfl(d[i], nxt, 64);
        }
        T out[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) a_inv_pair<T, F>(&sx[p], &dx[p], a.tp, out[2 * p], out[2 * p + 1]);
        if (valid) a_st<T, 8>(a.o0 + i2 * a.o02 + i3 * a.o03 + 8 * r, out);
    }
}

// ---------------------------------------------------------------------------------------------------
// Long filters (12..24 taps) on contiguous lines: the 8-samples-per-lane layout of k_fwd1d_stream / k_inv1d_stream,
// with the halo gathered from up to three lanes on either side by ds_bpermute (the DPP single-neighbour trick only
// reaches F <= 10).  Wave tiles overlap by the halo lanes; one line per blockIdx.y.
template <typename T, int F>
struct LongArgs {
    const T *a; int64_t a_ls;       // fw: src lines          inv: approximation source
    const T *b; int64_t b_ls;       // fw: unused             inv: detail source
    T *o0; int64_t o0_ls;           // fw: s destination      inv: dst lines
    T *o1; int64_t o1_ls;           // fw: d destination      inv: unused
    int64_t n;                      // line length (multiple of 8, >= 512)
    int64_t ntiles;
    TapsA<T, F> tp;
};

template <typename T, int F, int FW>
__global__ void __launch_bounds__(256) k_long_lines(LongArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2;
    constexpr int HF = (F - 2 + 7) / 8;           // forward: halo lanes on each side (F-2 samples, 8 per lane)
    constexpr int HI = (SH + 3) / 4;              // inverse: halo lanes on each side (SH coefficients, 4 per lane)
    constexpr int HL = FW ? HF : HI;
    constexpr int VP = (64 - 2 * HL) * 4;         // pairs owned by a wave tile
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t nx = a.n >> 1;
    const int64_t line = blockIdx.y;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k0 = tile * VP + (int64_t)(lane - HL) * 4;       // first pair of this lane (may wrap)
        int64_t kw = k0;
        if (kw < 0) kw += nx;
        if (kw >= nx) kw -= nx;
        const bool store = lane >= HL && lane < 64 - HL && k0 < nx;
        if (FW) {
            T v[8];
            a_ld<T, 8>(a.a + line * a.a_ls + 2 * kw, v);
            constexpr int LO = F - 2;                                   // ext[LO + e] = sample e of this lane
            T ext[8 + 2 * (F - 2)];
#pragma unroll
            for (int e = 0; e < 8; ++e) ext[LO + e] = v[e];
#pragma unroll
            for (int e = 0; e < F - 2; ++e) {
                ext[LO + 8 + e] = __shfl_down(v[e % 8], 1 + e / 8, 64);                       // following samples
                ext[LO - 1 - e] = __shfl_up(v[7 - (e % 8)], 1 + e / 8, 64);                   // preceding samples
            }
            T so[4], dO[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                T sacc = a.tp.h[0] * ext[LO + 2 * q];
#pragma unroll
                for (int m = 1; m < F; ++m) sacc = sacc + a.tp.h[m] * ext[LO + 2 * q + m];
                T dacc = a.tp.g[F - 1] * ext[LO + 2 * q + 1 - (F - 1)];
#pragma unroll
                for (int m = F - 2; m >= 0; --m) dacc = dacc + a.tp.g[m] * ext[LO + 2 * q + 1 - m];
                so[q] = sacc;
                dO[q] = dacc;
            }
            if (store) {
                a_st<T, 4>(a.o0 + line * a.o0_ls + k0, so);
                a_st<T, 4>(a.o1 + line * a.o1_ls + k0, dO);
            }
        } else {
            T sv[4], dv[4];
            a_ld<T, 4>(a.a + line * a.a_ls + kw, sv);
            a_ld<T, 4>(a.b + line * a.b_ls + kw, dv);
            // sx[i] = s[p = i - SH], dx[i] = d[p = i] relative to this lane's first pair
            T sx[4 + SH], dx[4 + SH];
#pragma unroll
            for (int i = 0; i < 4; ++i) { sx[SH + i] = sv[i]; dx[i] = dv[i]; }
#pragma unroll
            for (int e = 0; e < SH; ++e) {
                sx[SH - 1 - e] = __shfl_up(sv[3 - (e % 4)], 1 + e / 4, 64);
                dx[4 + e] = __shfl_down(dv[e % 4], 1 + e / 4, 64);
            }
            T out[8];
#pragma unroll
            for (int p = 0; p < 4; ++p) a_inv_pair<T, F>(&sx[p], &dx[p], a.tp, out[2 * p], out[2 * p + 1]);
            if (store) a_st<T, 8>(a.o0 + line * a.o0_ls + 2 * k0, out);
        }
    }
}

template <typename T, int F, int FW>
static hipError_t launch_long(hipStream_t st, const Taps<T> &taps, LongArgs<T, F> a, int64_t n, int64_t nlines, int cu_count)
{
    constexpr int SH = (F - 2) / 2, HL = FW ? (F - 2 + 7) / 8 : (SH + 3) / 4, VP = (64 - 2 * HL) * 4;
    a.n = n;
    a.ntiles = ((n >> 1) + VP - 1) / VP;
    a.tp = shrink_a<T, F>(taps);
    if (nlines <= 0) return hipSuccess;
    int64_t nblk = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 32;
    if (nblk * nlines > cap) { nblk = cap / nlines; if (nblk < 1) nblk = 1; }
    for (int64_t l0 = 0; l0 < nlines; l0 += 32768) {
        const int64_t nb = (nlines - l0 < 32768) ? (nlines - l0) : 32768;
        LongArgs<T, F> b = a;
        b.a = a.a + l0 * a.a_ls; b.b = a.b ? a.b + l0 * a.b_ls : nullptr; b.o0 = a.o0 + l0 * a.o0_ls; b.o1 = a.o1 ? a.o1 + l0 * a.o1_ls : nullptr;
        hipLaunchKernelGGL((k_long_lines<T, F, FW>), dim3((unsigned)nblk, (unsigned)nb), dim3(256), 0, st, b);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
static inline bool a_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool short_ok(int64_t n) { return n >= 16 && n <= 512 && (n % 8) == 0 && (64 % (n / 8)) == 0; }

template <typename T, int F, int FW>
static hipError_t launch_short(hipStream_t st, const Taps<T> &taps, ShortArgs<T, F> a, int n, int c2, int64_t c3)
{
    a.n = n; a.G = n / 8; a.c2 = c2; a.nlines = (int64_t)c2 * c3;
    a.tp = shrink_a<T, F>(taps);
    if (a.nlines <= 0) return hipSuccess;
    const int lpw = 64 / a.G;
    const int64_t nwaves = (a.nlines + lpw - 1) / lpw;
    hipLaunchKernelGGL((k_short_lines<T, F, FW>), dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}

template <typename T, int F, int FW>
static hipError_t launch_axis(hipStream_t st, const Taps<T> &taps, const T *src, int64_t lds, int64_t bs_src,
                              T *dst, int64_t ldd, int64_t bs_dst, int64_t R, int64_t C, int64_t batch, int cu_count,
                              int64_t col_lo = 0, int64_t col_hi = -1, int tj_forced = 0)
{
    constexpr int RPL = 16 / sizeof(T);
    AxisArgs<T, F> a;
    a.src = src; a.lds = lds; a.bs_src = bs_src; a.dst = dst; a.ldd = ldd; a.bs_dst = bs_dst; a.R = R; a.C = C; a.chunk0 = 0;
    a.nstrips = (int)((R + 64 * RPL - 1) / (64 * RPL));
    const int64_t units = FW ? C : (C >> 1);           // chunked quantity: input columns (fw) / output pairs (inv)
    const int unit = FW ? axis_ring(F) : axis_ring_inv(F);
    int TJ = FW ? 128 : 64;
    while (TJ > unit && (int64_t)a.nstrips * ((units + TJ - 1) / TJ) * batch < (int64_t)cu_count * 8) TJ >>= 1;
    if (tj_forced > 0) TJ = tj_forced;
    a.TJ = TJ;
    a.nchunks = (int)((units + TJ - 1) / TJ);
    if (col_hi >= 0) {                             // (forward only) the chunks of input columns [col_lo, col_hi): both multiples of TJ
        a.chunk0 = (int)(col_lo / TJ);
        a.nchunks = (int)((col_hi - col_lo) / TJ);
    }
    a.tp = shrink_a<T, F>(taps);
    for (int64_t b0 = 0; b0 < batch; b0 += 32768) {
        const int64_t nb = (batch - b0 < 32768) ? (batch - b0) : 32768;
        AxisArgs<T, F> b = a;
        b.src = a.src + b0 * bs_src; b.dst = a.dst + b0 * bs_dst;
        if (FW) hipLaunchKernelGGL((k_fwd_axis_stream<T, F, RPL>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)nb), dim3(64), 0, st, b);
        else hipLaunchKernelGGL((k_inv_axis_stream<T, F, RPL>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)nb), dim3(64), 0, st, b);
    }
    return hipGetLastError();
}

#define WL_DISPATCH_FA(F_, ...)                              \
    switch (F_) {                                            \
    case 2: { constexpr int FF = 2; __VA_ARGS__; } break;    \
    case 4: { constexpr int FF = 4; __VA_ARGS__; } break;    \
    case 6: { constexpr int FF = 6; __VA_ARGS__; } break;    \
    case 8: { constexpr int FF = 8; __VA_ARGS__; } break;    \
    case 10: { constexpr int FF = 10; __VA_ARGS__; } break;  \
    default: break;                                          \
    }

// One forward 3-D level: box (n0,n1,n2) read from `cur` (strides 1, c1, c2 with c1 == n0), details to y
// (dense full array strides 1, y1, y2), LLL corner to `ll` (dense h0,h1,h2) or to y when ll == nullptr.
// T0/T1: dense scratch of the box size.  Returns false when the shape is not eligible.
template <typename T>
bool fast3d_fwd_level(hipStream_t st, const Taps<T> &taps, const T *cur, int64_t c1, int64_t c2, T *y, int64_t y1, int64_t y2,
                      T *ll, const int64_t n[3], T *T0, T *T1, int cu_count, hipError_t *err, const char **kname)
{
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    *err = hipSuccess;
    if (kname) *kname = "k_fwd_axis_stream";
    // round 6: the whole level in one pass over HBM where the one-pass kernel takes the shape (wl_fwd3d.hip)
    if (fwd3d_one_ok<T>(F, cur, c1, c2, y, y1, y2, ll, n)) {
        *err = fwd3d_one_launch<T>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
        if (kname) *kname = "k_fwd3d_one";
        return true;
    }
    // ... and the small levels between the streaming sizes and the one-workgroup tail in one launch each (wl_level3.hip)
    if (level3_lds_ok<T>(F, n) && cur != y) {
        *err = level3_lds_launch<T>(st, taps, 1, cur, c1, c2, y, y1, y2, (const T *)nullptr, ll, n);
        if (kname) *kname = "k_level3_lds";
        return true;
    }
    const int64_t n0 = n[0], n1 = n[1], n2 = n[2], h0 = n0 >> 1, h1 = n1 >> 1, h2 = n2 >> 1;
    if ((F % 2) != 0 || F > 10 || !short_ok(n0) || n1 < 16 || n2 < 16 || (n1 % 16) != 0 || (n2 % 16) != 0 || c1 != n0 ||
        (c2 % VEC) != 0 || (y1 % VEC) != 0 || (y2 % VEC) != 0 || !a_al16(cur) || !a_al16(y) || !a_al16(T0) || !a_al16(T1) ||
        (ll && !a_al16(ll)) || n1 > 32767) {
        // not a shape of the axis kernels: the one-pass level from 2^19 elements, the LDS blocks up to 2^20, before the any-extent
        // kernels take it (three passes per level)
        if (fwd3d_one_ok<T>(F, cur, c1, c2, y, y1, y2, ll, n, true)) {
            *err = fwd3d_one_launch<T>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
            if (kname) *kname = "k_fwd3d_one";
            return true;
        }
        if (level3_lds_ok<T>(F, n, true) && cur != y) {
            *err = level3_lds_launch<T>(st, taps, 1, cur, c1, c2, y, y1, y2, (const T *)nullptr, ll, n);
            if (kname) *kname = "k_level3_lds";
            return true;
        }
        return false;
    }
    bool ok = false;
    WL_DISPATCH_FA(F, {
        // An option, OFF by default (round 5, measured): the level in SLABS of output plane pairs -- the axis-3 pass of a slab writes
        // 2 * slab planes of T0 and the plane kernel consumes exactly those planes next, in the hope that the intermediate array stays
        // in the Infinity Cache instead of making a round trip through HBM.  512^3 db4 L = 9: whole-box passes 566 us; slabs of 64 / 32 /
        // 16 / 8 plane pairs 593 / 611 / 767 / 1020 us -- three smaller launches per slab cost more than the cache gives back (the two
        // whole-box passes of level 1 already move their 2.1 GB at 4.7 TB/s).  Kept for the A/B (bit-identical, tested).
        const int64_t P = n0 * n1, SHp = (FF - 2) / 2;
        int64_t slab = opt("WL_3D_SLAB", 0);                         // output plane pairs per slab; 0: whole-box passes
        if (slab > 0) {
            while (slab > 8 && 2 * slab * P * (int64_t)sizeof(T) > ((int64_t)opt("WL_3D_SLAB_MIB", 64) << 20)) slab >>= 1;
            int tj = 128;
            while (tj > 16 && ((2 * slab) % tj) != 0) tj >>= 1;
            if (h2 % slab != 0 || (2 * slab) % tj != 0 || h2 <= slab || P * n2 * (int64_t)sizeof(T) < ((int64_t)opt("WL_3D_SLAB_MIN_MIB", 256) << 20) ||
                n0 < 256 || opt("WL_NO_PLANES", 0) != 0)
                slab = 0;
            if (slab > 0) {
                bool all = true;
                for (int64_t k0 = 0; k0 < h2 && *err == hipSuccess; k0 += slab) {
                    const int64_t k1 = k0 + slab;
                    *err = launch_axis<T, FF, 1>(st, taps, cur, c2, 0, T0, P, 0, P, n2, 1, cu_count, 2 * k0, 2 * k1, tj);
                    if (*err != hipSuccess) break;
                    // scaling planes k0 .. k1-1: their approximation quadrant goes on to ll
                    all = all && fwd2d_planes<T>(st, taps, T0 + k0 * P, y + k0 * y2, y1, y2, ll ? ll + k0 * h0 * h1 : nullptr, n0, n1, k1 - k0,
                                                 (int)(k1 - k0), cu_count, err);
                    if (*err != hipSuccess || !all) break;
                    // detail planes h2 + (k + SH) mod h2
                    int64_t d0 = k0 + SHp, d1 = k1 + SHp;
                    if (d0 >= h2) { d0 -= h2; d1 -= h2; }
                    const int64_t e1 = d1 > h2 ? h2 : d1;
                    all = all && fwd2d_planes<T>(st, taps, T0 + (h2 + d0) * P, y + (h2 + d0) * y2, y1, y2, nullptr, n0, n1, e1 - d0, 0, cu_count, err);
                    if (*err == hipSuccess && all && d1 > h2)
                        all = all && fwd2d_planes<T>(st, taps, T0 + h2 * P, y + h2 * y2, y1, y2, nullptr, n0, n1, d1 - h2, 0, cu_count, err);
                }
                if (!all && *err == hipSuccess) *err = hipErrorInvalidValue;   // (fwd2d_planes accepted this shape for the whole box below before)
                ok = true;
                break;
            }
        }
        // planes: axis 3 on the (n0*n1) x n2 matrix
        *err = launch_axis<T, FF, 1>(st, taps, cur, c2, 0, T0, n0 * n1, 0, n0 * n1, n2, 1, cu_count);
        // rows + columns of every plane in ONE launch when the planes are big enough for the fused 2-D level kernel
        bool planes = false;
        if (*err == hipSuccess && opt("WL_NO_PLANES", 0) == 0)
            planes = fwd2d_planes<T>(st, taps, T0, y, y1, y2, ll, n0, n1, n2, (int)h2, cu_count, err);
        // rows: axis 2 on n2 matrices of n0 x n1
        if (!planes && *err == hipSuccess) *err = launch_axis<T, FF, 1>(st, taps, T0, n0, n0 * n1, T1, n0, n0 * n1, n0, n1, n2, cu_count);
        // columns: short lines; the low-low corner sends its approximation to ll
        if (!planes && *err == hipSuccess) {
            ShortArgs<T, FF> s;
            s.a = T1; s.a2 = n0; s.a3 = n0 * n1; s.b = nullptr; s.b2 = s.b3 = 0;
            s.o0 = y; s.o02 = y1; s.o03 = y2; s.o1 = y + h0; s.o12 = y1; s.o13 = y2;
            s.ll = ll; s.ll2 = h0; s.ll3 = h0 * h1; s.l2 = (int)h1; s.l3 = h2;
            *err = launch_short<T, FF, 1>(st, taps, s, (int)n0, (int)n1, n2);
        }
        ok = true;
    });
    return ok;
}

// One inverse 3-D level (output box n): approximation from `llsrc` (dense h0,h1,h2) or from x when nullptr,
// details from x (dense full strides 1, x1, x2); result to `out` (strides 1, o1, o2 with o1 == n0).
template <typename T>
bool fast3d_inv_level(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *llsrc,
                      T *out, int64_t o1, int64_t o2, const int64_t n[3], T *T0, T *T1, int cu_count, hipError_t *err, const char **kname)
{
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    *err = hipSuccess;
    if (kname) *kname = "k_inv_axis_stream";
    if (inv3d_one_ok<T>(F, x, x1, x2, llsrc, out, o1, o2, n)) {
        *err = inv3d_one_launch<T>(st, taps, x, x1, x2, llsrc, out, o1, o2, n, cu_count);
        if (kname) *kname = "k_inv3d_one";
        return true;
    }
    if (level3_lds_ok<T>(F, n) && x != out && llsrc != out) {
        *err = level3_lds_launch<T>(st, taps, 0, x, x1, x2, out, o1, o2, llsrc, (T *)nullptr, n);
        if (kname) *kname = "k_level3_lds";
        return true;
    }
    const int64_t n0 = n[0], n1 = n[1], n2 = n[2], h0 = n0 >> 1, h1 = n1 >> 1, h2 = n2 >> 1;
    if ((F % 2) != 0 || F > 10 || !short_ok(n0) || n1 < 16 || n2 < 16 || (n1 % 16) != 0 || (n2 % 16) != 0 || o1 != n0 ||
        (o2 % VEC) != 0 || (x1 % VEC) != 0 || (x2 % VEC) != 0 || !a_al16(x) || !a_al16(out) || !a_al16(T0) || !a_al16(T1) ||
        (llsrc && !a_al16(llsrc)) || n1 > 32767 || (h0 % 4) != 0) {
        if (inv3d_one_ok<T>(F, x, x1, x2, llsrc, out, o1, o2, n, true)) {
            *err = inv3d_one_launch<T>(st, taps, x, x1, x2, llsrc, out, o1, o2, n, cu_count);
            if (kname) *kname = "k_inv3d_one";
            return true;
        }
        if (level3_lds_ok<T>(F, n, true) && x != out && llsrc != out) {
            *err = level3_lds_launch<T>(st, taps, 0, x, x1, x2, out, o1, o2, llsrc, (T *)nullptr, n);
            if (kname) *kname = "k_level3_lds";
            return true;
        }
        return false;
    }
    bool ok = false;
    WL_DISPATCH_FA(F, {
        // columns + rows of every plane in ONE launch when the planes are big enough for the fused 2-D inverse kernel
        bool planes = false;
        if (opt("WL_NO_PLANES", 0) == 0)
            planes = inv2d_planes<T>(st, taps, x, x1, x2, llsrc, T1, n0, n1, n2, (int)h2, cu_count, err);
        // columns first (transforms_filter.jl:269-273): merged lines into T0 (dense box); the low-low corner takes its
        // approximation from the deeper reconstruction
        if (!planes && *err == hipSuccess) {
            ShortArgs<T, FF> s;
            s.a = x; s.a2 = x1; s.a3 = x2; s.b = x + h0; s.b2 = x1; s.b3 = x2;
            s.o0 = T0; s.o02 = n0; s.o03 = n0 * n1; s.o1 = nullptr; s.o12 = s.o13 = 0;
            s.ll = const_cast<T *>(llsrc); s.ll2 = h0; s.ll3 = h0 * h1; s.l2 = (int)h1; s.l3 = h2;
            *err = launch_short<T, FF, 0>(st, taps, s, (int)n0, (int)n1, n2);
        }
        // rows: axis 2 on n2 matrices
        if (!planes && *err == hipSuccess) *err = launch_axis<T, FF, 0>(st, taps, T0, n0, n0 * n1, T1, n0, n0 * n1, n0, n1, n2, cu_count);
        // planes: axis 3
        if (*err == hipSuccess) *err = launch_axis<T, FF, 0>(st, taps, T1, n0 * n1, 0, out, o2, 0, n0 * n1, n2, 1, cu_count);
        ok = true;
    });
    return ok;
}

#define WL_DISPATCH_FL(F_, ...)                              \
    switch (F_) {                                            \
    case 12: { constexpr int FF = 12; __VA_ARGS__; } break;  \
    case 14: { constexpr int FF = 14; __VA_ARGS__; } break;  \
    case 16: { constexpr int FF = 16; __VA_ARGS__; } break;  \
    case 18: { constexpr int FF = 18; __VA_ARGS__; } break;  \
    case 20: { constexpr int FF = 20; __VA_ARGS__; } break;  \
    case 24: { constexpr int FF = 24; __VA_ARGS__; } break;  \
    default: break;                                          \
    }

static bool ring_filter_ok(int F) { return F == 12 || F == 14 || F == 16 || F == 18 || F == 20 || F == 24; }
bool long_filter_ok(int F) { return ring_filter_ok(F) || vlong_filter_ok(F); }
// smallest 2-D level the two-pass long-filter path takes (rows n0 contiguous, n1 columns)
bool long_shape2d_ok(int F, int64_t n0, int64_t n1)
{
    if (vlong_filter_ok(F)) return n0 >= 16 && (n0 % 8) == 0 && n1 >= 16 && (n1 % 8) == 0;
    return n0 >= 512 && (n0 % 8) == 0 && n1 >= 32 && (n1 % 32) == 0;
}

// One forward level of `nlines` lines with a long filter: s -> sdst, d -> ddst (line strides in elements).
template <typename T>
bool long_lines_fwd_level(hipStream_t st, const Taps<T> &taps, const T *src, int64_t src_ls, T *sdst, int64_t s_ls,
                          T *ddst, int64_t d_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err)
{
    constexpr int VEC = 16 / sizeof(T);
    *err = hipSuccess;
    if (!long_filter_ok(taps.F) || n < (vlong_filter_ok(taps.F) ? 16 : 512) || (n % 8) != 0 || !a_al16(src) || !a_al16(sdst) || !a_al16(ddst) ||
        (nlines > 1 && ((src_ls % VEC) != 0 || (s_ls % VEC) != 0 || (d_ls % VEC) != 0)))
        return false;
    if (vlong_only(taps.F) || (vlong_filter_ok(taps.F) && n < 512)) {
        *err = vl_lines_fwd<T>(st, taps, src, src_ls, sdst, s_ls, ddst, d_ls, n, nlines);
        return true;
    }
    bool ok = false;
    WL_DISPATCH_FL(taps.F, {
        LongArgs<T, FF> a;
        a.a = src; a.a_ls = src_ls; a.b = nullptr; a.b_ls = 0; a.o0 = sdst; a.o0_ls = s_ls; a.o1 = ddst; a.o1_ls = d_ls;
        *err = launch_long<T, FF, 1>(st, taps, a, n, nlines, cu_count);
        ok = true;
    });
    return ok;
}
template <typename T>
bool long_lines_inv_level(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls,
                          T *dst, int64_t o_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err)
{
    constexpr int VEC = 16 / sizeof(T);
    *err = hipSuccess;
    if (!long_filter_ok(taps.F) || n < (vlong_filter_ok(taps.F) ? 16 : 512) || (n % 8) != 0 || !a_al16(ssrc) || !a_al16(dsrc) || !a_al16(dst) ||
        (nlines > 1 && ((s_ls % VEC) != 0 || (d_ls % VEC) != 0 || (o_ls % VEC) != 0)))
        return false;
    if (vlong_only(taps.F) || (vlong_filter_ok(taps.F) && n < 512)) {
        *err = vl_lines_inv<T>(st, taps, ssrc, s_ls, dsrc, d_ls, dst, o_ls, n, nlines);
        return true;
    }
    bool ok = false;
    WL_DISPATCH_FL(taps.F, {
        LongArgs<T, FF> a;
        a.a = ssrc; a.a_ls = s_ls; a.b = dsrc; a.b_ls = d_ls; a.o0 = dst; a.o0_ls = o_ls; a.o1 = nullptr; a.o1_ls = 0;
        *err = launch_long<T, FF, 0>(st, taps, a, n, nlines, cu_count);
        ok = true;
    });
    return ok;
}
// The strided-axis pass (dim 2 of a matrix of R contiguous rows x C columns) with a long filter.
template <typename T>
bool long_axis_level(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t lds, T *dst, int64_t ldd,
                     int64_t R, int64_t C, int cu_count, hipError_t *err)
{
    constexpr int VEC = 16 / sizeof(T);
    *err = hipSuccess;
    if (!long_filter_ok(taps.F) || (C % (vlong_filter_ok(taps.F) ? 8 : 32)) != 0 || C < (vlong_filter_ok(taps.F) ? 16 : 32) || (R % VEC) != 0 || (lds % VEC) != 0 || (ldd % VEC) != 0 ||
        !a_al16(src) || !a_al16(dst))
        return false;
    if (vlong_only(taps.F) || (vlong_filter_ok(taps.F) && ((C % 32) != 0 || C < 32 || R < 512))) {
        *err = vl_axis<T>(st, taps, fw, src, lds, dst, ldd, R, C, cu_count);
        return true;
    }
    bool ok = false;
    WL_DISPATCH_FL(taps.F, {
        if (fw) *err = launch_axis<T, FF, 1>(st, taps, src, lds, 0, dst, ldd, 0, R, C, 1, cu_count);
        else *err = launch_axis<T, FF, 0>(st, taps, src, lds, 0, dst, ldd, 0, R, C, 1, cu_count);
        ok = true;
    });
    return ok;
}
#define WL_INST_LONG(T)                                                                                                          \
    template bool long_lines_fwd_level<T>(hipStream_t, const Taps<T> &, const T *, int64_t, T *, int64_t, T *, int64_t, int64_t, \
                                          int64_t, int, hipError_t *);                                                          \
    template bool long_lines_inv_level<T>(hipStream_t, const Taps<T> &, const T *, int64_t, const T *, int64_t, T *, int64_t,    \
                                          int64_t, int64_t, int, hipError_t *);                                                 \
    template bool long_axis_level<T>(hipStream_t, const Taps<T> &, int, const T *, int64_t, T *, int64_t, int64_t, int64_t, int, \
                                     hipError_t *);
WL_INST_LONG(float)
WL_INST_LONG(double)

template bool fast3d_fwd_level<float>(hipStream_t, const Taps<float> &, const float *, int64_t, int64_t, float *, int64_t, int64_t,
                                      float *, const int64_t[3], float *, float *, int, hipError_t *, const char **);
template bool fast3d_fwd_level<double>(hipStream_t, const Taps<double> &, const double *, int64_t, int64_t, double *, int64_t, int64_t,
                                       double *, const int64_t[3], double *, double *, int, hipError_t *, const char **);
template bool fast3d_inv_level<float>(hipStream_t, const Taps<float> &, const float *, int64_t, int64_t, const float *, float *,
                                      int64_t, int64_t, const int64_t[3], float *, float *, int, hipError_t *, const char **);
template bool fast3d_inv_level<double>(hipStream_t, const Taps<double> &, const double *, int64_t, int64_t, const double *, double *,
                                       int64_t, int64_t, const int64_t[3], double *, double *, int, hipError_t *, const char **);

}  // namespace wl
