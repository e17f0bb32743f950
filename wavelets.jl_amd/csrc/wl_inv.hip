// wl_inv.hip -- inverse filter-bank DWT: level loop and streaming kernels.
//
//   k_inv1d_stream     one inverse level of a line (also one line per blockIdx.y: batched columns, or
//                      every column of a 2-D block = the dim-1 pass of a 2-D inverse level).  Each lane
//                      holds 4 approximation + 4 detail coefficients (two 16-byte loads), takes the
//                      (F-2)/2 neighbouring coefficients from the adjacent lanes by DPP and stores 8
//                      reconstructed samples (two 16-byte stores).
//   k_inv2d_stream     one whole 2-D inverse level (dim-1 + dim-2 reconstruction) in a single pass over HBM:
//                      the default for 2-D blocks of >= 128 rows (see the comment at the kernel).
//   k_inv_dim2_stream  the dim-2 pass alone (fallback when the fused kernel's shape conditions do not hold): a
//                      wave owns 256 rows and marches along dim 2 with two 8-slot register rings.
//   k_tail_inv         the deepest levels (<= 4096 elements) inside one workgroup, coefficients staged to LDS once.
// Long filters and 3-D levels: wl_axis.hip.  Everything else: the generic kernels (wl_generic.hip).
// Arithmetic = the closed form of filtup! (wl_internal.h): x[o] = S + D with
//   S = sum over m descending, (o-m) even, of h[m]*s[(o-m)/2];  D = sum over m ascending, (o+m-1) even, of g[m]*d[(o+m-1)/2]
#include "wl_fast.h"
#include "wl_dev.h"


namespace wl {

template <typename T, int F>
struct TapsI { T h[F]; T g[F]; };
template <typename T, int F>
static TapsI<T, F> shrink_i(const Taps<T> &t)
{
    TapsI<T, F> r;
    for (int i = 0; i < F; ++i) { r.h[i] = t.h[i]; r.g[i] = t.g[i]; }
    return r;
}

__device__ __forceinline__ int i_dpp_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int i_dpp_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float i_next(float v) { return __int_as_float(i_dpp_next(__float_as_int(v))); }
__device__ __forceinline__ float i_prev(float v) { return __int_as_float(i_dpp_prev(__float_as_int(v))); }
__device__ __forceinline__ double i_next(double v) { return __hiloint2double(i_dpp_next(__double2hiint(v)), i_dpp_next(__double2loint(v))); }
__device__ __forceinline__ double i_prev(double v) { return __hiloint2double(i_dpp_prev(__double2hiint(v)), i_dpp_prev(__double2loint(v))); }

template <typename T, int N>
__device__ __forceinline__ void ldv(const T *p, T (&v)[N])
{
    constexpr int C = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t = *reinterpret_cast<const V *>(p + c * C);
#pragma unroll
        for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void stv(T *p, const T (&v)[N])
{
    constexpr int C = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        *reinterpret_cast<V *>(p + c * C) = t;
    }
}

// reconstruct the output pair (x[2p], x[2p+1]) from sw[0..SH] = s[p-SH..p] and dw[0..SH] = d[p..p+SH]
template <typename T, int F>
__device__ __forceinline__ void inv_pair(const T *sw, const T *dw, const TapsI<T, F> &tp, T &xe, T &xo)
{
    constexpr int SH = (F - 2) / 2;
    // even output o = 2p: S over even m descending (F-2, ..., 0) -> s[p - m/2]; D over odd m ascending -> d[p + (m-1)/2]
    T Se = tp.h[F - 2] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Se = Se + tp.h[F - 2 - 2 * q] * sw[q];
    T De = -tp.h[1] * dw[0];                    // g[m] = (-1)^m h[m] exactly (make_taps): only h occupies SGPRs
#pragma unroll
    for (int q = 1; q <= SH; ++q) De = De + -tp.h[1 + 2 * q] * dw[q];
    xe = Se + De;
    // odd output o = 2p+1: S over odd m descending (F-1, ..., 1) -> s[p - (m-1)/2]; D over even m ascending -> d[p + m/2]
    T So = tp.h[F - 1] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) So = So + tp.h[F - 1 - 2 * q] * sw[q];
    T Do = tp.h[0] * dw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Do = Do + tp.h[2 * q] * dw[q];
    xo = So + Do;
}

template <typename T, int N>
__device__ __forceinline__ void ldn(const T *p, T (&v)[N])
{
    constexpr int C = ((int)(16 / sizeof(T)) < N) ? (int)(16 / sizeof(T)) : N;
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t = *reinterpret_cast<const V *>(p + c * C);
#pragma unroll
        for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void stn(T *p, const T (&v)[N])
{
    constexpr int C = ((int)(16 / sizeof(T)) < N) ? (int)(16 / sizeof(T)) : N;
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        *reinterpret_cast<V *>(p + c * C) = t;
    }
}
template <typename T, int NB>
__device__ __forceinline__ T i_prev_n(T v)
{
#pragma unroll
    for (int i = 0; i < NB; ++i) v = i_prev(v);
    return v;
}
template <typename T, int NB>
__device__ __forceinline__ T i_next_n(T v)
{
#pragma unroll
    for (int i = 0; i < NB; ++i) v = i_next(v);
    return v;
}

// ---------------------------------------------------------------------------------------------------
template <typename T, int F>
struct Inv1DArgs {
    const T *ssrc; int64_t s_ls;
    const T *dsrc; int64_t d_ls;
    T *dst; int64_t o_ls;
    int64_t n;                      // output line length (multiple of 8, >= 512)
    int64_t ntiles;
    TapsI<T, F> tp;
};

template <typename T, int F>
__global__ void __launch_bounds__(256) k_inv1d_stream(Inv1DArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2, VP = 62 * 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t nx = a.n >> 1;
    const T *ssrc = a.ssrc + (int64_t)blockIdx.y * a.s_ls;
    const T *dsrc = a.dsrc + (int64_t)blockIdx.y * a.d_ls;
    T *dst = a.dst + (int64_t)blockIdx.y * a.o_ls;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k0 = tile * VP + (int64_t)(lane - 1) * 4;
        int64_t kw = k0;
        if (kw < 0) kw += nx;
        if (kw >= nx) kw -= nx;
        T s[4], d[4];
        ldv<T, 4>(ssrc + kw, s);
        ldv<T, 4>(dsrc + kw, d);
        // sx[i] = s[p = i - SH] for i = 0 .. 3 + SH;  dx[i] = d[p = i] for i = 0 .. 3 + SH
        T sx[4 + SH], dx[4 + SH];
#pragma unroll
        for (int i = 0; i < 4; ++i) { sx[SH + i] = s[i]; dx[i] = d[i]; }
#pragma unroll
        for (int i = 0; i < SH; ++i) {
            sx[i] = i_prev(s[4 - SH + i]);        // previous lane's last SH approximations
            dx[4 + i] = i_next(d[i]);             // next lane's first SH details
        }
        T out[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) inv_pair<T, F>(&sx[p], &dx[p], a.tp, out[2 * p], out[2 * p + 1]);
        if (lane >= 1 && lane <= 62 && k0 < nx) stv<T, 8>(dst + 2 * k0, out);
    }
}

// ---------------------------------------------------------------------------------------------------
template <typename T, int F>
struct InvD2Args {
    const T *src; int64_t lds;      // ms x ns block, columns [0,ns/2) = approximation, [ns/2,ns) = detail (along dim 2)
    T *dst; int64_t ldd;
    int64_t ms, ns;
    int TP;                         // output column pairs per chunk (multiple of 8)
    int nstrips, nchunks;
    TapsI<T, F> tp;
};

template <typename T, int F, int RPL>
__global__ void __launch_bounds__(64) k_inv_dim2_stream(InvD2Args<T, F> a)
{
    constexpr int SH = (F - 2) / 2, R = 8;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t logical = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);
    const int64_t row = (int64_t)strip * 64 * RPL + (int64_t)lane * RPL;
    const bool valid = row < a.ms;
    const int64_t rr = valid ? row : 0;
    const int64_t nxj = a.ns >> 1;
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < nxj) ? (p0 + a.TP) : nxj;
    const int S = (int)(pend - p0);              // multiple of 8
    const T *sbase = a.src + rr;
    const T *dbase = a.src + rr + nxj * a.lds;
    // ring slot c % R holds approximation column (p0 - SH + c) and detail column (p0 + c)
    T rs[R][RPL], rd[R][RPL];
#pragma unroll
    for (int c = 0; c < R - 1; ++c) {
        int64_t js = p0 - SH + c;
        if (js < 0) js += nxj;
        if (js >= nxj) js -= nxj;
        int64_t jd = p0 + c;
        if (jd >= nxj) jd -= nxj;
        ldv<T, RPL>(sbase + js * a.lds, rs[c]);
        ldv<T, RPL>(dbase + jd * a.lds, rd[c]);
    }
    T *out = a.dst + rr;
    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {
            int64_t js = p0 - SH + t + (R - 1);
            if (js >= nxj) js -= nxj;
            int64_t jd = p0 + t + (R - 1);
            if (jd >= nxj) jd -= nxj;
            ldv<T, RPL>(sbase + js * a.lds, rs[(u + R - 1) % R]);
            ldv<T, RPL>(dbase + jd * a.lds, rd[(u + R - 1) % R]);
        }
        T xe[RPL], xo[RPL];
#pragma unroll
        for (int q = 0; q < RPL; ++q) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int i = 0; i <= SH; ++i) { sw[i] = rs[(u + i) % R][q]; dw[i] = rd[(u + i) % R][q]; }
            inv_pair<T, F>(sw, dw, a.tp, xe[q], xo[q]);
        }
        if (valid) {
            const int64_t p = p0 + t;
            stv<T, RPL>(out + (2 * p) * a.ldd, xe);
            stv<T, RPL>(out + (2 * p + 1) * a.ldd, xo);
        }
    };
    int t0 = 0;
    for (; t0 < S - R; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u, true);
    }
#pragma unroll
    for (int u = 0; u < R; ++u) step(t0 + u, u, u <= SH);     // last group: columns beyond the chunk's window are not fetched
}


// ---------------------------------------------------------------------------------------------------
// LDS tail for the inverse: the deepest levels (output block <= 16 Ki f32 / 8 Ki f64 elements, 1-D line
// per workgroup or one 2-D block) reconstructed inside one workgroup.  The running reconstruction stays
// in LDS (fixed leading dimension, so the next level only has to stage its detail quadrants around it);
// true periodic indexing, any filter length (FC = 0: run-time length, odd lengths included).
template <typename T>
struct TailInvArgs {
    const T *x; int64_t ldx; int64_t x_item;     // coefficient array (details + deepest approximation)
    T *out; int64_t ldo; int64_t out_item;       // reconstruction of the last level done here
    int n0, n1;                                  // OUTPUT extents of the last level done here (n1 == 1: line)
    int nt, nlev, cap, ld;
};

__device__ __forceinline__ int iw(int i, int n)
{
    while (i >= n) i -= n;
    while (i < 0) i += n;
    return i;
}
// x[o] = S + D from s[0..nx) (stride ss) and d[0..nx) (stride sd), wl_internal.h closed form
template <typename T, int FC>
__device__ __forceinline__ T tail_inv_one(const T *sp, int ss, const T *dp, int sd, int o, int nx, const Taps<T> &tp)
{
    const int F = (FC > 0) ? FC : tp.F;
    T S = (T)0, D = (T)0;
    bool first = true;
    if constexpr (FC > 0) {
#pragma unroll
        for (int m = FC - 1; m >= 0; --m)
            if (((o - m) & 1) == 0) {
                T term = tp.h[m] * sp[iw((o - m) / 2, nx) * ss];
                S = first ? term : (S + term);
                first = false;
            }
        first = true;
#pragma unroll
        for (int m = 0; m < FC; ++m)
            if (((o + m - 1) & 1) == 0) {
                T term = tp.g[m] * dp[iw((o + m - 1) / 2, nx) * sd];
                D = first ? term : (D + term);
                first = false;
            }
    } else {
        // run-time length: every second tap, the coefficient index going up by one per term (one periodic reduction for the
        // start, a conditional wrap afterwards), blocks of 8 terms with their LDS reads issued before the arithmetic
        {
            const int mt = (((F - 1 - o) & 1) == 0) ? F - 1 : F - 2;       // largest tap with (o - m) even
            if (mt >= 0) {
                int k = iw((o - mt) / 2, nx);
                S = tp.h[mt] * sp[k * ss];
                for (int m0 = mt - 2; m0 >= 0; m0 -= 16) {
                    T xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 - 2 * e >= 0) {
                            if (++k >= nx) k = 0;
                            xv[e] = sp[k * ss];
                        }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 - 2 * e >= 0) S = S + tp.h[m0 - 2 * e] * xv[e];
                }
            }
        }
        {
            const int mb = (o & 1) ? 0 : 1;                                 // smallest tap with (o + m - 1) even
            if (mb < F) {
                int k = iw((o + mb - 1) / 2, nx);
                D = tp.g[mb] * dp[k * sd];
                for (int m0 = mb + 2; m0 < F; m0 += 16) {
                    T xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 + 2 * e < F) {
                            if (++k >= nx) k = 0;
                            xv[e] = dp[k * sd];
                        }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 + 2 * e < F) D = D + tp.g[m0 + 2 * e] * xv[e];
                }
            }
        }
        (void)first;
    }
    return S + D;
}

// the output pair (x[2p], x[2p+1]) for an even compile-time filter length, nx >= (FC-2)/2 (one wrap at most)
template <typename T, int FC>
__device__ __forceinline__ void tail_inv_pair(const T *sp, int ss, const T *dp, int sd, int p, int nx, const Taps<T> &tp, T &xe, T &xo)
{
    constexpr int SH = (FC - 2) / 2;
    T sw[SH + 1], dw[SH + 1];
#pragma unroll
    for (int q = 0; q <= SH; ++q) {
        int is = p - SH + q;
        if (is < 0) is += nx;
        int id = p + q;
        if (id >= nx) id -= nx;
        sw[q] = sp[is * ss];
        dw[q] = dp[id * sd];
    }
    T Se = tp.h[FC - 2] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Se = Se + tp.h[FC - 2 - 2 * q] * sw[q];
    T De = -tp.h[1] * dw[0];                    // g[m] = (-1)^m h[m] exactly (make_taps): only h occupies SGPRs
#pragma unroll
    for (int q = 1; q <= SH; ++q) De = De + -tp.h[1 + 2 * q] * dw[q];
    xe = Se + De;
    T So = tp.h[FC - 1] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) So = So + tp.h[FC - 1 - 2 * q] * sw[q];
    T Do = tp.h[0] * dw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Do = Do + tp.h[2 * q] * dw[q];
    xo = So + Do;
}
// e -> (e / d, e % d) without the emulated division when d is a power of two
__device__ __forceinline__ void split_idx(int e, int d, int &q, int &r)
{
    if ((d & (d - 1)) == 0) { const int lg = 31 - __clz(d); q = e >> lg; r = e & (d - 1); }
    else { q = e / d; r = e - q * d; }
}

template <typename T, int FC>
__global__ void __launch_bounds__(1024) k_tail_inv(TailInvArgs<T> a, Taps<T> tp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);
    T *Q = P + a.cap;
    const int tid = threadIdx.x, nthr = blockDim.x;
    const T *x = a.x + (int64_t)blockIdx.x * a.x_item;
    T *out = a.out + (int64_t)blockIdx.x * a.out_item;
    const int ld = a.ld;
    // The coefficients of every level done here lie inside the n0 (x n1) corner of x: stage that corner once
    // (one global-memory latency for the whole tail), then run the levels out of LDS.
    if (a.nt == 1) {
        for (int k = tid; k < a.n0; k += nthr) P[k] = x[k];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int n = a.n0 >> (a.nlev - 1);                       // output length of the deepest level
        // details always at P[nx..n); the running approximation starts at P[0..nx) and then ping-pongs between
        // Q and Q + n0/2 (never written where another thread may still be reading)
        const T *sp = P;
        T *q0 = Q, *q1 = Q + (a.n0 >> 1);
        for (int lev = 0; lev < a.nlev; ++lev) {
            const int nx = n >> 1;
            const bool last = (lev == a.nlev - 1);
            bool paired = false;
            if constexpr (FC > 0) {
                if (nx >= (FC - 2) / 2) {
                    paired = true;
                    for (int p = tid; p < nx; p += nthr) {
                        T xe, xo;
                        tail_inv_pair<T, FC>(sp, 1, P + nx, 1, p, nx, tp, xe, xo);
                        if (last) { out[2 * p] = xe; out[2 * p + 1] = xo; }
                        else { q0[2 * p] = xe; q0[2 * p + 1] = xo; }
                    }
                }
            }
            if (!paired) {
                for (int o = tid; o < n; o += nthr) {
                    T v = tail_inv_one<T, FC>(sp, 1, P + nx, 1, o, nx, tp);
                    if (last) out[o] = v;
                    else q0[o] = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            sp = q0;
            T *t = q0; q0 = q1; q1 = t;
            n <<= 1;
        }
    } else {
        for (int e = tid; e < a.n0 * a.n1; e += nthr) {
            int j, i;
            split_idx(e, a.n0, j, i);
            P[i + j * ld] = x[i + (int64_t)j * a.ldx];
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int m0 = a.n0 >> (a.nlev - 1), m1 = a.n1 >> (a.nlev - 1);
        for (int lev = 0; lev < a.nlev; ++lev) {
            const int h0 = m0 >> 1, h1 = m1 >> 1;
            const bool last = (lev == a.nlev - 1);
            bool paired = false;
            if constexpr (FC > 0) {
                if (h0 >= (FC - 2) / 2 && h1 >= (FC - 2) / 2) {
                    paired = true;
                    // dim-1 pass (columns): Q[:, j] from P[0..h0, j] (s) and P[h0..m0, j] (d), one output pair per item
                    for (int e = tid; e < h0 * m1; e += nthr) {
                        int j, p;
                        split_idx(e, h0, j, p);
                        T xe, xo;
                        tail_inv_pair<T, FC>(P + j * ld, 1, P + j * ld + h0, 1, p, h0, tp, xe, xo);
                        Q[2 * p + j * ld] = xe;
                        Q[2 * p + 1 + j * ld] = xo;
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    // dim-2 pass (rows): columns 2p, 2p+1 from Q[i, 0..h1) (s) and Q[i, h1..m1) (d); overwrites the m0 x m1 corner of P
                    for (int e = tid; e < m0 * h1; e += nthr) {
                        int p, i;
                        split_idx(e, m0, p, i);
                        T xe, xo;
                        tail_inv_pair<T, FC>(Q + i, ld, Q + i + h1 * ld, ld, p, h1, tp, xe, xo);
                        if (last) { out[i + (int64_t)(2 * p) * a.ldo] = xe; out[i + (int64_t)(2 * p + 1) * a.ldo] = xo; }
                        else { P[i + (2 * p) * ld] = xe; P[i + (2 * p + 1) * ld] = xo; }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            }
            if (!paired) {
                // dim-1 pass (columns): Q[:, j] from P[0..h0, j] (s) and P[h0..m0, j] (d)
                for (int e = tid; e < m0 * m1; e += nthr) {
                    int j, o;
                    split_idx(e, m0, j, o);
                    Q[o + j * ld] = tail_inv_one<T, FC>(P + j * ld, 1, P + j * ld + h0, 1, o, h0, tp);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                // dim-2 pass (rows): result[i, o] from Q[i, 0..h1) (s) and Q[i, h1..m1) (d); overwrites the m0 x m1 corner of P
                for (int e = tid; e < m0 * m1; e += nthr) {
                    int o, i;
                    split_idx(e, m0, o, i);
                    T v = tail_inv_one<T, FC>(Q + i, ld, Q + i + h1 * ld, ld, o, h1, tp);
                    if (last) out[i + (int64_t)o * a.ldo] = v;
                    else P[i + o * ld] = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            m0 <<= 1;
            m1 <<= 1;
        }
    }
}

template <typename T>
constexpr int inv_tail_cap() { return sizeof(T) == 4 ? 16384 : 8192; }

template <typename T>
static hipError_t launch_tail_inv(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, int64_t x_item,
                                  T *out, int64_t ldo, int64_t out_item, int nitems, int n0, int n1, int nt, int nlev)
{
    TailInvArgs<T> a;
    a.x = x; a.ldx = ldx; a.x_item = x_item; a.out = out; a.ldo = ldo; a.out_item = out_item;
    a.n0 = n0; a.n1 = n1; a.nt = nt; a.nlev = nlev;
    a.ld = (nt == 2) ? (n0 | 1) : n0;
    a.cap = (int)((((int64_t)a.ld * n1) + 15) & ~15);
    size_t shmem = 2 * (size_t)a.cap * sizeof(T);
    if (nt == 1) {      // P: the staged line (n0), Q: two ping-pong approximation buffers of n0/2
        a.cap = (int)(((int64_t)n0 + 15) & ~15);
        shmem = ((size_t)a.cap + (size_t)n0 + 16) * sizeof(T);
    }
    const int64_t work = (int64_t)n0 * n1;
    const int threads = work >= 4096 ? 1024 : (work >= 512 ? 256 : 64);
#define WL_TI(FC_)                                                                                             \
    do {                                                                                                       \
        static unsigned char attr_set[64] = {0};                                                               \
        int dev = 0;                                                                                           \
        (void)hipGetDevice(&dev);                                                                              \
        dev &= 63;                                                                                             \
        if (!attr_set[dev]) {                                                                                  \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail_inv<T, FC_>),            \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
            if (e != hipSuccess) return e;                                                                     \
            attr_set[dev] = 1;                                                                                 \
        }                                                                                                      \
        hipLaunchKernelGGL((k_tail_inv<T, FC_>), dim3((unsigned)nitems), dim3(threads), shmem, st, a, taps);   \
    } while (0)
    switch (taps.F) {
    case 2: WL_TI(2); break;
    case 4: WL_TI(4); break;
    case 6: WL_TI(6); break;
    case 8: WL_TI(8); break;
    case 10: WL_TI(10); break;
    default: WL_TI(0); break;
    }
#undef WL_TI
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
static inline int i_env(const char *name, int dflt) { return (int)opt(name, dflt); }   // per-context options
static inline bool i_al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T, int F>
static hipError_t launch_inv1d(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls,
                               T *dst, int64_t o_ls, int64_t n, int64_t nlines, int cu_count)
{
    Inv1DArgs<T, F> a;
    a.ssrc = ssrc; a.s_ls = s_ls; a.dsrc = dsrc; a.d_ls = d_ls; a.dst = dst; a.o_ls = o_ls; a.n = n;
    a.ntiles = ((n >> 1) + 247) / 248;
    a.tp = shrink_i<T, F>(taps);
    int64_t gx = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 8;
    if (gx > cap) gx = cap;
    const int64_t slab = (i_env("WL_SLAB_LINES", 32768) > 0) ? i_env("WL_SLAB_LINES", 32768) : 32768;
    for (int64_t l0 = 0; l0 < nlines; l0 += slab) {      // gridDim.y <= 65535
        const int64_t nl = (nlines - l0 < slab) ? (nlines - l0) : slab;
        Inv1DArgs<T, F> b = a;
        b.ssrc = a.ssrc + l0 * a.s_ls; b.dsrc = a.dsrc + l0 * a.d_ls; b.dst = a.dst + l0 * a.o_ls;
        hipLaunchKernelGGL((k_inv1d_stream<T, F>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// TWO inverse levels of a line per launch: the level-l reconstruction (4 samples per lane) stays in registers and
// is consumed, together with the level-(l-1) details, by the second reconstruction (8 samples per lane).  Traffic
// for the two levels: read n, write n (level by level: 3n) and half the launches.  Halos by DPP lane shifts:
// ceil(SH/2) lanes for level l, ceil(SH/4) more on the approximation side for level l-1.
template <typename T, int F>
struct Inv1D2Args {
    const T *s2; int64_t s2_ls;     // approximation of the deeper level (n/4 per line)
    const T *d2; int64_t d2_ls;     // details of the deeper level (n/4)
    const T *d1; int64_t d1_ls;     // details of the shallower level (n/2)
    T *dst; int64_t o_ls;           // output lines (n)
    int64_t n;                      // OUTPUT line length (multiple of 16, >= 1024)
    int64_t ntiles;
    TapsI<T, F> tp;
};

template <typename T, int F>
__global__ void __launch_bounds__(256) k_inv1d_stream2(Inv1D2Args<T, F> a)
{
    constexpr int SH = (F - 2) / 2;
    constexpr int H2 = (SH + 1) / 2, H1 = (SH + 3) / 4;
    constexpr int HL = H2 + H1, HR = H2 > H1 ? H2 : H1;
    constexpr int VP2 = (64 - HL - HR) * 2;          // deeper-level pairs owned by a wave tile
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t nx2 = a.n >> 2;
    const T *s2 = a.s2 + (int64_t)blockIdx.y * a.s2_ls;
    const T *d2 = a.d2 + (int64_t)blockIdx.y * a.d2_ls;
    const T *d1 = a.d1 + (int64_t)blockIdx.y * a.d1_ls;
    T *dst = a.dst + (int64_t)blockIdx.y * a.o_ls;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k2 = tile * VP2 + (int64_t)(lane - HL) * 2;
        int64_t kw = k2;
        if (kw < 0) kw += nx2;
        if (kw >= nx2) kw -= nx2;
        T sv[2], dv[2], d1v[4];
        ldn<T, 2>(s2 + kw, sv);
        ldg_pol<WL_P_I1D2_LD != 0, T, 2>(d2 + kw, dv);
        ldg_pol<WL_P_I1D2_LD != 0, T, 4>(d1 + 2 * kw, d1v);
        // deeper level: pairs kw, kw+1 -> a1[0..3]
        T sx2[2 + SH], dx2[2 + SH];
#pragma unroll
        for (int i = 0; i < 2; ++i) { sx2[SH + i] = sv[i]; dx2[i] = dv[i]; }
        if constexpr (SH >= 1) { sx2[SH - 1] = i_prev_n<T, 1>(sv[1]); dx2[2] = i_next_n<T, 1>(dv[0]); }
        if constexpr (SH >= 2) { sx2[SH - 2] = i_prev_n<T, 1>(sv[0]); dx2[3] = i_next_n<T, 1>(dv[1]); }
        if constexpr (SH >= 3) { sx2[SH - 3] = i_prev_n<T, 2>(sv[1]); dx2[4] = i_next_n<T, 2>(dv[0]); }
        if constexpr (SH >= 4) { sx2[SH - 4] = i_prev_n<T, 2>(sv[0]); dx2[5] = i_next_n<T, 2>(dv[1]); }
        T a1[4];
#pragma unroll
        for (int p = 0; p < 2; ++p) inv_pair<T, F>(&sx2[p], &dx2[p], a.tp, a1[2 * p], a1[2 * p + 1]);
        // shallower level: pairs 2kw .. 2kw+3 -> 8 samples
        T sx1[4 + SH], dx1[4 + SH];
#pragma unroll
        for (int i = 0; i < 4; ++i) { sx1[SH + i] = a1[i]; dx1[i] = d1v[i]; }
        if constexpr (SH >= 1) { sx1[SH - 1] = i_prev_n<T, 1>(a1[3]); dx1[4] = i_next_n<T, 1>(d1v[0]); }
        if constexpr (SH >= 2) { sx1[SH - 2] = i_prev_n<T, 1>(a1[2]); dx1[5] = i_next_n<T, 1>(d1v[1]); }
        if constexpr (SH >= 3) { sx1[SH - 3] = i_prev_n<T, 1>(a1[1]); dx1[6] = i_next_n<T, 1>(d1v[2]); }
        if constexpr (SH >= 4) { sx1[SH - 4] = i_prev_n<T, 1>(a1[0]); dx1[7] = i_next_n<T, 1>(d1v[3]); }
        T out[8];
#pragma unroll
        for (int p = 0; p < 4; ++p) inv_pair<T, F>(&sx1[p], &dx1[p], a.tp, out[2 * p], out[2 * p + 1]);
        if (lane >= HL && lane < 64 - HR && k2 < nx2) stg_pol<WL_P_I1D2_ST != 0, T, 8>(dst + 4 * k2, out);
    }
}

template <typename T, int F>
static hipError_t launch_inv1d2(hipStream_t st, const Taps<T> &taps, const T *s2, int64_t s2_ls, const T *d2, int64_t d2_ls,
                                const T *d1, int64_t d1_ls, T *dst, int64_t o_ls, int64_t n, int64_t nlines, int cu_count)
{
    constexpr int SH = (F - 2) / 2, H2 = (SH + 1) / 2, H1 = (SH + 3) / 4, HL = H2 + H1, HR = H2 > H1 ? H2 : H1;
    constexpr int VP2 = (64 - HL - HR) * 2;
    Inv1D2Args<T, F> a;
    a.s2 = s2; a.s2_ls = s2_ls; a.d2 = d2; a.d2_ls = d2_ls; a.d1 = d1; a.d1_ls = d1_ls; a.dst = dst; a.o_ls = o_ls; a.n = n;
    a.ntiles = ((n >> 2) + VP2 - 1) / VP2;
    a.tp = shrink_i<T, F>(taps);
    int64_t gx = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 8;
    if (gx > cap) gx = cap;
    const int64_t slab = (i_env("WL_SLAB_LINES", 32768) > 0) ? i_env("WL_SLAB_LINES", 32768) : 32768;
    for (int64_t l0 = 0; l0 < nlines; l0 += slab) {      // gridDim.y <= 65535
        const int64_t nl = (nlines - l0 < slab) ? (nlines - l0) : slab;
        Inv1D2Args<T, F> b = a;
        b.s2 = a.s2 + l0 * a.s2_ls; b.d2 = a.d2 + l0 * a.d2_ls; b.d1 = a.d1 + l0 * a.d1_ls; b.dst = a.dst + l0 * a.o_ls;
        hipLaunchKernelGGL((k_inv1d_stream2<T, F>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
    }
    return hipGetLastError();
}

template <typename T, int F>
static hipError_t launch_inv_dim2(hipStream_t st, const Taps<T> &taps, const T *src, int64_t lds, T *dst, int64_t ldd,
                                  int64_t ms, int64_t ns, int cu_count)
{
    constexpr int RPL = 16 / sizeof(T);
    InvD2Args<T, F> a;
    a.src = src; a.lds = lds; a.dst = dst; a.ldd = ldd; a.ms = ms; a.ns = ns;
    a.nstrips = (int)((ms + 64 * RPL - 1) / (64 * RPL));
    const int64_t nxj = ns >> 1;
    int TP = 64;
    while (TP > 8 && (int64_t)a.nstrips * ((nxj + TP - 1) / TP) < (int64_t)cu_count * 8) TP >>= 1;
    a.TP = TP;
    a.nchunks = (int)((nxj + TP - 1) / TP);
    a.tp = shrink_i<T, F>(taps);
    hipLaunchKernelGGL((k_inv_dim2_stream<T, F, RPL>), dim3((unsigned)(a.nstrips * a.nchunks)), dim3(64), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// One fused 2-D inverse level (columns = dim-1 pass, then rows = dim-2 pass, transforms_filter.jl:231-243)
// in a single pass over HBM.  Lanes run along dim 1 (PPL coefficient pairs = 2*PPL output rows per lane,
// halo lanes on both sides of the wave refetch their neighbours' coefficients); the wave marches along the
// output column pairs p.  Per step it loads the raw coefficient column p of the left half and column
// p + SH of the right half (prefetched 4 steps ahead in a static register ring), reconstructs both along
// dim 1 (DPP neighbour exchange) into two 4-slot rings of dim-1-reconstructed columns, and combines
// columns p-SH..p / p..p+SH of those rings into output columns 2p, 2p+1.
// halo lanes on either side of a strip.  With 2 or 4 coefficient pairs per lane the count is rounded up so that the strip pitch
// (2 * VP output rows) is a multiple of 32 rows = 128 bytes of Float32: strips that start in the middle of a line cost ~7 % in a
// same-box A/B of the lifting kernels, and the first version's 240-row pitch was exactly that case.
constexpr int inv2d_halo_lanes(int SH, int PPL)
{
    int hl = (SH + PPL - 1) / PPL;
    if (PPL >= 2) { const int q = 16 / (2 * PPL); while (((64 - 2 * hl) % (2 * q)) != 0) ++hl; }
    return hl;
}
template <typename T, int F>
struct Inv2DArgs {
    const T *x; int64_t ldx;        // coefficient array (details; approximation quadrant too when ll == nullptr)
    const T *ll; int64_t ldl;       // deeper reconstruction, h0 x h1 (or nullptr)
    T *dst; int64_t ldd;
    int64_t n0, n1;                 // OUTPUT block extents
    int TP;                         // output column pairs per chunk (multiple of 4)
    int nstrips, nchunks;
    // batch of independent blocks (blockIdx.y; the planes of a 3-D level): element strides; only the first nll
    // blocks take their approximation quadrant from ll
    int64_t bs_x, bs_ll, bs_dst; int nll;
    TapsI<T, F> tp;
};

// dim-1 reconstruction of one column: the lane's PPL approximation / detail coefficients -> 2*PPL samples
template <typename T, int F, int PPL>
__device__ __forceinline__ void inv_column(const T (&s)[PPL], const T (&d)[PPL], const TapsI<T, F> &tp, T (&out)[2 * PPL])
{
    constexpr int SH = (F - 2) / 2;
    T sx[PPL + SH], dx[PPL + SH];
#pragma unroll
    for (int i = 0; i < PPL; ++i) { sx[SH + i] = s[i]; dx[i] = d[i]; }
    if constexpr (SH >= 1) { constexpr int rel = 1, nb = (rel + PPL - 1) / PPL; sx[SH - 1] = i_prev_n<T, nb>(s[nb * PPL - rel]); }
    if constexpr (SH >= 2) { constexpr int rel = 2, nb = (rel + PPL - 1) / PPL; sx[SH - 2] = i_prev_n<T, nb>(s[nb * PPL - rel]); }
    if constexpr (SH >= 3) { constexpr int rel = 3, nb = (rel + PPL - 1) / PPL; sx[SH - 3] = i_prev_n<T, nb>(s[nb * PPL - rel]); }
    if constexpr (SH >= 4) { constexpr int rel = 4, nb = (rel + PPL - 1) / PPL; sx[SH - 4] = i_prev_n<T, nb>(s[nb * PPL - rel]); }
    if constexpr (SH >= 1) { constexpr int rel = PPL + 0, nb = rel / PPL; dx[PPL + 0] = i_next_n<T, nb>(d[rel - nb * PPL]); }
    if constexpr (SH >= 2) { constexpr int rel = PPL + 1, nb = rel / PPL; dx[PPL + 1] = i_next_n<T, nb>(d[rel - nb * PPL]); }
    if constexpr (SH >= 3) { constexpr int rel = PPL + 2, nb = rel / PPL; dx[PPL + 2] = i_next_n<T, nb>(d[rel - nb * PPL]); }
    if constexpr (SH >= 4) { constexpr int rel = PPL + 3, nb = rel / PPL; dx[PPL + 3] = i_next_n<T, nb>(d[rel - nb * PPL]); }
#pragma unroll
    for (int p = 0; p < PPL; ++p) inv_pair<T, F>(&sx[p], &dx[p], tp, out[2 * p], out[2 * p + 1]);
}

// (10 taps, Float32: left to itself the compiler takes 175 VGPRs = 2 waves per SIMD; this kernel is latency-bound per wave, so the
//  third resident wave is worth the few rematerialised values)
template <typename T, int F, int PPL>
__global__ void __launch_bounds__(64, (F == 10 && sizeof(T) == 4 && PPL == 2) ? 3 : 1) k_inv2d_stream(Inv2DArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2, HL = inv2d_halo_lanes(SH, PPL), VP = (64 - 2 * HL) * PPL, R = (SH <= 3) ? 4 : 8;
    static_assert(SH <= 4, "ring depth");
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t logical = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);
    const int64_t h0 = a.n0 >> 1, h1 = a.n1 >> 1;
    const int64_t k0 = (int64_t)strip * VP + (int64_t)(lane - HL) * PPL;
    int64_t kw = k0;
    if (kw < 0) kw += h0;
    if (kw >= h0) kw -= h0;
    const bool store = lane >= HL && lane < 64 - HL && k0 < h0;
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < h1) ? (p0 + a.TP) : h1;
    const int S = (int)(pend - p0);                  // multiple of 4
    // left-half columns: approximation rows from ll (when given), detail rows from x; right half: both from x
    const T *xb = a.x + (int64_t)blockIdx.y * a.bs_x;
    const bool from_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    const T *ls_base = (from_ll ? a.ll + (int64_t)blockIdx.y * a.bs_ll : xb) + kw;
    const int64_t ls_ld = from_ll ? a.ldl : a.ldx;
    const T *ld_base = xb + h0 + kw;
    const T *rs_base = xb + h1 * a.ldx + kw;
    const T *rd_base = rs_base + h0;
    T rLs[R][PPL], rLd[R][PPL], rRs[R][PPL], rRd[R][PPL];      // raw columns in flight
    T iS[R][2 * PPL], iD[R][2 * PPL];                           // dim-1-reconstructed columns
    auto load_raw = [&](const int64_t t, const int slot) __attribute__((always_inline)) {
        int64_t js = p0 + t;
        if (js < 0) js += h1;
        int64_t jd = p0 + t + SH;
        if (jd >= h1) jd -= h1;
        ldg_pol<WL_P_I2DS_LD != 0, T, PPL>(ls_base + js * ls_ld, rLs[slot]);
        ldg_pol<WL_P_I2DS_LD != 0, T, PPL>(ld_base + js * a.ldx, rLd[slot]);
        ldg_pol<WL_P_I2DS_LD != 0, T, PPL>(rs_base + jd * a.ldx, rRs[slot]);
        ldg_pol<WL_P_I2DS_LD != 0, T, PPL>(rd_base + jd * a.ldx, rRd[slot]);
    };
    // prologue: steps -SH .. -1 fill the reconstructed rings; their raw columns pass through slots (c - SH) mod 4
#pragma unroll
    for (int c = 0; c < SH; ++c) load_raw(c - SH, (c - SH + R) % R);
#pragma unroll
    for (int c = 0; c < SH; ++c) {
        constexpr int dummy = 0; (void)dummy;
        inv_column<T, F, PPL>(rLs[(c - SH + R) % R], rLd[(c - SH + R) % R], a.tp, iS[(c - SH + R) % R]);
        inv_column<T, F, PPL>(rRs[(c - SH + R) % R], rRd[(c - SH + R) % R], a.tp, iD[(c - SH + R) % R]);
    }
#pragma unroll
    for (int c = 0; c < R; ++c) load_raw(c, c);
    T *out = a.dst + (int64_t)blockIdx.y * a.bs_dst + 2 * k0;
    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        inv_column<T, F, PPL>(rLs[u], rLd[u], a.tp, iS[u]);
        inv_column<T, F, PPL>(rRs[u], rRd[u], a.tp, iD[u]);
        if (prefetch) load_raw(t + R, u);
        T xe[2 * PPL], xo[2 * PPL];
#pragma unroll
        for (int q = 0; q < 2 * PPL; ++q) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int i = 0; i <= SH; ++i) { sw[i] = iS[(u + i + R - SH) % R][q]; dw[i] = iD[(u + i + R - SH) % R][q]; }
            inv_pair<T, F>(sw, dw, a.tp, xe[q], xo[q]);
        }
        if (store) {
            const int64_t p = p0 + t;
            stg_pol<WL_P_I2DS_ST != 0, T, 2 * PPL>(out + (2 * p) * a.ldd, xe);
            stg_pol<WL_P_I2DS_ST != 0, T, 2 * PPL>(out + (2 * p + 1) * a.ldd, xo);
        }
    };
    int t0 = 0;
    for (; t0 < S - R; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u, true);
    }
#pragma unroll
    for (int u = 0; u < R; ++u) step(t0 + u, u, false);
}

template <typename T, int F, int PPL>
static hipError_t launch_inv2d(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl,
                               T *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count,
                               int64_t nbatch = 1, int64_t bs_x = 0, int64_t bs_ll = 0, int64_t bs_dst = 0, int nll = 1)
{
    constexpr int SH = (F - 2) / 2, HL = inv2d_halo_lanes(SH, PPL), VP = (64 - 2 * HL) * PPL;
    Inv2DArgs<T, F> a;
    a.x = x; a.ldx = ldx; a.ll = ll; a.ldl = ldl; a.dst = dst; a.ldd = ldd; a.n0 = n0; a.n1 = n1;
    a.bs_x = bs_x; a.bs_ll = bs_ll; a.bs_dst = bs_dst; a.nll = nll;
    const int64_t h0 = n0 >> 1, h1 = n1 >> 1;
    a.nstrips = (int)((h0 + VP - 1) / VP);
    int TP = i_env("WL_INV2D_TP", 64);
    if (TP < 8 || (TP % 8) != 0) TP = 64;                 // (a test knob must not be able to divide by zero)
    const int64_t wpc = (i_env("WL_INV2D_WAVES", 16) > 0) ? i_env("WL_INV2D_WAVES", 16) : 16;
    while (TP > 8 && (int64_t)a.nstrips * ((h1 + TP - 1) / TP) * nbatch < (int64_t)cu_count * wpc) TP >>= 1;
    a.TP = TP;
    a.nchunks = (int)((h1 + TP - 1) / TP);
    a.tp = shrink_i<T, F>(taps);
    hipLaunchKernelGGL((k_inv2d_stream<T, F, PPL>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)nbatch), dim3(64), 0, st, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// TWO fused 2-D inverse levels per launch (levels l+1 and l of a big Float32 block; the mirror of k_fwd2d_pair): the level-(l+1)
// reconstruction -- the approximation quadrant of level l -- never goes to HBM.  A workgroup of three waves owns 4 VP = 448
// output rows of level l and a chunk of TP output column pairs:
//   * wave 0 runs the body of k_inv2d_stream on level l+1 for the 240 (VP + 128) approximation rows the other two waves need
//     (two halo lanes per side: exactly the reach of the filter) and parks its output columns -- two per step -- in an 8-slot
//     LDS column ring instead of storing them;
//   * waves 1 and 2 run that body on level l for two adjacent strips of 2 VP rows; the scaling rows of their left-half columns
//     come from the ring (one ds_read_b64 per step), the three detail quadrants from HBM as before.
// Wave 0 makes one step per TWO barriers (dim-1 reconstruction of its raw columns | dim-2 combination + ring write), waves 1-2
// one step per barrier, two barriers behind: ring column c is written before barrier 2 floor(c/2) + 1 and read after barrier
// c + 2 - delta (delta = SH mod 2: the first ring column is the even column at or below p0 - SH).
// Arithmetic: inv_column / inv_pair of the single-level kernel -- bit-identical to two launches of it.
template <int F>
struct InvPairArgs {
    const float *x; int64_t ldx;        // coefficient array
    const float *ll; int64_t ldl;       // reconstruction of level l+2 = approximation source of level l+1 (nullptr: x)
    float *dst; int64_t ldd;            // level-l output, n0 x n1
    int64_t n0, n1;
    int TP;                             // level-l output column pairs per chunk (multiple of 8)
    int nstrips, nchunks;
    int prio;
    TapsI<float, F> tp;
};

template <int F, int MW>
__global__ void __launch_bounds__(192, MW) k_inv2d_pair(InvPairArgs<F> a)
{
    typedef float T;
    constexpr int PPL = 2, SH = (F - 2) / 2, HLB = inv2d_halo_lanes(SH, PPL), VP = (64 - 2 * HLB) * PPL, R = (SH <= 3) ? 4 : 8;
    constexpr int HLA = (HLB == 0) ? 0 : 2;                 // (SH + 1) / 2 <= 2 lanes of reach; 64 - 2 HLA lanes x 4 rows = VP + 128
    constexpr int RL = VP + 128 + 8, NSLOT = 8;
    constexpr int DELTA = SH & 1, CA = (SH + 1) / 2;        // ring column 0 = level-l left-half column p0 - SH - DELTA = 2 (p0/2 - CA)
    static_assert((64 - 2 * HLA) * 4 == VP + 128, "the level-(l+1) wave must cover both strips and their halo lanes");
    __shared__ __attribute__((aligned(16))) T ring[NSLOT * RL];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t logical = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    const int sA = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);
    const int64_t h0 = a.n0 >> 1, h1 = a.n1 >> 1;           // level l: pairs per column / per row
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < h1) ? (p0 + a.TP) : h1;
    const int S = (int)(pend - p0);                         // multiple of 8
    const int TPA = (S + SH - 1 + DELTA) / 2 + 1;           // steps of the level-(l+1) wave: ring columns 0 .. S + SH - 1 + DELTA
    const int NBAR = (SH + S + 2 > 2 * TPA) ? (SH + S + 2) : (2 * TPA);

    // wave priorities by role (r04, 8192^2 db4 both levels: none 126.8 us, producer first 125.0, consumers first 123.9-124.2): the
    // two level-l waves carry the stores and two thirds of the arithmetic; the producer works one step per two barriers
    if (a.prio == 1 && wv == 0) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 2 && wv != 0) __builtin_amdgcn_s_setprio(2);
    if (wv == 0) {
        // =============================== level l+1 -> ring ===============================
        const int64_t h0a = h0 >> 1, h1a = h1 >> 1;
        int64_t kw = (int64_t)sA * VP - HLB + (int64_t)(lane - HLA) * PPL;      // first coefficient pair of this lane (level l+1)
        if (kw < 0) kw += h0a;
        if (kw >= h0a) kw -= h0a;
        if (kw >= h0a) kw = 0;                                                   // (lanes past the block: their rows are never read)
        const bool park = lane >= HLA && lane < 64 - HLA;
        const bool from_ll = (a.ll != nullptr);
        const T *ls_base = (from_ll ? a.ll : a.x) + kw;
        const int64_t ls_ld = from_ll ? a.ldl : a.ldx;
        const T *ld_base = a.x + h0a + kw;
        const T *rs_base = a.x + h1a * a.ldx + kw;
        const T *rd_base = rs_base + h0a;
        const int64_t p0a = (p0 >> 1) - CA;
        T rLs[R][PPL], rLd[R][PPL], rRs[R][PPL], rRd[R][PPL];
        T iS[R][2 * PPL], iD[R][2 * PPL];
        auto load_raw = [&](const int64_t t, const int slot) __attribute__((always_inline)) {
            int64_t js = p0a + t;
            if (js < 0) js += h1a;
            if (js >= h1a) js -= h1a;
            int64_t jd = p0a + t + SH;
            if (jd < 0) jd += h1a;
            if (jd >= h1a) jd -= h1a;
            ldg_pol<WL_P_IPAIR_LD2 != 0, T, PPL>(ls_base + js * ls_ld, rLs[slot]);
            ldg_pol<WL_P_IPAIR_LD2 != 0, T, PPL>(ld_base + js * a.ldx, rLd[slot]);
            ldg_pol<WL_P_IPAIR_LD2 != 0, T, PPL>(rs_base + jd * a.ldx, rRs[slot]);
            ldg_pol<WL_P_IPAIR_LD2 != 0, T, PPL>(rd_base + jd * a.ldx, rRd[slot]);
        };
#pragma unroll
        for (int c = 0; c < SH; ++c) load_raw(c - SH, (c - SH + R) % R);
#pragma unroll
        for (int c = 0; c < SH; ++c) {
            inv_column<T, F, PPL>(rLs[(c - SH + R) % R], rLd[(c - SH + R) % R], a.tp, iS[(c - SH + R) % R]);
            inv_column<T, F, PPL>(rRs[(c - SH + R) % R], rRd[(c - SH + R) % R], a.tp, iD[(c - SH + R) % R]);
        }
#pragma unroll
        for (int c = 0; c < R; ++c) load_raw(c, c);
        for (int tb = 0; tb < TPA; tb += R) {
#pragma unroll
            for (int u = 0; u < R; ++u) {
                const int t = tb + u;
                if (t < TPA) {                                               // (workgroup-uniform)
                    inv_column<T, F, PPL>(rLs[u], rLd[u], a.tp, iS[u]);
                    inv_column<T, F, PPL>(rRs[u], rRd[u], a.tp, iD[u]);
                    if (t + R < TPA) load_raw(t + R, u);
                    wg_lds_sync(true);                                       // barrier 2t
                    T xe[2 * PPL], xo[2 * PPL];
#pragma unroll
                    for (int q = 0; q < 2 * PPL; ++q) {
                        T sw[SH + 1], dw[SH + 1];
#pragma unroll
                        for (int i = 0; i <= SH; ++i) { sw[i] = iS[(u + i + R - SH) % R][q]; dw[i] = iD[(u + i + R - SH) % R][q]; }
                        inv_pair<T, F>(sw, dw, a.tp, xe[q], xo[q]);
                    }
                    if (park) {
                        T *const c0 = ring + ((2 * t) & (NSLOT - 1)) * RL + 4 * (lane - HLA);
                        stn<T, 4>(c0, xe);
                        stn<T, 4>(c0 + RL, xo);                              // (2t + 1: the next slot, never the wrap -- NSLOT is even)
                    }
                    wg_lds_sync(true);                                       // barrier 2t + 1
                }
            }
        }
        for (int i = 2 * TPA; i < NBAR; ++i) wg_lds_sync(true);
        return;
    }

    // =============================== level l, strips 2 sA and 2 sA + 1 ===============================
    const int w = wv - 1;
    const int64_t k0 = ((int64_t)2 * sA + w) * VP + (int64_t)(lane - HLB) * PPL;
    int64_t kw = k0;
    if (kw < 0) kw += h0;
    if (kw >= h0) kw -= h0;
    if (kw >= h0) kw = 0;
    const bool store = lane >= HLB && lane < 64 - HLB && k0 < h0;
    const T *ld_base = a.x + h0 + kw;
    const T *rs_base = a.x + h1 * a.ldx + kw;
    const T *rd_base = rs_base + h0;
    const T *const lrow = ring + w * VP + PPL * lane;        // this lane's two scaling rows in a ring column
    T rLd[R][PPL], rRs[R][PPL], rRd[R][PPL];
    T iS[R][2 * PPL], iD[R][2 * PPL];
    auto load_raw = [&](const int64_t t, T (&ld)[PPL], T (&rs)[PPL], T (&rd)[PPL]) __attribute__((always_inline)) {
        int64_t js = p0 + t;
        if (js < 0) js += h1;
        int64_t jd = p0 + t + SH;
        if (jd >= h1) jd -= h1;
        ldg_pol<WL_P_IPAIR_LD != 0, T, PPL>(ld_base + js * a.ldx, ld);
        ldg_pol<WL_P_IPAIR_LD != 0, T, PPL>(rs_base + jd * a.ldx, rs);
        ldg_pol<WL_P_IPAIR_LD != 0, T, PPL>(rd_base + jd * a.ldx, rd);
    };
    T pLd[SH > 0 ? SH : 1][PPL], pRs[SH > 0 ? SH : 1][PPL], pRd[SH > 0 ? SH : 1][PPL];      // raw columns of the prologue steps
#pragma unroll
    for (int c = 0; c < SH; ++c) load_raw(c - SH, pLd[c], pRs[c], pRd[c]);
#pragma unroll
    for (int c = 0; c < R; ++c) load_raw(c, rLd[c], rRs[c], rRd[c]);
    wg_lds_sync(true);                                       // barriers 0, 1
    wg_lds_sync(true);
    // step j (0 .. SH + S - 1) runs after barrier j + 2 and takes ring column j + DELTA
#pragma unroll
    for (int c = 0; c < SH; ++c) {
        wg_lds_sync(true);
        T sv[PPL];
        ldn<T, PPL>(lrow + ((c + DELTA) & (NSLOT - 1)) * RL, sv);
        inv_column<T, F, PPL>(sv, pLd[c], a.tp, iS[(c - SH + R) % R]);
        inv_column<T, F, PPL>(pRs[c], pRd[c], a.tp, iD[(c - SH + R) % R]);
    }
    T *out = a.dst + 2 * k0;
    for (int t0 = 0; t0 < S; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int t = t0 + u;
            wg_lds_sync(true);                               // barrier SH + t + 2
            T sv[PPL];
            ldn<T, PPL>(lrow + ((t + SH + DELTA) & (NSLOT - 1)) * RL, sv);
            inv_column<T, F, PPL>(sv, rLd[u], a.tp, iS[u]);
            inv_column<T, F, PPL>(rRs[u], rRd[u], a.tp, iD[u]);
            if (t + R < S) load_raw(t + R, rLd[u], rRs[u], rRd[u]);
            T xe[2 * PPL], xo[2 * PPL];
#pragma unroll
            for (int q = 0; q < 2 * PPL; ++q) {
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int i = 0; i <= SH; ++i) { sw[i] = iS[(u + i + R - SH) % R][q]; dw[i] = iD[(u + i + R - SH) % R][q]; }
                inv_pair<T, F>(sw, dw, a.tp, xe[q], xo[q]);
            }
            if (store) {
                const int64_t p = p0 + t;
                stg_pol<WL_P_IPAIR_ST != 0, T, 2 * PPL>(out + (2 * p) * a.ldd, xe);
                stg_pol<WL_P_IPAIR_ST != 0, T, 2 * PPL>(out + (2 * p + 1) * a.ldd, xo);
            }
        }
    }
    for (int i = SH + S + 2; i < NBAR; ++i) wg_lds_sync(true);
}

bool inv2d_pair_ok(int F, int64_t n0, int64_t n1)
{
    // level l output n0 x n1; level l+1 output n0/2 x n1/2 must satisfy the single-level kernel's conditions too
    // (10 taps: the two code paths together need 239 VGPRs -- two waves per SIMD -- and lose to two launches: 209 vs 188 us)
    if (F < 2 || F > (int)opt("WL_INV_PAIR_FMAX", 8) || F > 10 || (F & 1)) return false;
    return n0 >= 1024 && (n0 % 16) == 0 && n1 >= 64 && (n1 % 32) == 0 && n0 < ((int64_t)1 << 30);
}

template <int F>
static hipError_t launch_inv2d_pair_f(hipStream_t st, const Taps<float> &taps, const float *x, int64_t ldx, const float *ll, int64_t ldl,
                                      float *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count)
{
    constexpr int SH = (F - 2) / 2, HLB = inv2d_halo_lanes(SH, 2), VP = (64 - 2 * HLB) * 2;
    InvPairArgs<F> a;
    a.x = x; a.ldx = ldx; a.ll = ll; a.ldl = ldl; a.dst = dst; a.ldd = ldd; a.n0 = n0; a.n1 = n1;
    const int64_t h0 = n0 >> 1, h1 = n1 >> 1;
    a.nstrips = (int)((h0 + 2 * VP - 1) / (2 * VP));
    int TP = i_env("WL_INVPAIR_TP", 64);
    if (TP < 8 || (TP % 8) != 0) TP = 64;
    while (TP > 16 && (int64_t)a.nstrips * ((h1 + TP - 1) / TP) < (int64_t)cu_count * 4) TP >>= 1;
    a.TP = TP;
    a.nchunks = (int)((h1 + TP - 1) / TP);
    a.tp = shrink_i<float, F>(taps);
    a.prio = i_env("WL_INVPAIR_PRIO", 2);
    // four waves per SIMD: the 8-tap instance spills 18 VGPRs for it and is still far ahead of three waves without spills
    // (8192^2 db4, both levels: 126 us against 175) -- like the single-level kernel this one is latency-bound per wave
    constexpr int MW = (F <= 8) ? 4 : 2;
    hipLaunchKernelGGL((k_inv2d_pair<F, MW>), dim3((unsigned)(a.nstrips * a.nchunks)), dim3(192), 0, st, a);
    return hipGetLastError();
}

static hipError_t launch_inv2d_pair(hipStream_t st, const Taps<float> &taps, const float *x, int64_t ldx, const float *ll, int64_t ldl,
                                    float *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count)
{
    switch (taps.F) {
    case 2: return launch_inv2d_pair_f<2>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count);
    case 4: return launch_inv2d_pair_f<4>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count);
    case 6: return launch_inv2d_pair_f<6>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count);
    case 8: return launch_inv2d_pair_f<8>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count);
    case 10: return launch_inv2d_pair_f<10>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count);
    default: return hipErrorInvalidValue;
    }
}

#define WL_DISPATCH_FI(F_, ...)                              \
    switch (F_) {                                            \
    case 2: { constexpr int FF = 2; __VA_ARGS__; } break;    \
    case 4: { constexpr int FF = 4; __VA_ARGS__; } break;    \
    case 6: { constexpr int FF = 6; __VA_ARGS__; } break;    \
    case 8: { constexpr int FF = 8; __VA_ARGS__; } break;    \
    case 10: { constexpr int FF = 10; __VA_ARGS__; } break;  \
    default: break;                                          \
    }

template <typename T>
int filter_inv_levels(void *ws, bool ws_gen, int cu_count, int path, hipStream_t st, const BoxSpec &b,
                      T *y, const T *x, const Taps<T> &taps, int L, const char **kernel_name, int *hip_err)
{
#define WL_TRYI(expr)                                                  \
    do {                                                               \
        hipError_t e__ = (expr);                                       \
        if (e__ != hipSuccess) { if (hip_err) *hip_err = (int)e__; return WL_EHIP; } \
    } while (0)
    const int64_t N = b.dims[0] * b.dims[1] * b.dims[2];
    Work<T> w = carve<T>(ws, N, b.nt, ws_gen);
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    const bool fastF = (path == 0) && (F % 2 == 0) && (F <= 10) && i_env("WL_NO_INVFAST", 0) == 0;
    const bool two_d = (b.nd == 2 && b.nt == 2);
    const bool lines = (b.nt == 1 && b.nd <= 2);
    const int64_t nlines = lines ? b.dims[1] : 1;
    const char *dominant = nullptr;

    int pp = 0;
    const T *llsrc = nullptr;              // reconstruction of the deeper level (dense), nullptr: take it from x
    Strides3 llsrc_st = {{0, 0, 0}};
    int l_start = L;
    // ---- LDS tail: the deepest levels whose output fits one workgroup's LDS ----
    if (path == 0 && i_env("WL_NO_INVTAIL", 0) == 0 && (two_d || lines) && b.full.s[0] == 1) {
        int l_lo = L + 1;
        for (int q = L; q >= 1; --q) {
            int64_t nq[3];
            level_box(b, q, nq);
            const int64_t blk = two_d ? nq[0] * nq[1] : nq[0];
            const bool fits = two_d ? (((nq[0] | 1) * nq[1]) <= inv_tail_cap<T>() + 256 && nq[1] <= 256)
                                    : (2 * nq[0] <= 2 * (int64_t)inv_tail_cap<T>());
            // 12 ... 20 taps on power-of-two blocks of <= 4096 elements / lines: the LDS-resident tail is instantiated for them (as in
            // the forward direction); other long-filter shapes leave the tail at 64 / 16 samples, the chip-wide kernels take the rest
            const bool long_tail = F >= 12 && i_env("WL_LONG_TAIL", 1) != 0 && i_env("WL_TAIL2", 1) != 0 &&
                                   (two_d ? tail2_inv_ok<T>(F, 2, nq[0], nq[1], L - q + 1, y, 0) : tail2_inv_ok<T>(F, 1, nq[0], 1, L - q + 1, y, b.full.s[1]));
            const int64_t vl_cap = long_tail ? (int64_t)4096
                                   : ((vlong_filter_ok(taps.F) && i_env("WL_NO_LONGF", 0) == 0) ? (two_d ? 64 : 16) : ((int64_t)1 << 40));
            if (blk <= (int64_t)i_env("WL_INVTAIL_MAX", 4096) && blk <= vl_cap && blk <= inv_tail_cap<T>() && fits && nq[0] < (1 << 20)) l_lo = q; else break;
        }
        if (l_lo <= L) {
            int64_t nq[3];
            level_box(b, l_lo, nq);
            const bool to_y = (l_lo == 1);
            T *res = to_y ? y : (pp ? w.B : w.A);
            Strides3 res_st = to_y ? b.full : dense_strides(nq);
            // power-of-two blocks / lines of <= 16 KiB with a <= 10-tap filter: the latency-optimised tail (wl_tail.hip)
            const bool t2 = i_env("WL_TAIL2", 1) != 0 &&
                            (two_d ? tail2_inv_ok<T>(F, 2, nq[0], nq[1], L - l_lo + 1, res, 0)
                                   : tail2_inv_ok<T>(F, 1, nq[0], 1, L - l_lo + 1, res, res_st.s[1]));
            if (two_d) {
                if (t2) WL_TRYI(launch_tail2_inv<T>(st, taps, x, b.full.s[1], 0, res, res_st.s[1], 0, 1, (int)nq[0], (int)nq[1], 2, L - l_lo + 1));
                else WL_TRYI(launch_tail_inv<T>(st, taps, x, b.full.s[1], 0, res, res_st.s[1], 0, 1, (int)nq[0], (int)nq[1], 2, L - l_lo + 1));
            } else {
                if (t2) WL_TRYI(launch_tail2_inv<T>(st, taps, x, 0, b.full.s[1], res, 0, res_st.s[1], (int)nlines, (int)nq[0], 1, 1, L - l_lo + 1));
                else WL_TRYI(launch_tail_inv<T>(st, taps, x, 0, b.full.s[1], res, 0, res_st.s[1], (int)nlines, (int)nq[0], 1, 1, L - l_lo + 1));
            }
            dominant = t2 ? "k_tail2_inv" : "k_tail_inv";
            llsrc = res; llsrc_st = dense_strides(nq); pp ^= 1;
            l_start = l_lo - 1;
        }
    }
    // ---- 3-D: the deepest levels whose output is a power-of-two box of <= 4096 elements, one workgroup (wl_tail.hip) ----
    if (path == 0 && b.nd == 3 && b.nt == 3 && i_env("WL_TAIL3", 1) != 0 && b.full.s[0] == 1 && l_start == L) {
        int l_lo = L + 1;
        for (int q = L; q >= 1; --q) {
            int64_t nq[3];
            level_box(b, q, nq);
            if (tail3_ok<T>(F, nq[0], nq[1], nq[2], L - q + 1)) l_lo = q; else break;
        }
        if (l_lo <= L) {
            int64_t nq[3];
            level_box(b, l_lo, nq);
            const bool to_y = (l_lo == 1);
            T *res = to_y ? y : (pp ? w.B : w.A);
            Strides3 res_st = to_y ? b.full : dense_strides(nq);
            WL_TRYI(launch_tail3<T>(st, taps, 0, x, b.full.s[1], b.full.s[2], res, res_st.s[1], res_st.s[2], (int)nq[0], (int)nq[1], (int)nq[2],
                                    L - l_lo + 1));
            dominant = "k_tail3";
            llsrc = res; llsrc_st = dense_strides(nq); pp ^= 1;
            l_start = l_lo - 1;
        }
    }
    for (int l = l_start; l >= 1; --l) {
        int64_t n[3];
        level_box(b, l, n);
        Extent3 ext = {{n[0], n[1], n[2]}};
        Extent3 lo = low_corner(b, n);
        Strides3 box_st = dense_strides(n);
        T *res = (l == 1) ? y : (pp ? w.B : w.A);
        Strides3 res_st = (l == 1) ? b.full : box_st;
        bool done = false;
        // the LDS-exchange level kernel (wl_inv2d_long.hip: which filter lengths per element type), output rows a multiple of 256
        auto try_lds_long = [&]() -> hipError_t {
            if (!done && path == 0 && two_d && i_env("WL_INVLONG2D", 1) != 0 && n[0] >= i_env("WL_INVLONG2D_MIN_ROWS", 256) && b.full.s[0] == 1 &&
                inv2d_long_ok(F, n[0], n[1], (int)sizeof(T)) && (b.full.s[1] % VEC) == 0 && (res_st.s[1] % VEC) == 0 && i_al16(x) && i_al16(res) &&
                (!llsrc || (i_al16(llsrc) && (llsrc_st.s[1] % 2) == 0))) {
                hipError_t e = inv2d_long_launch<T>(st, taps, x, b.full.s[1], llsrc, llsrc ? llsrc_st.s[1] : 0, res, res_st.s[1], n[0], n[1], cu_count);
                if (e != hipSuccess) return e;
                dominant = "k_inv2d_lds_long";
                done = true;
            }
            return hipSuccess;
        };
        // 8 / 10 taps: ahead of the streaming kernels where it is enabled for them (WL_INVLONG_FMIN / _FMIN64) and the block is large
        if (F <= 10 && n[0] * n[1] >= (int64_t)i_env("WL_INVLONG_SHORT_MIN", 1 << 22) &&
            !(sizeof(T) == 4 && F <= 8 && l >= 2 && (l % 2) == 0)) WL_TRYI(try_lds_long());

        // ---- big 2-D blocks: levels l and l-1 in one launch, the level-l reconstruction handed over through an LDS column ring ----
        if constexpr (sizeof(T) == 4) {
            // (l even: the pairs end at level 1, so the two biggest levels share a launch)
            if (!done && fastF && two_d && l >= 2 && (l % 2) == 0 && i_env("WL_INV_PAIR", 1) != 0 && b.full.s[0] == 1 && (b.full.s[1] % VEC) == 0 && i_al16(x) &&
                (!llsrc || (i_al16(llsrc) && (llsrc_st.s[1] % 2) == 0))) {
                int64_t n1[3];
                level_box(b, l - 1, n1);                    // output extents of the shallower level
                T *res1 = (l - 1 == 1) ? y : (pp ? w.B : w.A);
                const int64_t r_ld = (l - 1 == 1) ? b.full.s[1] : n1[0];
                if (n1[0] == 2 * n[0] && n1[1] == 2 * n[1] && n1[0] * n1[1] >= (int64_t)i_env("WL_INV_PAIR_MIN", 1 << 24) &&
                    inv2d_pair_ok(F, n1[0], n1[1]) && (r_ld % VEC) == 0 && i_al16(res1)) {
                    WL_TRYI(launch_inv2d_pair(st, taps, x, b.full.s[1], llsrc, llsrc_st.s[1], res1, r_ld, n1[0], n1[1], cu_count));
                    dominant = "k_inv2d_pair";
                    llsrc = res1; llsrc_st = dense_strides(n1); pp ^= 1;
                    --l;                                     // two levels consumed
                    continue;
                }
            }
        }
        // ---- small 2-D blocks: two levels (l and l-1) per launch, LDS tiles (wl_tile.hip) ----
        // (round 5: 12..20 taps as well -- these levels ran one streaming launch each and the 128^2 level as two line passes)
        if (!done && (fastF || (F >= 12 && F <= 20 && (F % 2) == 0 && i_env("WL_TILE_INV_LONG", 1) != 0)) && two_d && path == 0 && l >= 2 &&
            i_env("WL_TILE_INV", 1) != 0 && b.full.s[0] == 1) {
            int64_t n1[3];
            level_box(b, l - 1, n1);                        // output extents of the shallower level
            if (inv2d_tile2_ok<T>(F, n1[0], n1[1]) && n1[0] <= i_env("WL_TILE_INV_MAX", 1024) && n1[1] <= i_env("WL_TILE_INV_MAX", 1024)) {
                T *res1 = (l - 1 == 1) ? y : (pp ? w.B : w.A);
                const int64_t r_ld = (l - 1 == 1) ? b.full.s[1] : n1[0];
                const T *ss = llsrc ? llsrc : x;
                const int64_t sls = llsrc ? llsrc_st.s[1] : b.full.s[1];
                WL_TRYI(inv2d_tile2_launch<T>(st, taps, x, b.full.s[1], ss, sls, res1, r_ld, (int)n1[0], (int)n1[1]));
                dominant = "k_inv2d_tile2";
                llsrc = res1; llsrc_st = dense_strides(n1); pp ^= 1;
                --l;                                         // two levels consumed
                continue;
            }
        }
        // ---- lines: two levels (l and l-1) per launch ----
        if (fastF && lines && l >= 2 && i_env("WL_NO_INV1D2", 0) == 0 && n[0] >= 512 && (n[0] % 8) == 0 && b.full.s[0] == 1 &&
            i_al16(x) && i_al16(y) && (nlines == 1 || ((b.full.s[1] % VEC) == 0 && (!llsrc || (llsrc_st.s[1] % VEC) == 0)))) {
            int64_t n1[3];
            level_box(b, l - 1, n1);                        // output extents of the shallower level: n1[0] = 2 n[0]
            T *res1 = (l - 1 == 1) ? y : (pp ? w.B : w.A);
            const int64_t r_ls = (l - 1 == 1) ? b.full.s[1] : n1[0];
            const T *ss = llsrc ? llsrc : x;
            const int64_t sls = llsrc ? llsrc_st.s[1] : b.full.s[1];
            if (i_al16(ss) && i_al16(res1)) {
                WL_DISPATCH_FI(F, WL_TRYI((launch_inv1d2<T, FF>(st, taps, ss, sls, x + (n[0] >> 1), b.full.s[1], x + n[0], b.full.s[1],
                                                                res1, r_ls, n1[0], nlines, cu_count)));
                               done = true);
            }
            if (done) {
                dominant = "k_inv1d_stream2";
                llsrc = res1; llsrc_st = dense_strides(n1); pp ^= 1;
                --l;                                         // two levels consumed
                continue;
            }
        }
        if (fastF && lines && n[0] >= 512 && (n[0] % 8) == 0 && b.full.s[0] == 1 && i_al16(x) && i_al16(y) &&
            (nlines == 1 || (b.full.s[1] % VEC) == 0)) {
            const T *ss = llsrc ? llsrc : x;
            const int64_t sls = llsrc ? llsrc_st.s[1] : b.full.s[1];
            WL_DISPATCH_FI(F, WL_TRYI((launch_inv1d<T, FF>(st, taps, ss, sls, x + (n[0] >> 1), b.full.s[1], res, res_st.s[1],
                                                           n[0], nlines, cu_count)));
                           done = true);
            if (done) dominant = "k_inv1d_stream";
        }
        // ---- fused 2-D level: one pass over HBM ----
        if (!done && fastF && two_d && i_env("WL_NO_INV2D", 0) == 0 && n[0] >= 128 && (n[0] % 8) == 0 && n[1] >= 16 &&
            (n[1] % 8) == 0 && (F < 10 || (n[1] % 16) == 0) && b.full.s[0] == 1 && (b.full.s[1] % VEC) == 0 && (res_st.s[1] % VEC) == 0 && i_al16(x) && i_al16(res) &&
            (!llsrc || (i_al16(llsrc) && (llsrc_st.s[1] % 2) == 0))) {
            // pairs per lane: 2 (8-byte loads / 16-byte stores for f32) from 256 rows, 1 below; 4 is a tuning option for f32
            int ppl = i_env("WL_INV2D_PPL", 2);
            if (n[0] < 256) ppl = 1;
            else if (n[0] < 512 || sizeof(T) != 4 || ppl != 4) ppl = 2;
            if (ppl == 4) {
                if constexpr (sizeof(T) == 4) {
                    WL_DISPATCH_FI(F, WL_TRYI((launch_inv2d<T, FF, 4>(st, taps, x, b.full.s[1], llsrc, llsrc_st.s[1], res, res_st.s[1],
                                                                       n[0], n[1], cu_count)));
                                    done = true);
                }
            } else if (ppl == 2) {
                WL_DISPATCH_FI(F, WL_TRYI((launch_inv2d<T, FF, 2>(st, taps, x, b.full.s[1], llsrc, llsrc_st.s[1], res, res_st.s[1],
                                                                   n[0], n[1], cu_count)));
                                done = true);
            } else {
                WL_DISPATCH_FI(F, WL_TRYI((launch_inv2d<T, FF, 1>(st, taps, x, b.full.s[1], llsrc, llsrc_st.s[1], res, res_st.s[1],
                                                                   n[0], n[1], cu_count)));
                                done = true);
            }
            if (done) dominant = "k_inv2d_stream";
        }
        if (!done && fastF && two_d && n[0] >= 512 && (n[0] % 8) == 0 && n[1] >= 32 && (n[1] % 16) == 0 &&
            b.full.s[0] == 1 && (b.full.s[1] % VEC) == 0 && i_al16(x) && i_al16(y)) {
            // dim-1 pass: every column is a line of length n0: s = rows [0,h0), d = rows [h0,n0)
            const int64_t h0 = n[0] >> 1, h1 = n[1] >> 1;
            if (!w.T0) return WL_RETRY_GEN;
            T *tmp = w.T0;                         // n0 x n1 dense
            bool ok = true;
            if (llsrc) {
                // columns [0,h1): approximation from the deeper reconstruction; columns [h1,n1): from x
                WL_DISPATCH_FI(F, WL_TRYI((launch_inv1d<T, FF>(st, taps, llsrc, llsrc_st.s[1], x + h0, b.full.s[1], tmp, n[0],
                                                               n[0], h1, cu_count)));
                               WL_TRYI((launch_inv1d<T, FF>(st, taps, x + h1 * b.full.s[1], b.full.s[1], x + h1 * b.full.s[1] + h0,
                                                            b.full.s[1], tmp + h1 * n[0], n[0], n[0], h1, cu_count))));
            } else {
                WL_DISPATCH_FI(F, WL_TRYI((launch_inv1d<T, FF>(st, taps, x, b.full.s[1], x + h0, b.full.s[1], tmp, n[0], n[0], n[1],
                                                               cu_count))));
            }
            // dim-2 pass
            WL_DISPATCH_FI(F, WL_TRYI((launch_inv_dim2<T, FF>(st, taps, tmp, n[0], res, res_st.s[1], n[0], n[1], cu_count))));
            (void)ok;
            done = true;
            dominant = "k_inv_dim2_stream";
        }
        // ---- batch of independent 2-D blocks (box n0 x n1 x B, first two axes transformed): the fused inverse level
        //      kernel with the planes over blockIdx.y ----
        if (!done && path == 0 && (fastF || (F >= 12 && F <= 20)) && b.nd == 3 && b.nt == 2 && b.full.s[0] == 1 && (l == 1 || res_st.s[1] == n[0])) {
            hipError_t e = hipSuccess;
            const char *kn = nullptr;
            done = inv2d_planes<T>(st, taps, x, b.full.s[1], b.full.s[2], llsrc, res, n[0], n[1], n[2], llsrc ? (int)n[2] : 0, cu_count, &e, &kn,
                                   res_st.s[2]);      // (l == 1: y's own plane stride, the caller's image_stride; deeper levels: dense)
            WL_TRYI(e);
            if (done) dominant = kn;
        }
        // ---- 12..20 taps, output rows a multiple of 256: the whole 2-D level in one pass (wl_inv2d_long.hip) ----
        if (!done && F >= 12) WL_TRYI(try_lds_long());
        // ---- long filters (12..24 taps) ----
        if (!done && path == 0 && long_filter_ok(F) && i_env("WL_NO_LONGF", 0) == 0 && b.full.s[0] == 1) {
            hipError_t e = hipSuccess;
            const int64_t h0 = n[0] >> 1, h1 = n[1] >> 1, ldx = b.full.s[1];
            if (lines) {
                const T *ss = llsrc ? llsrc : x;
                const int64_t sls = llsrc ? llsrc_st.s[1] : ldx;
                done = long_lines_inv_level<T>(st, taps, ss, sls, x + h0, ldx, res, res_st.s[1], n[0], nlines, cu_count, &e);
                WL_TRYI(e);
            } else if (two_d && long_shape2d_ok(F, n[0], n[1]) && (ldx % VEC) == 0 &&
                       (res_st.s[1] % VEC) == 0 && i_al16(x) && i_al16(res) && (!llsrc || (i_al16(llsrc) && (llsrc_st.s[1] % VEC) == 0))) {
                // columns (dim 1) into T0, the approximation quadrant from the deeper reconstruction; then rows (dim 2)
                const T *ss = llsrc ? llsrc : x;
                const int64_t sls = llsrc ? llsrc_st.s[1] : ldx;
                if (!w.T0) return WL_RETRY_GEN;
                done = long_lines_inv_level<T>(st, taps, ss, sls, x + h0, ldx, w.T0, n[0], n[0], h1, cu_count, &e);
                WL_TRYI(e);
                if (done) {
                    bool ok = long_lines_inv_level<T>(st, taps, x + h1 * ldx, ldx, x + h1 * ldx + h0, ldx, w.T0 + h1 * n[0], n[0], n[0], h1,
                                                      cu_count, &e);
                    WL_TRYI(e);
                    ok = ok && long_axis_level<T>(st, taps, 0, w.T0, n[0], res, res_st.s[1], n[0], n[1], cu_count, &e);
                    WL_TRYI(e);
                    if (!ok) return WL_EINVAL_ARG;
                }
            }
            if (done) dominant = (vlong_only(F) || n[0] < 512) ? "k_vl_lines" : "k_long_lines";
        }
        if (!done && fastF && b.nd == 3 && b.nt == 3 && i_env("WL_NO_FAST3D", 0) == 0 && b.full.s[0] == 1 && res_st.s[0] == 1) {
            hipError_t e3 = hipSuccess;
            if (!w.T0) return WL_RETRY_GEN;
            const char *k3 = "k_inv_axis_stream";
            done = fast3d_inv_level<T>(st, taps, x, b.full.s[1], b.full.s[2], llsrc, res, res_st.s[1], res_st.s[2], n,
                                       w.T0, w.T1, cu_count, &e3, &k3);
            WL_TRYI(e3);
            if (done) dominant = k3;
        }
        // ---- 2-D level of any even extents, F <= 10: one LDS-tile launch instead of two generic passes (wl_gtile.hip) ----
        if (!done && fastF && two_d && path == 0 && i_env("WL_GTILE", 1) && b.full.s[0] == 1 && res_st.s[0] == 1 && gtile_ok(F, n[0], n[1])) {
            WL_TRYI(gtile_launch<T>(st, taps, 0, x, b.full.s[1], res, res_st.s[1], const_cast<T *>(llsrc), llsrc ? llsrc_st.s[1] : 0, (int)n[0],
                                    (int)n[1]));
            dominant = "k_inv2d_gtile";
            done = true;
        }
        if (!done) {
            if (b.nt > 1 && !w.T0) return WL_RETRY_GEN;
            const T *in = x;
            Strides3 in_st = b.full;
            int tog = 0;
            for (int a = 0; a < b.nt; ++a) {
                const bool firstp = (a == 0), lastp = (a == b.nt - 1);
                T *out; Strides3 out_st;
                if (lastp) { out = res; out_st = res_st; }
                else { out = tog ? w.T1 : w.T0; out_st = box_st; tog ^= 1; }
                const bool any = path == 0 && i_env("WL_ANYAXIS", 1) && any_axis_ok(F, ext, a);
                if (any)
                    WL_TRYI(any_axis_pass<T>(st, taps, 0, in, in_st, out, out_st, firstp ? const_cast<T *>(llsrc) : (T *)nullptr, llsrc_st, ext, a, lo));
                else
                    WL_TRYI(generic_inv_filter_pass<T>(st, taps, in, in_st, firstp ? llsrc : (const T *)nullptr, llsrc_st, out, out_st, ext, a, lo));
                in = out; in_st = out_st;
                if (lastp && !dominant) dominant = any ? "k_inv_any" : "k_generic_inv_filter";
            }
        }
        llsrc = res; llsrc_st = box_st; pp ^= 1;
    }
    if (kernel_name) *kernel_name = dominant ? dominant : "none";
    return WL_OK;
#undef WL_TRYI
}

template <typename T>
bool fast_lines_inv_level(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls,
                          T *dst, int64_t o_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err)
{
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    *err = hipSuccess;
    if ((F % 2) != 0 || F > 10 || n < 512 || (n % 8) != 0 || !i_al16(ssrc) || !i_al16(dsrc) ||
        !i_al16(dst) || (s_ls % VEC) != 0 || (d_ls % VEC) != 0 || (o_ls % VEC) != 0)
        return false;
    bool done = false;
    WL_DISPATCH_FI(F, *err = launch_inv1d<T, FF>(st, taps, ssrc, s_ls, dsrc, d_ls, dst, o_ls, n, nlines, cu_count);
                   done = true);
    return done;
}
template bool fast_lines_inv_level<float>(hipStream_t, const Taps<float> &, const float *, int64_t, const float *, int64_t,
                                          float *, int64_t, int64_t, int64_t, int, hipError_t *);
template bool fast_lines_inv_level<double>(hipStream_t, const Taps<double> &, const double *, int64_t, const double *, int64_t,
                                           double *, int64_t, int64_t, int64_t, int, hipError_t *);

// The fused 2-D inverse level kernel on a batch of planes (the dim-1 + dim-2 passes of a 3-D inverse level): plane p of
// x (strides x1, x2; the first nll planes take their approximation quadrant from plane p of ll, dense h0 x h1)
// -> plane p of dst (dense n0 x n1).
template <typename T>
bool inv2d_planes(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *ll, T *dst, int64_t n0, int64_t n1,
                  int64_t nplanes, int nll, int cu_count, hipError_t *err, const char **kernel, int64_t dst_ps)
{
    if (dst_ps <= 0) dst_ps = n0 * n1;               // dense destination planes (the 3-D inverse level); a batch of images passes its own
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    *err = hipSuccess;
    if ((F % 2) != 0 || F > 20 || n0 < 256 || (n0 % 8) != 0 || n1 < 16 || (n1 % 16) != 0 || (x1 % VEC) != 0 || (x2 % VEC) != 0 ||
        (dst_ps % VEC) != 0 || !i_al16(x) || !i_al16(dst) || (ll && !i_al16(ll)) || nplanes > 65535)
        return false;
    // the LDS-exchange level kernel (wl_inv2d_long.hip) where it is enabled for this filter length and the batch is large
    if (i_env("WL_INVLONG2D", 1) != 0 && inv2d_long_ok(F, n0, n1, (int)sizeof(T)) && n0 >= i_env("WL_INVLONG2D_MIN_ROWS", 256) &&
        (F >= 12 || n0 * n1 * nplanes >= (int64_t)i_env("WL_INVLONG_SHORT_MIN", 1 << 22))) {
        const InvLongBatch bt = {nplanes, x2, (n0 >> 1) * (n1 >> 1), dst_ps, ll ? nll : 0};
        *err = inv2d_long_launch<T>(st, taps, x, x1, ll, n0 >> 1, dst, n0, n0, n1, cu_count, bt);
        if (kernel) *kernel = "k_inv2d_lds_long";
        return true;
    }
    if (F > 10) return false;
    if (kernel) *kernel = "k_inv2d_stream";
    bool ok = false;
    WL_DISPATCH_FI(F, *err = launch_inv2d<T, FF, 2>(st, taps, x, x1, ll, n0 >> 1, dst, n0, n0, n1, cu_count, nplanes, x2,
                                                     (n0 >> 1) * (n1 >> 1), dst_ps, nll);
                   ok = true);
    return ok;
}
template bool inv2d_planes<float>(hipStream_t, const Taps<float> &, const float *, int64_t, int64_t, const float *, float *, int64_t,
                                  int64_t, int64_t, int, int, hipError_t *, const char **, int64_t);
template bool inv2d_planes<double>(hipStream_t, const Taps<double> &, const double *, int64_t, int64_t, const double *, double *, int64_t,
                                   int64_t, int64_t, int, int, hipError_t *, const char **, int64_t);

template int filter_inv_levels<float>(void *, bool, int, int, hipStream_t, const BoxSpec &, float *, const float *,
                                      const Taps<float> &, int, const char **, int *);
template int filter_inv_levels<double>(void *, bool, int, int, hipStream_t, const BoxSpec &, double *, const double *,
                                       const Taps<double> &, int, const char **, int *);

}  // namespace wl
