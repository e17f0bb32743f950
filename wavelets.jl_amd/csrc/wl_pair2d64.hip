// wl_pair2d64.hip -- TWO fused forward 2-D filter-bank levels per launch, Float64, even F <= 10: the Float64 instance of
// k_fwd2d_pair (wl_pair2d.hip; read its header for the scheme).  Differences: a lane holds TWO rows (one 16-byte load per
// column), a main wave covers 128 rows, so a workgroup of W main waves + helper + level-(l+1) wave owns a strip of 128 W rows
// = 64 W approximation rows; the level-(l+1) wave holds W of them per lane (window W + 8 rows); the helper supplies 24 halo
// rows (12 lanes) and the 8 approximation rows above the strip.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <int F>
struct Pair2DArgs64 {
    const double *src; int64_t lds;
    double *y; int64_t ldy;
    double *ll; int64_t ldll;         // approximation after both levels: next stage's input buffer, or y itself
    int64_t ms, ns;                   // level-l block
    int TJ;                           // owned input columns per chunk (multiple of 32)
    int nstrips, nchunks;
    int rev;
    int prio;
    TapsF<double, F> tp;
};

template <int F, int W, int LVL1>
__global__ void __launch_bounds__(64 * (W + 2), 3) k_fwd2d_pair64(Pair2DArgs64<F> a)
{
    typedef double T;
    typedef D2 T2;
    constexpr int SH = (F - 2) / 2;
    constexpr int R = 16, U = 8, PFD = (R - F) / 2;
    constexpr int NT1 = 64 * (W + 1);                 // lanes of the level-l exchange (main waves + helper)
    constexpr int ROWS1 = 2 * NT1 + 16;
    constexpr int NPL = 64 * W;                       // owned lanes
    constexpr int NLOAD = NPL + 12;                   // + halo lanes: 24 rows above the strip
    constexpr int RL = 64 * W + 16;                   // approximation rows per ring slot: 64 W owned + 8 halo (+ 8 pad)
    constexpr int NSLOT = 16;
    constexpr int RW = W;                             // approximation rows per lane of the level-(l+1) wave
    constexpr int HS = W / 2;                         // its s2 (and d2) rows per lane
    __shared__ __attribute__((aligned(16))) T2 x1[2 * ROWS1];
    __shared__ __attribute__((aligned(16))) T ll1[NSLOT * RL];
    __shared__ __attribute__((aligned(16))) T2 x2[RL];

    // g[m] = (-1)^m h[m] exactly: only the scaling taps occupy SGPRs, a detail term multiplies by the negated tap (a source modifier)
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1u : 0u);
    uint32_t logical = first + (b >> 3);
    if (a.rev) logical = first + (cnt - 1 - (logical - first));
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1, nxj2 = ns >> 2;
    const int msi = (int)ms, hmi = msi >> 1, hm2i = msi >> 2;
    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S_own = (int)((jend - j0) >> 1);        // steps whose level-l outputs this chunk owns (multiple of 16)
    const int S = S_own + F;                          // the level-(l+1) column of ring columns c .. c+F-1 is taken up at step c + F
    T *const yb = a.y;
    T *const llb = a.ll ? a.ll : a.y;
    const int64_t ldl = a.ll ? a.ldll : a.ldy;

    if (a.prio && wv >= W) __builtin_amdgcn_s_setprio(2);     // (as in k_fwd2d_pair: the single-wave roles issue first)
    if (wv == W + 1) {
        // =============================== the level-(l+1) wave ===============================
        const int j = (int)(threadIdx.x & 63);
        const int s2row = strip * (32 * W) + HS * j;                         // first s2 row of this lane
        int d2row = s2row + 4;  if (d2row >= hm2i) d2row -= hm2i;            // first d2 row (details are stored shifted by 4)
        const T *const l1 = ll1 + RW * j;
        T2 *const xw = x2 + RW * j;
        const int64_t kbase2 = j0 >> 2;
        for (int t = 0; t < F; ++t) wg_lds_sync(true);
        for (int t = F; t < S; t += 2) {
            wg_lds_sync(true);                                               // barrier t (even): ring columns <= t-1 are visible
            __builtin_amdgcn_sched_barrier(0);
            {
                T sa[RW], da[RW];
#pragma unroll
                for (int m = 0; m < F; ++m) {
                    const T *const col = l1 + ((t - F + m) & (NSLOT - 1)) * RL;
                    T xm[RW];
#pragma unroll
                    for (int c = 0; c < RW / 2; ++c) {
                        const T2 v = *reinterpret_cast<const T2 *>(col + 2 * c);
                        xm[2 * c] = v.x; xm[2 * c + 1] = v.y;
                    }
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        if (m == 0) { sa[r] = a.tp.h[0] * xm[r]; da[r] = gq(F - 1) * xm[r]; }
                        else { sa[r] = sa[r] + a.tp.h[m] * xm[r]; da[r] = da[r] + gq(F - 1 - m) * xm[r]; }
                    }
                }
#pragma unroll
                for (int r = 0; r < RW; ++r) xw[r] = T2{sa[r], da[r]};
            }
            wg_lds_sync(true);                                               // barrier t + 1 (odd)
            __builtin_amdgcn_sched_barrier(0);
            {
                T2 E[RW + 8];
#pragma unroll
                for (int c = 0; c < RW + 8; ++c) E[c] = xw[c];
                T2 P[HS], Q[HS];                       // P[q] = {ss2, sd2} of row s2row + q;  Q[q] = {ds2, dd2} of row d2row + q
#pragma unroll
                for (int q = 0; q < HS; ++q) {
                    T2 s = a.tp.h[0] * E[2 * q];
#pragma unroll
                    for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * E[2 * q + m];
                    T2 d = gq(F - 1) * E[2 * q + 10 - F];
#pragma unroll
                    for (int m = F - 2; m >= 0; --m) d = d + gq(m) * E[2 * q + 9 - m];
                    P[q] = s;
                    Q[q] = d;
                }
                const int64_t k2 = kbase2 + ((t - F) >> 1);
                int64_t kd2 = k2 + SH;
                if (kd2 >= nxj2) kd2 -= nxj2;
                T *const ck = yb + k2 * a.ldy, *const ckd = yb + (nxj2 + kd2) * a.ldy, *const cl = llb + k2 * ldl;   // (uniform)
                if constexpr (HS == 2) {
                    *reinterpret_cast<T2 *>(cl + s2row) = T2{P[0].x, P[1].x};
                    store_pol<WL_P_PAIR64_ST2>(reinterpret_cast<T2 *>(ck + (hm2i + d2row)), T2{Q[0].x, Q[1].x});
                    store_pol<WL_P_PAIR64_ST2>(reinterpret_cast<T2 *>(ckd + s2row), T2{P[0].y, P[1].y});
                    store_pol<WL_P_PAIR64_ST2>(reinterpret_cast<T2 *>(ckd + (hm2i + d2row)), T2{Q[0].y, Q[1].y});
                } else {
                    cl[s2row] = P[0].x;
                    ck[hm2i + d2row] = Q[0].x;
                    ckd[s2row] = P[0].y;
                    ckd[hm2i + d2row] = Q[0].y;
                }
            }
        }
        return;
    }

    // =============================== main waves and the helper: level l ===============================
    const int lp = threadIdx.x;                       // L': lane index within the workgroup's strip (0 .. NT1-1)
    const int gi = strip * (2 * NPL) + 2 * lp;        // first row of this lane (halo lanes may exceed ms: wrap)
    int row = gi;
    if (row >= msi) row -= msi;
    const bool loader = lp < NLOAD;
    const bool helper = (wv == W);
    const int ko = gi >> 1;
    int kod = ko + 4;  if (kod >= hmi) kod -= hmi;    // the d row of this lane
    const bool odd = (lp & 1) != 0;
    // lanes that hold no input rows (the helper's upper lanes, lanes past the strip) load the first rows of the same column instead
    // of being masked: one extra cache line per column, and no `if` -- hence no phi and no register copy -- around the
    // asynchronous loads (wl_dev.h: gload16_if)
    const T *base = a.src + (loader ? row : 0);
    const int64_t kbase = j0 >> 1;

    T2 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T2{0.0, 0.0};
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        gload16<WL_P_PAIR64_LD != 0>(ring[c], base + jc * a.lds);
    }
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);

    auto step = [&](const int t, const int u, const bool prefetch, const bool full, const bool produce) __attribute__((always_inline)) {
        if (prefetch) {                                    // (compile-time)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                gload16<WL_P_PAIR64_LD != 0>(ring[(2 * u + R - 2 + e) % R], base + jc * a.lds);
            }
        }
        if (produce) {
            // loads only in the count (see wl_dev.h)
            if (prefetch) wait_vm<2 * PFD>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
            else wait_vm<0>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
        }
        T2 *const w1 = x1 + (t & 1) * ROWS1;
        const bool sonly = !full;
        if (produce) {
            // ---- level l, dim-2 pass on the lane's two rows ----
            T2 sa = a.tp.h[0] * ring[(2 * u) % R];
#pragma unroll
            for (int m = 1; m < F; ++m) sa = sa + a.tp.h[m] * ring[(2 * u + m) % R];
            T2 da = T2{0.0, 0.0};
            if (!sonly) {
                da = gq(F - 1) * ring[(2 * u) % R];
#pragma unroll
                for (int m = 1; m < F; ++m) da = da + gq(F - 1 - m) * ring[(2 * u + m) % R];
            }
            w1[2 * lp] = T2{sa.x, da.x};
            w1[2 * lp + 1] = T2{sa.y, da.y};
        }
        wg_lds_sync(true);
        __builtin_amdgcn_sched_barrier(0);
        // ---- helper, even steps: level-(l+1) dim-2 pass of the halo rows 64 W + i (lanes i = 0..7) ----
        if (helper && !(u & 1) && t >= F) {
            if (lp < NPL + 8) {
                T s2, d2;
#pragma unroll
                for (int m = 0; m < F; ++m) {
                    const T xm = ll1[((t - F + m) & (NSLOT - 1)) * RL + lp];
                    if (m == 0) { s2 = a.tp.h[0] * xm; d2 = gq(F - 1) * xm; }
                    else { s2 = s2 + a.tp.h[m] * xm; d2 = d2 + gq(F - 1 - m) * xm; }
                }
                x2[lp] = T2{s2, d2};
            }
        }
        if (!produce) return;
        T *const slot = ll1 + (t & (NSLOT - 1)) * RL + lp;
        if (sonly || helper) {
            // ---- approximation only: ss row ko from window rows 2L' .. 2L'+F-1 ----
            T p0 = a.tp.h[0] * w1[2 * lp].x;
#pragma unroll
            for (int m = 1; m < F; ++m) p0 = p0 + a.tp.h[m] * w1[2 * lp + m].x;
            if (lp < NPL + 8) *slot = p0;
            return;
        }
        // ---- level l, dim-1 pass: window rows 2L' .. 2L'+9 as {A, B} pairs ----
        T2 E[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) E[c] = w1[2 * lp + c];
        T2 P = a.tp.h[0] * E[0];                       // {ss, sd} of row ko
#pragma unroll
        for (int m = 1; m < F; ++m) P = P + a.tp.h[m] * E[m];
        T2 Q = gq(F - 1) * E[10 - F];              // {ds, dd} of row kod
#pragma unroll
        for (int m = F - 2; m >= 0; --m) Q = Q + gq(m) * E[9 - m];
        *slot = P.x;                                   // approximation column kbase + t -> ring
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        // even lane: ds rows kod, kod+1 of column k;  odd lane: sd rows ko-1, ko and dd rows kod-1, kod of column kd
        const T rA = from_partner(odd ? Q.x : P.y);
        const T rB = from_partner(Q.y);
        T *const ck = yb + k * a.ldy, *const ckd = yb + (nxj + kd) * a.ldy;      // (uniform)
        if (!odd) {
            store_pol<WL_P_PAIR64_ST1>(reinterpret_cast<T2 *>(ck + (hmi + kod)), T2{Q.x, rA});
        } else {
            store_pol<WL_P_PAIR64_ST1>(reinterpret_cast<T2 *>(ckd + (ko - 1)), T2{rA, P.y});
            store_pol<WL_P_PAIR64_ST1>(reinterpret_cast<T2 *>(ckd + (hmi + kod - 1)), T2{rB, Q.y});
        }
    };

    int t0 = 0;
    for (; t0 < S_own; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u, true, true, true);
    }
    // 2 taps: none of the steps below waits for a load, yet the last PFD steps' prefetches are still on their way
    if constexpr (F == 2) drain_ring(ring);
#pragma unroll
    for (int u = 0; u < F; ++u) step(t0 + u, u, u + PFD < F - 2, false, u < F - 2);
}

// ------------------------------------------------------------------------------------------
bool fwd2d_pair64_ok(int F, int64_t ms, int64_t ns)
{
    if (F < 2 || F > 10 || (F & 1)) return false;
    if (ms >= ((int64_t)1 << 29)) return false;
    // rows: exact strips of 256 (W = 2) or 512 (W = 4); columns: level-(l+1) columns come in pairs of steps, chunks of 32
    return ms >= 256 && (ms % 256) == 0 && ns >= 64 && (ns % 32) == 0;
}

template <int F, int W>
static hipError_t launch_pair64_fw(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds, double *y,
                                   int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    Pair2DArgs64<F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.nstrips = (int)(ms / (128 * W));
    int TJ = (int)opt("WL_TJ2", 128);
    if (TJ < 32) TJ = 32;
    TJ &= ~31;
    auto nwgs = [&](int tj) { return (int64_t)a.nstrips * ((ns + tj - 1) / tj); };
    // (8192^2 db4, levels 1-2: chunks of 128 columns 267 us, 64: 276, 256: 283; strips of 512 rows (W = 4): 329; two launches: 274)
    while (TJ > 32 && (TJ % 64) == 0 && nwgs(TJ) < (int64_t)cu_count * opt("WL_PAIR_WG_PER_CU", 3)) TJ >>= 1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.rev = (!lvl1 && opt("WL_REVERSE", 1)) ? 1 : 0;
    a.prio = opt("WL_PAIR64_PRIO", 0) != 0 ? 1 : 0;       // (Float64: raised helper / level-(l+1) priority measured +1 %: off)
    a.tp = shrink<double, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    if (lvl1) hipLaunchKernelGGL((k_fwd2d_pair64<F, W, 1>), dim3(nwg), dim3(64 * (W + 2)), 0, st, a);
    else hipLaunchKernelGGL((k_fwd2d_pair64<F, W, 0>), dim3(nwg), dim3(64 * (W + 2)), 0, st, a);
    return hipGetLastError();
}

template <int F>
static hipError_t launch_pair64_f(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds, double *y,
                                  int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    int W = (int)opt("WL_PAIR_W64", 2);
    if (W != 2 && W != 4) W = 2;
    if ((ms % (128 * W)) != 0) W = 2;
    if (W == 4) return launch_pair64_fw<F, 4>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    return launch_pair64_fw<F, 2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
}

hipError_t fwd2d_pair64_launch(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds, double *y,
                               int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    switch (taps.F) {
    case 2: return launch_pair64_f<2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 4: return launch_pair64_f<4>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 6: return launch_pair64_f<6>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 8: return launch_pair64_f<8>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 10: return launch_pair64_f<10>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
