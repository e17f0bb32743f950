// wl_fwd.hip -- forward filter-bank DWT: the level loop and its CDNA4 fast paths.
//
// Hot path of the build (BASELINE.json north_star): one fused kernel per level.
//
//   k_fwd2d_stream  2-D level, rows (dim 2) + columns (dim 1) fused, register streaming.
//                   One wave owns a strip of 64*RPL rows (RPL contiguous rows per lane, one
//                   16-byte load per lane per column => 1 KiB per wave-instruction, coalesced
//                   along dim 1) and marches along dim 2 with an F-column sliding window held
//                   in VGPRs (the dim-2 pass never leaves registers).  The dim-1 pass takes
//                   the neighbouring rows from adjacent lanes with wave shuffles; strips
//                   overlap by 16 rows instead of exchanging halos through LDS, so there is no
//                   barrier and no LDS traffic at all.  The detail column computed from a
//                   window is d[k+(F-2)/2] (same F inputs as s[k]), which halves the halo.
//                   HBM traffic: each level's block is read once and written once.
//   k_fwd2d_stream2 TWO 2-D levels per launch (blocks >= 4096^2, Float32): the level-l approximation stays in an
//                   8-slot register ring and level l+1 runs on it every second step (see the kernel).
//   k_fwd1d_multi   up to 4 fused 1-D levels per pass over HBM through LDS tiles (lines, batched columns);
//   k_fwd2d_multi   two 2-D levels of a small (<= 128^2) block per launch (LDS tiles with recomputed halo).
//   k_fwd1d_stream  1-D level (also one line per blockIdx.y: batched columns / WPT segments):
//                   8 samples per lane, halo from the two neighbouring lanes by shuffle.
//   k_tail_fwd      all remaining levels of a small block (<= 16 Ki f32 / 8 Ki f64 elements) in
//                   ONE workgroup with the block resident in LDS (true periodic indexing, so
//                   lines shorter than the filter are handled) -- removes ~2 launches per level
//                   for the deep levels of a full-depth transform.
//
// Long filters (12..24 taps) and 3-D levels: wl_axis.hip.
// All of them evaluate the closed forms of wl_internal.h in the reference's summation order
// with separate multiply/add roundings: results are bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"


namespace wl {


// One level of the QMF pair along the lane axis.  v[] holds this lane's PER consecutive
// samples of the line; sample offsets outside [0, PER) come from lane + floordiv(off, PER).
// Produces PER/2 (s, d) pairs: pair q uses s: offsets 2q..2q+F-1, d: offsets 2q+2-F..2q+1.
template <typename T, int F, int PER>
__device__ __forceinline__ void lane_axis_pair(const T (&v)[PER], const TapsF<T, F> &tp,
                                               T (&so)[PER / 2], T (&dO)[PER / 2])
{
    constexpr int LO = -(F - 2), HI = PER + F - 2;
    T ext[HI - LO];
#pragma unroll
    for (int e = 0; e < PER; ++e) ext[e - LO] = v[e];
    {   // offsets >= PER: repeated shifts towards lower lanes
        T cur[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) cur[e] = v[e];
#pragma unroll
        for (int base = PER; base < HI; base += PER) {
#pragma unroll
            for (int e = 0; e < PER; ++e)
                if (base + e < HI) {            // only elements still needed at this or a later distance
                    cur[e] = from_next(cur[e]);
                    ext[base + e - LO] = cur[e];
                }
        }
    }
    {   // offsets < 0: repeated shifts towards higher lanes
        T cur[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) cur[e] = v[e];
#pragma unroll
        for (int base = -PER; base + PER > LO; base -= PER) {
#pragma unroll
            for (int e = PER - 1; e >= 0; --e)
                if (base + e >= LO) {
                    cur[e] = from_prev(cur[e]);
                    ext[base + e - LO] = cur[e];
                }
        }
    }
#pragma unroll
    for (int q = 0; q < PER / 2; ++q) {
        T s = tp.h[0] * ext[2 * q - LO];
#pragma unroll
        for (int m = 1; m < F; ++m) s = s + tp.h[m] * ext[2 * q + m - LO];
        T d = tp.g[F - 1] * ext[2 * q + 1 - (F - 1) - LO];
#pragma unroll
        for (int m = F - 2; m >= 0; --m) d = d + tp.g[m] * ext[2 * q + 1 - m - LO];
        so[q] = s;
        dO[q] = d;
    }
}

// Packed variant for the 2-D kernel: v[] holds PER pairs {a, b} (two independent lines, e.g. the
// dim-2 scaling and detail results of the same row); both lines get the same QMF pair along the
// lane axis.  Written on 2-vectors so that f32 maps onto v_pk_mul_f32 / v_pk_add_f32 with the tap
// broadcast from an SGPR -- no operand shuffling moves.
//   P[q] = { s-of-a, s-of-b } = sum_m asc  h[m] * E[2q+m]
//   Q[q] = { d-of-a, d-of-b } = sum_m desc g[m] * E[2q+1-m]
template <typename T, int F, int PER>
__device__ __forceinline__ void lane_axis_pair2(const typename VecOf<T, 2>::type (&v)[PER], const TapsF<T, F> &tp,
                                                typename VecOf<T, 2>::type (&P)[PER / 2],
                                                typename VecOf<T, 2>::type (&Q)[PER / 2])
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int LO = -(F - 2), HI = PER + F - 2;
    T2 ext[HI - LO];
#pragma unroll
    for (int e = 0; e < PER; ++e) ext[e - LO] = v[e];
    {
        T2 cur[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) cur[e] = v[e];
#pragma unroll
        for (int base = PER; base < HI; base += PER) {
#pragma unroll
            for (int e = 0; e < PER; ++e)
                if (base + e < HI) {
                    cur[e].x = from_next(cur[e].x);
                    cur[e].y = from_next(cur[e].y);
                    ext[base + e - LO] = cur[e];
                }
        }
    }
    {
        T2 cur[PER];
#pragma unroll
        for (int e = 0; e < PER; ++e) cur[e] = v[e];
#pragma unroll
        for (int base = -PER; base + PER > LO; base -= PER) {
#pragma unroll
            for (int e = PER - 1; e >= 0; --e)
                if (base + e >= LO) {
                    cur[e].x = from_prev(cur[e].x);
                    cur[e].y = from_prev(cur[e].y);
                    ext[base + e - LO] = cur[e];
                }
        }
    }
#pragma unroll
    for (int q = 0; q < PER / 2; ++q) {
        T2 s = tp.h[0] * ext[2 * q - LO];
#pragma unroll
        for (int m = 1; m < F; ++m) s = s + tp.h[m] * ext[2 * q + m - LO];
        T2 d = tp.g[F - 1] * ext[2 * q + 1 - (F - 1) - LO];
#pragma unroll
        for (int m = F - 2; m >= 0; --m) d = d + tp.g[m] * ext[2 * q + 1 - m - LO];
        P[q] = s;
        Q[q] = d;
    }
}

// ==========================================================================================
// 2-D fused level
template <typename T, int F>
struct Fwd2DArgs {
    const T *src; int64_t lds;      // input block (ms x ns), leading dimension
    T *y; int64_t ldy;              // output array (detail quadrants, and LL when ll == nullptr)
    T *ll; int64_t ldll;            // approximation buffer for the next level (or nullptr)
    int64_t ms, ns;
    int TJ;                         // input columns per chunk (even)
    int nt;                         // bit 2: walk this XCD's range backwards
    int nstrips, nchunks;
    // batch of independent blocks (blockIdx.y; the planes of a 3-D level): element strides; only the first nll
    // blocks send their approximation quadrant to ll, the others write it into y
    int64_t bs_src, bs_y, bs_ll; int nll;
    TapsF<T, F> tp;
};

template <typename T, int F, int RPL, int LVL1>
__global__ void __launch_bounds__(64) k_fwd2d_stream(Fwd2DArgs<T, F> a)
{
    constexpr int ML = 8 / RPL;            // margin lanes on each side (8 rows)
    constexpr int VR = 64 * RPL - 16;      // rows of output responsibility per wave
    constexpr int NO = RPL / 2;            // output rows per lane per subband
    constexpr int SH = (F - 2) / 2;        // detail-column shift
    const int lane = threadIdx.x & 63;

    // XCD-aware bijective remap: consecutive logical ids (strip fastest) share an XCD's L2 -- and the
    // WPB waves of a workgroup (adjacent strips of one chunk) a CU's L1 -- so the 16-row overlap of
    // neighbouring strips is a cache hit.
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t lwg = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    uint32_t logical = lwg * wpb + (threadIdx.x >> 6);
    if (logical >= (uint32_t)(a.nstrips * a.nchunks)) return;
    // levels >= 2 walk each XCD's range backwards: the columns the previous level wrote last are the
    // ones most likely still resident in that XCD's L2 / the Infinity Cache
    if (a.nt & 4) {
        const uint32_t cnt = q8 + (xcd < r8 ? 1u : 0u), first = xcd * q8 + (xcd < r8 ? xcd : r8);
        logical = first + (cnt - 1 - (logical - first));
    }
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1, hm = ms >> 1;
    const int64_t gi = (int64_t)strip * VR + (int64_t)(lane - ML) * RPL;   // first row of this lane
    int64_t row = gi;
    if (row < 0) row += ms;
    if (row >= ms) row -= ms;
    const bool valid = (lane >= ML) && (lane < 64 - ML) && (gi < ms);
    const int64_t ko = gi >> 1;

    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S = (int)((jend - j0) >> 1);
    const T *base = a.src + (int64_t)blockIdx.y * a.bs_src + row;

    // Column ring of R = 16 register slots: input column c (relative to j0) lives in slot c % R.
    // Step t consumes columns 2t .. 2t+F-1 and refills the two slots freed by step t-1 with
    // columns 2t+R-2, 2t+R-1, i.e. loads run PFD = (R-F)/2 steps ahead of their use.  The loop is
    // unrolled by R/2 = 8 steps so every slot index is a compile-time constant: no register
    // rotation moves, no waits on just-issued loads.  (S % 8 == 0 is guaranteed by the launcher.)
    constexpr int R = 16, U = R / 2, PFD = (R - F) / 2;
    T ring[R][RPL];
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        vload<T, RPL>(base + jc * a.lds, ring[c]);
    }

    T *const yl = a.y + (int64_t)blockIdx.y * a.bs_y + ko;
    const bool to_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    T *const llp = to_ll ? (a.ll + (int64_t)blockIdx.y * a.bs_ll + ko) : yl;
    const int64_t ldl = to_ll ? a.ldll : a.ldy;
    const int64_t kbase = j0 >> 1;

    // u = t % U is a compile-time constant at every call site
    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                vload<T, RPL>(base + jc * a.lds, ring[(2 * u + R - 2 + e) % R]);
            }
        }
        // dim-2 pass in registers on row pairs (adjacent registers of the 16-byte loads):
        // A2[p] = scaling, B2[p] = detail of rows 2p, 2p+1 for this column pair
        typedef typename VecOf<T, 2>::type T2;
        T2 A2[RPL / 2], B2[RPL / 2];
#pragma unroll
        for (int pr = 0; pr < RPL / 2; ++pr) {
            T2 x0 = T2{ring[(2 * u) % R][2 * pr], ring[(2 * u) % R][2 * pr + 1]};
            T2 sa = a.tp.h[0] * x0;
            T2 da = a.tp.g[F - 1] * x0;
#pragma unroll
            for (int m = 1; m < F; ++m) {
                T2 xm = T2{ring[(2 * u + m) % R][2 * pr], ring[(2 * u + m) % R][2 * pr + 1]};
                sa = sa + a.tp.h[m] * xm;
                da = da + a.tp.g[F - 1 - m] * xm;
            }
            A2[pr] = sa;
            B2[pr] = da;
        }
        // re-pair as RD[q] = { scaling, detail } of row q for the lane-axis pass
        T2 RD[RPL];
#pragma unroll
        for (int pr = 0; pr < RPL / 2; ++pr) {
            RD[2 * pr] = T2{A2[pr].x, B2[pr].x};
            RD[2 * pr + 1] = T2{A2[pr].y, B2[pr].y};
        }
        // dim-1 pass across lanes: P[q] = {ss, sd}, Q[q] = {ds, dd} for output row ko + q
        T2 P[NO], Q[NO];
        lane_axis_pair2<T, F, RPL>(RD, a.tp, P, Q);
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        if constexpr (RPL == 4) {
            // Lane pairs (2i, 2i+1) own output rows ko..ko+1 and ko+2..ko+3 of the same columns.  They
            // swap halves so that the even lane stores rows ko..ko+3 of the two s_j subbands' column k
            // and the odd lane rows ko-2..ko+1 of the d_j subbands' column kd: 16-byte stores instead
            // of 8-byte ones (measured +8 % on the store-side of this access pattern).
            const bool odd = (lane & 1) != 0;
            T rP[NO], rQ[NO];
#pragma unroll
            for (int qq = 0; qq < NO; ++qq) {
                rP[qq] = from_partner(odd ? P[qq].x : P[qq].y);
                rQ[qq] = from_partner(odd ? Q[qq].x : Q[qq].y);
            }
            T vP[4], vQ[4];
#pragma unroll
            for (int qq = 0; qq < NO; ++qq) {
                vP[qq] = odd ? rP[qq] : P[qq].x;  vP[NO + qq] = odd ? P[qq].y : rP[qq];
                vQ[qq] = odd ? rQ[qq] : Q[qq].x;  vQ[NO + qq] = odd ? Q[qq].y : rQ[qq];
            }
            if (valid) {
                // even: LL / ds of column k at row ko;   odd: sd / dd of column kd at row ko-2
                T *pP = odd ? (yl - NO + (nxj + kd) * a.ldy) : (llp + k * ldl);
                T *pQ = odd ? (yl - NO + (nxj + kd) * a.ldy + hm) : (yl + k * a.ldy + hm);
                vstore<T, 4>(pP, vP);
                vstore<T, 4>(pQ, vQ);
            }
        } else {
            if (valid) {
                T ss[NO], ds[NO], sd[NO], dd[NO];
#pragma unroll
                for (int qq = 0; qq < NO; ++qq) { ss[qq] = P[qq].x; sd[qq] = P[qq].y; ds[qq] = Q[qq].x; dd[qq] = Q[qq].y; }
                vstore<T, NO>(llp + k * ldl, ss);
                vstore<T, NO>(yl + k * a.ldy + hm, ds);
                vstore<T, NO>(yl + (nxj + kd) * a.ldy, sd);
                vstore<T, NO>(yl + (nxj + kd) * a.ldy + hm, dd);
            }
        }
    };

    int t0 = 0;
    for (; t0 < S - U; t0 += U) {            // steady state: every prefetch is inside the chunk's range
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u, true);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) step(t0 + u, u, u < U - PFD);   // last group: stop prefetching PFD steps early
}


// ==========================================================================================
// 2-D fused PAIR of levels (l, l+1): same register-streaming scheme as k_fwd2d_stream, but the
// level-l approximation never goes to HBM -- each lane keeps its two LL rows of the last eight LL
// columns in an 8-slot register ring and every second step runs level l+1 on them (dim-2 pass on
// the ring, dim-1 pass across lanes with 3-deep DPP chains).  Traffic for the two levels: read the
// level-l block once, write its three detail quadrants and the four level-(l+1) quadrants
// (4.19 -> 3.31 bytes per byte of level-l input: the LL write + re-read disappear), one launch less.
// Cost: a wave owns 208 rows instead of 240 (lanes 6..57) and a chunk runs 8 extra steps.
template <typename T, int F>
struct Fwd2D2Args {
    const T *src; int64_t lds;
    T *y; int64_t ldy;
    T *ll; int64_t ldll;            // level-(l+1) approximation: next stage's input buffer, or y itself
    int64_t ms, ns;                 // level-l block
    int TJ;                         // owned input columns per chunk (multiple of 16)
    int nstrips, nchunks;
    int rev;
    TapsF<T, F> tp;
};

template <typename T, int F, int LVL1>
__global__ void __launch_bounds__(64) k_fwd2d_stream2(Fwd2D2Args<T, F> a)
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int RPL = 4, ML = 6, VR = (64 - 2 * ML) * RPL, NO = 2, SH = (F - 2) / 2;
    constexpr int R = 16, U = 8, PFD = (R - F) / 2;
    const int lane = threadIdx.x;
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1u : 0u);
    uint32_t logical = first + (b >> 3);
    if (a.rev) logical = first + (cnt - 1 - (logical - first));
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1, hm = ms >> 1, nxj2 = ns >> 2, hm2 = ms >> 2;
    const int64_t gi = (int64_t)strip * VR + (int64_t)(lane - ML) * RPL;
    int64_t row = gi;
    if (row < 0) row += ms;
    if (row >= ms) row -= ms;
    const bool own = (lane >= ML) && (lane < 64 - ML) && (gi < ms);
    const int64_t ko = gi >> 1, ko2 = gi >> 2;
    const bool odd = (lane & 1) != 0;

    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S_own = (int)((jend - j0) >> 1);         // level-l steps whose outputs this chunk owns (multiple of 8)
    const int S = S_own + U;                           // + 8 steps that only feed level l+1
    const T *base = a.src + row;

    T ring[R][RPL];
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        vload<T, RPL>(base + jc * a.lds, ring[c]);
    }
    T2 ring2[U];                                       // LL rows (ko, ko+1) of LL column k in slot k % 8
    T *const yl = a.y + ko;
    const int64_t kbase = j0 >> 1;                     // multiple of 8
    const int64_t kbase2 = j0 >> 2;

    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                vload<T, RPL>(base + jc * a.lds, ring[(2 * u + R - 2 + e) % R]);
            }
        }
        // ---- level l: dim-2 pass on row pairs, dim-1 pass across lanes ----
        T2 A2[2], B2[2];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            T2 x0 = T2{ring[(2 * u) % R][2 * pr], ring[(2 * u) % R][2 * pr + 1]};
            T2 sa = a.tp.h[0] * x0;
            T2 da = a.tp.g[F - 1] * x0;
#pragma unroll
            for (int m = 1; m < F; ++m) {
                T2 xm = T2{ring[(2 * u + m) % R][2 * pr], ring[(2 * u + m) % R][2 * pr + 1]};
                sa = sa + a.tp.h[m] * xm;
                da = da + a.tp.g[F - 1 - m] * xm;
            }
            A2[pr] = sa;
            B2[pr] = da;
        }
        T2 RD[RPL];
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            RD[2 * pr] = T2{A2[pr].x, B2[pr].x};
            RD[2 * pr + 1] = T2{A2[pr].y, B2[pr].y};
        }
        T2 P[NO], Q[NO];                               // P[q] = {ss, sd}, Q[q] = {ds, dd} of output row ko + q
        lane_axis_pair2<T, F, RPL>(RD, a.tp, P, Q);
        ring2[u] = T2{P[0].x, P[1].x};                 // LL column kbase + t
        if (t < S_own) {
            const int64_t k = kbase + t;
            int64_t kd = k + SH;
            if (kd >= nxj) kd -= nxj;
            // even lane: ds rows ko..ko+3 of column k;  odd lane: sd and dd rows ko-2..ko+1 of column kd
            T rA[NO], rB[NO];
#pragma unroll
            for (int q = 0; q < NO; ++q) {
                rA[q] = from_partner(odd ? Q[q].x : P[q].y);
                rB[q] = from_partner(Q[q].y);
            }
            T v0[4], v1[4];
#pragma unroll
            for (int q = 0; q < NO; ++q) {
                v0[q] = odd ? rA[q] : Q[q].x;  v0[NO + q] = odd ? P[q].y : rA[q];
                v1[q] = rB[q];                 v1[NO + q] = Q[q].y;
            }
            if (own) {
                T *p0 = odd ? (yl - NO + (nxj + kd) * a.ldy) : (yl + k * a.ldy + hm);
                vstore<T, 4>(p0, v0);
                if (odd) vstore<T, 4>(yl - NO + (nxj + kd) * a.ldy + hm, v1);
            }
        }
        // ---- level l+1: every second step, on the LL ring (window = LL columns k-7 .. k) ----
        if ((u & 1) && t >= U - 1) {
            T2 sa = a.tp.h[0] * ring2[(u + 1) % U];
            T2 da = a.tp.g[F - 1] * ring2[(u + 1) % U];
#pragma unroll
            for (int m = 1; m < F; ++m) {
                sa = sa + a.tp.h[m] * ring2[(u + 1 + m) % U];
                da = da + a.tp.g[F - 1 - m] * ring2[(u + 1 + m) % U];
            }
            T2 RD2[2];
            RD2[0] = T2{sa.x, da.x};
            RD2[1] = T2{sa.y, da.y};
            T2 P2[1], Q2[1];                           // P2 = {ss2, sd2}, Q2 = {ds2, dd2} of row ko2
            lane_axis_pair2<T, F, 2>(RD2, a.tp, P2, Q2);
            if (t < S_own + U - 1) {
                const int64_t k2 = kbase2 + ((t - (U - 1)) >> 1);
                int64_t kd2 = k2 + SH;
                if (kd2 >= nxj2) kd2 -= nxj2;
                const T rP = from_partner(odd ? P2[0].x : P2[0].y);
                const T rQ = from_partner(odd ? Q2[0].x : Q2[0].y);
                T w0[2], w1[2];
                w0[0] = odd ? rP : P2[0].x;  w0[1] = odd ? P2[0].y : rP;
                w1[0] = odd ? rQ : Q2[0].x;  w1[1] = odd ? Q2[0].y : rQ;
                if (own) {
                    // even lane: ss2 / ds2 rows ko2, ko2+1 of column k2;  odd lane: sd2 / dd2 rows ko2-1, ko2 of column kd2
                    T *p0 = odd ? (a.y + (ko2 - 1) + (nxj2 + kd2) * a.ldy) : (a.ll + ko2 + k2 * a.ldll);
                    T *p1 = odd ? (a.y + (ko2 - 1) + (nxj2 + kd2) * a.ldy + hm2) : (a.y + ko2 + k2 * a.ldy + hm2);
                    vstore<T, 2>(p0, w0);
                    vstore<T, 2>(p1, w1);
                }
            }
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

// ==========================================================================================
// 1-D level (one line per blockIdx.y)
template <typename T, int F>
struct Fwd1DArgs {
    const T *src; int64_t src_ls;   // line stride of the source
    T *sdst; int64_t s_ls;          // approximation destination
    T *ddst; int64_t d_ls;          // detail destination
    int64_t n;                      // line length (multiple of 8, >= 512)
    int64_t ntiles;                 // tiles of 248 pairs
    TapsF<T, F> tp;
};

template <typename T, int F, int LVL1>
__global__ void __launch_bounds__(256) k_fwd1d_stream(Fwd1DArgs<T, F> a)
{
    constexpr int SPL = 8, VP = 62 * 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t n = a.n, nx = n >> 1;
    const T *src = a.src + (int64_t)blockIdx.y * a.src_ls;
    T *sdst = a.sdst + (int64_t)blockIdx.y * a.s_ls;
    T *ddst = a.ddst + (int64_t)blockIdx.y * a.d_ls;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k0 = tile * VP + (int64_t)(lane - 1) * 4;     // first pair of this lane
        int64_t gs = 2 * k0;
        if (gs < 0) gs += n;
        if (gs >= n) gs -= n;
        T v[SPL];
        vload16<T, SPL>(src + gs, v);
        T so[4], dO[4];
        lane_axis_pair<T, F, SPL>(v, a.tp, so, dO);
        if (lane >= 1 && lane <= 62 && k0 < nx) {
            vstore16<T, 4>(sdst + k0, so);
            vstore16<T, 4>(ddst + k0, dO);
        }
    }
}


// ==========================================================================================
// 1-D multi-level tile kernel: NL (<= 8; 4 on lines that do not fit L2) consecutive levels of a line in ONE pass over HBM.
// A workgroup owns TS input samples, loads them plus a halo of H0 = (F-2)(2^NL - 1) samples on
// each side into LDS (periodic wrap resolved while staging), then runs the levels LDS -> LDS:
// at level t the local array covers the owned range widened by H_t = (F-2)(2^(NL-t) - 1), so pair i
// of level t reads the fixed window [2i, 2i + 2F - 2) of level t-1 -- no index arithmetic, no wrap,
// and two adjacent pairs are four aligned ds_read_b128.  Details of the owned range stream to their
// final place in y, the last approximation goes to the next stage.  Traffic: read n + write n for NL
// levels (the level-by-level scheme moves (4 - 2^(2-NL)) n); halo re-reads are 2*H0/TS (~2 %).
template <typename T, int F>
struct Multi1DArgs {
    const T *src; int64_t src_ls;
    T *y; int64_t y_ls;             // detail level t (1..NL) goes to y[(n >> t) + k]
    T *sdst; int64_t s_ls;          // approximation after NL levels
    int64_t n;                      // line length entering this stage
    int NL, TS;
    TapsF<T, F> tp;
};

// the NL levels of one staged tile, LDS -> LDS (bufA holds the tile + halo); ends with a barrier after the last level
template <typename T, int F>
__device__ __forceinline__ void multi1d_levels(const Multi1DArgs<T, F> &a, const int (&H)[10], const int64_t own0, const int own_len,
                                               T *bufA, T *bufB, T *y)
{
    constexpr int VEC = 16 / sizeof(T);
    const int tid = threadIdx.x;
    const int64_t n = a.n;
    const int NL = a.NL;
    T *Ain = bufA, *Aout = bufB;
    // g[m] = (-1)^m h[m] exactly: only the scaling taps occupy SGPRs (a detail term multiplies by the negated tap)
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    // (Tried in round 3 and dropped: ONE buffer, each level writing its approximation over the input it has consumed after an
    //  extra barrier -- 17 KiB of LDS, eight resident workgroups instead of six: 8192 x 2^16 1107 us against 997.)
    // A thread owns PPT = 16 / sizeof(T) consecutive pairs: one 16-byte window row per ds_read_b128, one 16-byte store per
    // output stream.  (Round 2 gave a thread two pairs: Float32 details left in 8-byte stores, half the width, and the 2 F
    // window values read for two pairs are enough for four.)
    constexpr int PPT = VEC;
    constexpr int NWIN = ((2 * PPT + 2 * F - 4) + VEC - 1) / VEC * VEC;
    const int lane = tid & 63;
    for (int t = 1; t <= NL; ++t) {
        const int ownt = own_len >> t;
        const int Lout = ownt + 2 * H[t];
        const int64_t k0 = own0 >> t;                 // global index of the first owned pair
        T *dd = y + (n >> t);
        const bool lastlev = (t == NL);
        T *sg = a.sdst + (int64_t)blockIdx.y * a.s_ls;
        // global position of local pair i is k0 + i - H[t]; k0 is a multiple of PPT (tiles of >= 4096 bytes), so a thread's
        // group starts PPT - mis pairs before an aligned 16-byte group, mis = H[t] mod PPT (0, or 2 for Float32)
        const int mis = H[t] & (PPT - 1);
        const bool quads = (ownt % PPT) == 0 && (k0 % PPT) == 0;
        // wave-uniform trip count: the lane exchange below needs the neighbour lane inside the loop
        for (int i0 = PPT * (tid - lane); i0 < Lout; i0 += PPT * 256) {
            const int i = i0 + PPT * lane;
            T xv[NWIN];
            vload16<T, NWIN>(Ain + 2 * i, xv);        // window of pairs i .. i+PPT-1 (2i is a multiple of 2 PPT)
            T so[PPT], dO[PPT];
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                T sv = a.tp.h[0] * xv[2 * q + F - 2];
#pragma unroll
                for (int m = 1; m < F; ++m) sv = sv + a.tp.h[m] * xv[2 * q + F - 2 + m];
                T dv = gq(F - 1) * xv[2 * q];
#pragma unroll
                for (int m = F - 2; m >= 0; --m) dv = dv + gq(m) * xv[2 * q + F - 1 - m];
                so[q] = sv;
                dO[q] = dv;
            }
            if (!lastlev && i < Lout) vstore16<T, PPT>(Aout + i, so);      // (the tail of the last group lands in the padding)
            const int io = i - H[t];                  // pair i + q is owned iff 0 <= io + q < ownt
            if (quads && mis == 0) {
                if (io >= 0 && io + PPT <= ownt) {
                    stg_pol<WL_P_M1D_ST != 0, T, PPT>(dd + k0 + io, dO);
                    if (lastlev) vstore16<T, PPT>(sg + k0 + io, so);
                }
            } else if (quads && PPT == 4 && mis == 2) {
                // pairs io+2, io+3 open an aligned group that the next lane's first two pairs complete
                constexpr int Q2 = (PPT == 4) ? 2 : 0, Q3 = (PPT == 4) ? 3 : 1;      // (Float32 only; keeps the Float64 instance well-formed)
                T q4[4] = {dO[Q2], dO[Q3], from_next(dO[0]), from_next(dO[1])};
                T s4[4] = {so[Q2], so[Q3], from_next(so[0]), from_next(so[1])};
                const int ia = io + 2;
                if (lane != 63) {
                    if (ia >= 0 && ia + 4 <= ownt) {
                        stg_pol<WL_P_M1D_ST != 0, T, 4>(dd + k0 + ia, q4);
                        if (lastlev) vstore16<T, 4>(sg + k0 + ia, s4);
                    }
                } else if (ia >= 0 && ia + 2 <= ownt) {   // (the completing lane belongs to another wave: two 8-byte halves)
                    vstore<T, 2>(dd + k0 + ia, reinterpret_cast<const T(&)[2]>(q4[0]));
                    if (lastlev) vstore<T, 2>(sg + k0 + ia, reinterpret_cast<const T(&)[2]>(s4[0]));
                }
                if (lane == 0 && io >= 0 && io + 2 <= ownt) {
                    vstore<T, 2>(dd + k0 + io, reinterpret_cast<const T(&)[2]>(dO[0]));
                    if (lastlev) vstore<T, 2>(sg + k0 + io, reinterpret_cast<const T(&)[2]>(so[0]));
                }
            } else {
                // short / oddly placed tiles: element by element
#pragma unroll
                for (int q = 0; q < PPT; ++q)
                    if (io + q >= 0 && io + q < ownt && i + q < Lout) {
                        dd[k0 + io + q] = dO[q];
                        if (lastlev) sg[k0 + io + q] = so[q];
                    }
            }
        }
        lds_barrier();
        T *tmp = Ain; Ain = Aout; Aout = tmp;
    }
}

template <typename T, int F, int LVL1>
__global__ void __launch_bounds__(256) k_fwd1d_multi(Multi1DArgs<T, F> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int VEC = 16 / sizeof(T);
    const int tid = threadIdx.x;
    const int64_t n = a.n;
    const int NL = a.NL;
    int H[10];
    H[NL] = 0;
    for (int t = NL; t >= 1; --t) H[t - 1] = 2 * H[t] + (F - 2);
    const int64_t own0 = (int64_t)blockIdx.x * a.TS;
    const int own_len = (int)((own0 + a.TS <= n) ? a.TS : (n - own0));
    const int lenA = own_len + 2 * H[0];
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + ((a.TS + 2 * H[0] + 7) & ~7) + 16;
    const T *src = a.src + (int64_t)blockIdx.y * a.src_ls;
    T *y = a.y + (int64_t)blockIdx.y * a.y_ls;

    // stage: A[j] = x[(own0 - H0 + j) mod n], 16-byte chunks at aligned global positions
    {
        const int64_t start = own0 - H[0];
        const int r = (int)(((start % VEC) + VEC) % VEC);
        const int64_t astart = start - r;
        const int nch = (lenA + r + VEC - 1) / VEC;
        // all of a thread's chunk loads are issued before the first LDS write (up to 10 in flight)
        constexpr int UL = 10;
        for (int c0 = tid; c0 < nch; c0 += UL * 256) {
            T v[UL][VEC];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int c = c0 + u * 256;
                if (c < nch) {
                    int64_t gidx = astart + (int64_t)c * VEC;
                    if (gidx < 0) gidx += n;
                    if (gidx >= n) gidx -= n;
                    ldg_pol<WL_P_M1D_LD != 0, T, VEC>(src + gidx, v[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int c = c0 + u * 256;
                if (c < nch) {
                    const int j0 = c * VEC - r;
                    if (r == 0 && j0 + VEC <= lenA) {
                        vstore16<T, VEC>(bufA + j0, v[u]);                                   // one 16-byte LDS write
                    } else if (VEC == 4 && r == 2 && j0 >= 0 && j0 + VEC <= lenA) {
                        vstore<T, 2>(bufA + j0, reinterpret_cast<const T(&)[2]>(v[u][0]));   // two 8-byte halves
                        vstore<T, 2>(bufA + j0 + 2, reinterpret_cast<const T(&)[2]>(v[u][2]));
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const int j = j0 + e;
                            if (j >= 0 && j < lenA) bufA[j] = v[u][e];
                        }
                    }
                }
            }
        }
    }
    lds_barrier_vm();
    multi1d_levels<T, F>(a, H, own0, own_len, bufA, bufB, y);
}

// (Tried in round 3 and dropped: a persistent variant that walks 2-8 consecutive tiles of a line and requests tile k+1 -- five
//  16-byte loads per thread, held in registers -- right after tile k has reached LDS.  Bit-identical, but 8192 x 2^16 db4 went
//  from 1000 us to 1137-1175 us, db2 859 -> 917-959, Float64 937 -> 1005-1025: the 20 extra VGPRs cost one to three resident
//  workgroups per CU, and six to eight independent workgroups already overlap each other's staging latency.)

// ==========================================================================================
// 2-D multi-level tile kernel for the small, cache-resident levels: NL (1 or 2) consecutive 2-D
// levels per launch.  A workgroup owns an OT x OT piece of the input block, stages it with a halo
// of H0 = (F-2)(2^NL - 1) on every side into LDS (periodic wrap resolved while staging) and runs
// dim-2 pass / dim-1 pass / next level entirely LDS -> LDS with fixed windows (same indexing as
// k_fwd1d_multi, per dimension).  Halo work is recomputed instead of exchanged, so workgroups are
// independent; the redundant reads hit L2 (these levels are <= 16 MiB).  Replaces the per-level
// launches whose cost was a ~7 us latency chain each, and shrinks what is left for k_tail_fwd.
template <typename T, int F>
struct Multi2DArgs {
    const T *src; int64_t lds;      // input block M x N
    T *y; int64_t ldy;
    T *ll; int64_t ldll;            // approximation after NL levels (dense buffer or y itself)
    int M, N;
    int NL, OT;
    int ld0;                        // LDS leading dimension of the staged tile (multiple of 4)
    TapsF<T, F> tp;
};

template <typename T, int F>
__device__ __forceinline__ void window_pair(const T (&xv)[2 * F - 2], const TapsF<T, F> &tp, T &sv, T &dv)
{
    sv = tp.h[0] * xv[F - 2];
#pragma unroll
    for (int m = 1; m < F; ++m) sv = sv + tp.h[m] * xv[F - 2 + m];
    dv = tp.g[F - 1] * xv[0];
#pragma unroll
    for (int m = F - 2; m >= 0; --m) dv = dv + tp.g[m] * xv[F - 1 - m];
}

template <typename T, int F>
__global__ void __launch_bounds__(512) k_fwd2d_multi(Multi2DArgs<T, F> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int ti = tid & 63, tj = tid >> 6;        // 64 threads along dim 1, 8 along dim 2
    const int NL = a.NL;
    int H[4];
    H[NL] = 0;
    for (int t = NL; t >= 1; --t) H[t - 1] = 2 * H[t] + (F - 2);
    const int r0 = blockIdx.x * a.OT, c0 = blockIdx.y * a.OT;     // owned origin (input coordinates)
    const int S0 = a.OT + 2 * H[0];
    const int ld = a.ld0;
    T *A = reinterpret_cast<T *>(smem_raw);                        // S x S (current level input), ld
    T *Rs = A + (size_t)ld * S0;                                   // S x Lout, ld
    T *Rd = Rs + (size_t)ld * ((S0 - (F - 2)) / 2 + 1);

    // stage A[i + j*ld] = X[(r0 - H0 + i) mod M, (c0 - H0 + j) mod N]; 8-byte loads along dim 1
    // (r0 - H0 is even and M is even, so a row pair never straddles the periodic wrap)
    {
        typedef typename VecOf<T, 2>::type T2;
        const int S0h = S0 >> 1;                   // S0 is even
        int gr = r0 - H[0] + 2 * ti;
        for (int ih = ti; ih < S0h; ih += 64, gr += 128) {
            int g = gr;
            if (g < 0) g += a.M;
            if (g >= a.M) g -= a.M;
            for (int j = tj; j < S0; j += 4 * 8) {
                T2 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u * 8;
                    if (jj < S0) {
                        int gc = c0 - H[0] + jj;
                        if (gc < 0) gc += a.N;
                        if (gc >= a.N) gc -= a.N;
                        v[u] = *reinterpret_cast<const T2 *>(a.src + g + (int64_t)gc * a.lds);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int jj = j + u * 8;
                    if (jj < S0) { A[2 * ih + jj * ld] = v[u].x; A[2 * ih + 1 + jj * ld] = v[u].y; }
                }
            }
        }
    }
    lds_barrier_vm();
    int S = S0;
    for (int t = 1; t <= NL; ++t) {
        const int own = a.OT >> t;
        const int Lout = own + 2 * H[t];
        // dim-2 pass: rows i in [0,S) (threads along i), column pairs c in [0,Lout): window A[i, 2c .. 2c+2F-3]
        for (int i = ti; i < S; i += 64) {
            for (int c = tj; c < Lout; c += 8) {
                T xv[2 * F - 2];
#pragma unroll
                for (int q = 0; q < 2 * F - 2; ++q) xv[q] = A[i + (2 * c + q) * ld];
                T sv, dv;
                window_pair<T, F>(xv, a.tp, sv, dv);
                Rs[i + c * ld] = sv;
                Rd[i + c * ld] = dv;
            }
        }
        lds_barrier();
        // dim-1 pass: columns c of Rs and Rd (2*Lout of them), row pairs r in [0,Lout) (threads along r)
        const int Mt = a.M >> t, Nt = a.N >> t;                    // quadrant extents of this level
        const int gr0 = (r0 >> t) - H[t], gc0 = (c0 >> t) - H[t];
        const bool lastlev = (t == NL);
        for (int r = ti; r < Lout; r += 64) {
            const int ro = r - H[t];
            const int64_t grow = gr0 + r;
            for (int cc = tj; cc < 2 * Lout; cc += 8) {
                const int which = cc >= Lout;                       // 0: from Rs (s_j), 1: from Rd (d_j)
                const int c = which ? cc - Lout : cc;
                const T *Rp = (which ? Rd : Rs) + c * ld + 2 * r;
                T xv[2 * F - 2];
#pragma unroll
                for (int q = 0; q < 2 * F - 2; ++q) xv[q] = Rp[q];
                T sv, dv;
                window_pair<T, F>(xv, a.tp, sv, dv);
                const int co = c - H[t];
                const bool owned = ro >= 0 && ro < own && co >= 0 && co < own;
                const int64_t gcol = gc0 + c;                      // global coordinates inside the quadrant
                if (which == 0) {
                    if (!lastlev) A[r + c * ld] = sv;              // LL of this level -> next level input
                    if (owned) {
                        if (lastlev) a.ll[grow + gcol * a.ldll] = sv;
                        a.y[(Mt + grow) + gcol * a.ldy] = dv;      // d_i(s_j)
                    }
                } else if (owned) {
                    a.y[grow + (Nt + gcol) * a.ldy] = sv;          // s_i(d_j)
                    a.y[(Mt + grow) + (Nt + gcol) * a.ldy] = dv;  // d_i(d_j)
                }
            }
        }
        lds_barrier();
        S = Lout;
    }
}

// ==========================================================================================
// tail: all remaining levels of a small block inside one workgroup, block resident in LDS
template <typename T>
struct TailArgs {
    const T *src; int64_t s1;       // source block: element (i, j) at src[i + j*s1]
    T *y; int64_t ldy;              // destination array (same origin as the block)
    int64_t src_item, y_item;       // per-blockIdx.x offsets (batched lines)
    int m0, m1;                     // block extents (m1 == 1 for 1-D lines)
    int nt;                         // 1: transform dim 1 only; 2: both dims
    int nlev;                       // levels to run (>= 1)
    int cap;                        // elements per LDS buffer
};

__device__ __forceinline__ int wrap_idx(int i, int n)
{
    // periodic index without integer division: one trip for n >= F, a few for tiny lines
    while (i >= n) i -= n;
    while (i < 0) i += n;
    return i;
}
// (s, d) pair k of a line of length n stored at p with element stride `stride`.
// FC > 0: compile-time filter length (all operands fetched before the arithmetic, so the LDS
// reads are issued back to back); FC == 0: run-time length.
template <typename T, int FC>
__device__ __forceinline__ void tail_pair(const T *p, int stride, int k, int n, const Taps<T> &tp, T &s, T &d)
{
    if constexpr (FC > 0) {
        // window x[2k-(FC-2) .. 2k+FC-1] covers both branches
        T xv[2 * FC - 2];
        const int lo = 2 * k - (FC - 2);
        if (lo >= 0 && lo + 2 * FC - 2 <= n) {
            // interior pair: no periodic wrap, plain strided reads (no per-element index math)
            const T *q = p + lo * stride;
#pragma unroll
            for (int e = 0; e < 2 * FC - 2; ++e) xv[e] = q[e * stride];
        } else {
            int i = wrap_idx(lo, n);
#pragma unroll
            for (int e = 0; e < 2 * FC - 2; ++e) {
                xv[e] = p[i * stride];
                if (++i >= n) i -= n;
            }
        }
        s = tp.h[0] * xv[FC - 2];
#pragma unroll
        for (int m = 1; m < FC; ++m) s = s + tp.h[m] * xv[FC - 2 + m];
        d = tp.g[FC - 1] * xv[0];
#pragma unroll
        for (int m = FC - 2; m >= 0; --m) d = d + tp.g[m] * xv[FC - 1 - m];
    } else {
        // run-time length: taps in blocks of 8 with the block's LDS reads issued before its arithmetic (one dependent
        // read per tap made a 59-tap line cost ~100 cycles per tap)
        const int F = tp.F;
        int i = wrap_idx(2 * k, n);
        s = tp.h[0] * p[i * stride];
        for (int m0 = 1; m0 < F; m0 += 8) {
            T xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 + e < F) {
                    if (++i >= n) i -= n;
                    xv[e] = p[i * stride];
                }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 + e < F) s = s + tp.h[m0 + e] * xv[e];
        }
        i = wrap_idx(2 * k + 1 - (F - 1), n);
        d = tp.g[F - 1] * p[i * stride];
        for (int m0 = F - 2; m0 >= 0; m0 -= 8) {
            T xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 - e >= 0) {
                    if (++i >= n) i -= n;
                    xv[e] = p[i * stride];
                }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 - e >= 0) d = d + tp.g[m0 - e] * xv[e];
        }
    }
}
// idx -> (idx % m, idx / m) with a shift when m is a power of two
__device__ __forceinline__ void split_idx(int idx, int m, int lg, int &lo, int &hi)
{
    if (lg >= 0) { lo = idx & (m - 1); hi = idx >> lg; }
    else { hi = idx / m; lo = idx - hi * m; }
}
__device__ __forceinline__ int ilog2_or_neg(int m) { return (m & (m - 1)) == 0 ? (31 - __clz(m)) : -1; }

// LDS layout: column j of the current block starts at j * ld with ld = m0 + 1 (odd pad) whenever
// the block has more than one column, so that both the dim-2 pass (threads along i) and the dim-1
// pass (threads along k, stride-2 reads) are free of systematic bank conflicts.
template <typename T, int FC>
__global__ void __launch_bounds__(1024) k_tail_fwd(TailArgs<T> a, Taps<T> tp)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *A = reinterpret_cast<T *>(smem_raw);
    T *B = A + a.cap;
    const int tid = threadIdx.x;
    int nthr = blockDim.x;
    bool multi = nthr > 64;      // several waves -> workgroup barriers; wave 0 alone -> LDS wait only
    const T *src = a.src + (int64_t)blockIdx.x * a.src_item;
    T *y = a.y + (int64_t)blockIdx.x * a.y_item;
    int m0 = a.m0, m1 = a.m1;
    const int ld = (m1 > 1) ? (a.m0 | 1) : a.m0;      // fixed for all levels
    {
        const int lg = ilog2_or_neg(m0);
        const int total = m0 * m1;
        constexpr int VW = 16 / sizeof(T);
        const bool vec_ok = (m0 % VW) == 0 && (a.s1 % VW) == 0 && (a.src_item % VW) == 0 &&
                            ((reinterpret_cast<uintptr_t>(a.src) & 15) == 0);
        if (vec_ok) {
            // 16-byte loads, four per thread in flight (covers a 16 Ki-element block with 1024 threads)
            const int totalv = total / VW, m0v = m0 / VW;
            const int lgv = ilog2_or_neg(m0v);
            for (int idx = tid; idx < totalv; idx += 4 * nthr) {
                T v[4][VW];
                int ii[4], jj[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int id = idx + r * nthr;
                    if (id < totalv) {
                        split_idx(id, m0v, lgv, ii[r], jj[r]);
                        vload<T, VW>(src + ii[r] * VW + (int64_t)jj[r] * a.s1, v[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (idx + r * nthr < totalv) {
#pragma unroll
                        for (int e = 0; e < VW; ++e) A[ii[r] * VW + e + jj[r] * ld] = v[r][e];
                    }
            }
        } else {
            for (int idx = tid; idx < total; idx += 4 * nthr) {
                T v[4];
                int ii[4], jj[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int id = idx + r * nthr;
                    if (id < total) {
                        split_idx(id, m0, lg, ii[r], jj[r]);
                        v[r] = src[ii[r] + (int64_t)jj[r] * a.s1];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (idx + r * nthr < total) A[ii[r] + jj[r] * ld] = v[r];
            }
        }
    }
    lds_barrier();
    for (int lev = 0; lev < a.nlev; ++lev) {
        const bool last = (lev == a.nlev - 1);
        const int h0 = m0 >> 1;
        // small levels: hand the rest of the transform to wave 0 (no barrier round trips)
        if (multi && m0 * m1 <= 256) {
            if (tid >= 64) return;
            multi = false; nthr = 64;
        }
        const int lg0 = ilog2_or_neg(m0), lgh = ilog2_or_neg(h0);
        if (a.nt == 1) {
            // single line of length m0 in A
            for (int k = tid; k < h0; k += nthr) {
                T s, d;
                tail_pair<T, FC>(A, 1, k, m0, tp, s, d);
                y[h0 + k] = d;
                if (last) y[k] = s;
                else B[k] = s;
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            T *t = A; A = B; B = t;
            m0 = h0;
        } else {
            const int h1 = m1 >> 1;
            // dim-2 pass: A (m0 x m1) -> B (m0 x m1), [s ; d] along j
            for (int idx = tid; idx < m0 * h1; idx += 2 * nthr) {
                int i0, k0, i1 = 0, k1 = 0;
                const bool two = (idx + nthr) < m0 * h1;
                split_idx(idx, m0, lg0, i0, k0);
                if (two) split_idx(idx + nthr, m0, lg0, i1, k1);
                T s0, d0, s1 = (T)0, d1 = (T)0;
                tail_pair<T, FC>(A + i0, ld, k0, m1, tp, s0, d0);
                if (two) tail_pair<T, FC>(A + i1, ld, k1, m1, tp, s1, d1);
                B[i0 + k0 * ld] = s0;
                B[i0 + (h1 + k0) * ld] = d0;
                if (two) { B[i1 + k1 * ld] = s1; B[i1 + (h1 + k1) * ld] = d1; }
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // dim-1 pass: B -> details to y, LL to A (h0 x h1) or y
            for (int idx = tid; idx < h0 * m1; idx += 2 * nthr) {
                int k0, j0, k1 = 0, j1 = 0;
                const bool two = (idx + nthr) < h0 * m1;
                split_idx(idx, h0, lgh, k0, j0);
                if (two) split_idx(idx + nthr, h0, lgh, k1, j1);
                T s0, d0, s1 = (T)0, d1 = (T)0;
                tail_pair<T, FC>(B + j0 * ld, 1, k0, m0, tp, s0, d0);
                if (two) tail_pair<T, FC>(B + j1 * ld, 1, k1, m0, tp, s1, d1);
                y[(int64_t)j0 * a.ldy + h0 + k0] = d0;
                if (j0 < h1 && !last) A[k0 + j0 * ld] = s0;
                else y[(int64_t)j0 * a.ldy + k0] = s0;
                if (two) {
                    y[(int64_t)j1 * a.ldy + h0 + k1] = d1;
                    if (j1 < h1 && !last) A[k1 + j1 * ld] = s1;
                    else y[(int64_t)j1 * a.ldy + k1] = s1;
                }
            }
            if (multi) lds_barrier(); else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            m0 = h0;
            m1 = h1;
        }
    }
}

// ==========================================================================================
// host side
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is sticky per (function, device): do it once
static hipError_t set_max_lds_once(const void *fn, size_t bytes, unsigned char (&done)[64], size_t (&cur)[64])
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (done[dev] && cur[dev] >= bytes) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes < 65536 ? 65536 : bytes));
    if (e == hipSuccess) { done[dev] = 1; cur[dev] = bytes < 65536 ? 65536 : bytes; }
    return e;
}
// tuning / test switches: per-context options (wl_ctx_set_option), see wl_internal.h
#define env_int(name, dflt) ((int)opt(name, dflt))
#define env_int_raw(name, dflt) ((int)opt(name, dflt))

template <typename T>
constexpr int tail_cap() { return sizeof(T) == 4 ? 16384 : 8192; }     // block elements
template <typename T>
constexpr int tail_lds_elems() { return tail_cap<T>() + 256; }          // + odd-pad columns

template <typename T>
static hipError_t launch_tail(hipStream_t st, const Taps<T> &taps, const T *src, int64_t s1, T *y, int64_t ldy,
                              int64_t src_item, int64_t y_item, int nitems, int m0, int m1, int nt, int nlev)
{
    TailArgs<T> a;
    a.src = src; a.s1 = s1; a.y = y; a.ldy = ldy; a.src_item = src_item; a.y_item = y_item;
    // right-sized LDS (two buffers of the padded block) so that several small blocks share a CU
    a.m0 = m0; a.m1 = m1; a.nt = nt; a.nlev = nlev;
    a.cap = (int)((((int64_t)(m1 > 1 ? (m0 | 1) : m0) * m1) + 15) & ~15);
    const size_t shmem = 2 * (size_t)a.cap * sizeof(T);
    int64_t work = (int64_t)m0 * m1 / 2;
    int threads = work >= 4096 ? 1024 : (work >= 512 ? 256 : 64);
#define WL_TAIL_LAUNCH(FC_)                                                                                   \
    do {                                                                                                      \
        static unsigned char done__[64]; static size_t cur__[64];                                             \
        hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_tail_fwd<T, FC_>), 160 * 1024, done__, cur__); \
        if (e != hipSuccess) return e;                                                                        \
        hipLaunchKernelGGL((k_tail_fwd<T, FC_>), dim3((unsigned)nitems), dim3(threads), shmem, st, a, taps);  \
    } while (0)
    switch (taps.F) {
    case 2: WL_TAIL_LAUNCH(2); break;
    case 4: WL_TAIL_LAUNCH(4); break;
    case 6: WL_TAIL_LAUNCH(6); break;
    case 8: WL_TAIL_LAUNCH(8); break;
    case 10: WL_TAIL_LAUNCH(10); break;
    default: WL_TAIL_LAUNCH(0); break;
    }
#undef WL_TAIL_LAUNCH
    return hipGetLastError();
}

template <typename T, int F, int RPL>
static hipError_t launch_fwd2d_r(hipStream_t st, const Taps<T> &taps, bool lvl1, const T *src, int64_t lds,
                                 T *y, int64_t ldy, T *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                                 int64_t nbatch = 1, int64_t bs_src = 0, int64_t bs_y = 0, int64_t bs_ll = 0, int nll = 1)
{
    constexpr int VR = 64 * RPL - 16;
    Fwd2DArgs<T, F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.bs_src = bs_src; a.bs_y = bs_y; a.bs_ll = bs_ll; a.nll = nll;
    a.nstrips = (int)((ms + VR - 1) / VR);
    a.nt = (!lvl1 && env_int("WL_REVERSE", 1)) ? 4 : 0;
    int TJ = env_int("WL_TJ", 128);
    // smaller levels: trade chunk length for parallelism (>= ~8 waves per CU while chunks stay >= 32
    // columns, >= 2 per CU down to 16 columns); every chunk length stays a multiple of 16
    auto nwaves = [&](int tj) { return (int64_t)a.nstrips * ((ns + tj - 1) / tj) * nbatch; };
    const int wpc = env_int("WL_WAVES_PER_CU", 8);
    while (TJ > 32 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * wpc) TJ >>= 1;
    while (TJ > 16 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * env_int("WL_WAVES_MIN", 8)) TJ >>= 1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.tp = shrink<T, F>(taps);
    const unsigned wpb = 1;      // waves per workgroup (4-wave workgroups measured no faster)
    const unsigned nwg = ((unsigned)(a.nstrips * a.nchunks) + wpb - 1) / wpb;
    if (lvl1) hipLaunchKernelGGL((k_fwd2d_stream<T, F, RPL, 1>), dim3(nwg, (unsigned)nbatch), dim3(64 * wpb), 0, st, a);
    else hipLaunchKernelGGL((k_fwd2d_stream<T, F, RPL, 0>), dim3(nwg, (unsigned)nbatch), dim3(64 * wpb), 0, st, a);
    return hipGetLastError();
}

template <typename T, int F>
static hipError_t launch_fwd2d(hipStream_t st, const Taps<T> &taps, bool lvl1, const T *src, int64_t lds,
                               T *y, int64_t ldy, T *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    constexpr int RPL = 16 / sizeof(T);
    return launch_fwd2d_r<T, F, RPL>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
}


template <typename T, int F>
static hipError_t launch_fwd2d_pair(hipStream_t st, const Taps<T> &taps, bool lvl1, const T *src, int64_t lds,
                                    T *y, int64_t ldy, T *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    Fwd2D2Args<T, F> a;
    constexpr int VR = 52 * 4;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.nstrips = (int)((ms + VR - 1) / VR);
    int TJ = env_int("WL_TJ2", 128);
    auto nwaves = [&](int tj) { return (int64_t)a.nstrips * ((ns + tj - 1) / tj); };
    const int wpc = env_int("WL_WAVES_PER_CU", 8);
    while (TJ > 32 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * wpc) TJ >>= 1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.rev = (!lvl1 && env_int("WL_REVERSE", 1)) ? 1 : 0;
    a.tp = shrink<T, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    if (lvl1) hipLaunchKernelGGL((k_fwd2d_stream2<T, F, 1>), dim3(nwg), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_fwd2d_stream2<T, F, 0>), dim3(nwg), dim3(64), 0, st, a);
    return hipGetLastError();
}

template <typename T, int F>
static hipError_t launch_fwd1d(hipStream_t st, const Taps<T> &taps, bool lvl1, const T *src, int64_t src_ls,
                               T *sdst, int64_t s_ls, T *ddst, int64_t d_ls, int64_t n, int64_t nlines, int cu_count)
{
    Fwd1DArgs<T, F> a;
    a.src = src; a.src_ls = src_ls; a.sdst = sdst; a.s_ls = s_ls; a.ddst = ddst; a.d_ls = d_ls; a.n = n;
    a.ntiles = ((n >> 1) + 247) / 248;
    a.tp = shrink<T, F>(taps);
    int64_t gx = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 8;
    if (gx > cap) gx = cap;
    // gridDim.y is limited to 65535: launch in slabs of lines
    const int64_t slab = (env_int_raw("WL_SLAB_LINES", 32768) > 0) ? env_int_raw("WL_SLAB_LINES", 32768) : 32768;
    for (int64_t l0 = 0; l0 < nlines; l0 += slab) {
        const int64_t nl = (nlines - l0 < slab) ? (nlines - l0) : slab;
        Fwd1DArgs<T, F> b = a;
        b.src = a.src + l0 * a.src_ls; b.sdst = a.sdst + l0 * a.s_ls; b.ddst = a.ddst + l0 * a.d_ls;
        if (lvl1) hipLaunchKernelGGL((k_fwd1d_stream<T, F, 1>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
        else hipLaunchKernelGGL((k_fwd1d_stream<T, F, 0>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
    }
    return hipGetLastError();
}


template <typename T, int F>
static hipError_t launch_fwd1d_multi(hipStream_t st, const Taps<T> &taps, bool lvl1, const T *src, int64_t src_ls,
                                     T *y, int64_t y_ls, T *sdst, int64_t s_ls, int64_t n, int64_t nlines, int NL)
{
    Multi1DArgs<T, F> a;
    a.src = src; a.src_ls = src_ls; a.y = y; a.y_ls = y_ls; a.sdst = sdst; a.s_ls = s_ls; a.n = n; a.NL = NL;
    // tile = 32 KiB of input for long lines; shorter tiles (down to 4 KiB) when there would otherwise be
    // fewer workgroups than CUs -- a workgroup's latency chain (stage, NL levels, barriers) is ~constant
    a.TS = env_int("WL_TS", 0);
    if (a.TS < 256 || (a.TS % 64) != 0) {
        a.TS = (int)(16384 / sizeof(T));
        // Long lines: shave the tile so that level 1 -- TS / 2 + 2 H1 pairs, 256 threads x (16 / sizeof(T)) pairs per round -- takes
        // TWO rounds of the workgroup instead of two and a sliver (TS = 4096, db4, 4 levels: 521 thread groups for 512 slots per two
        // rounds; level 2 likewise 265 for 256): 7 rounds per tile become 5 for 5 % fewer samples.  Same bits (tiles are a
        // partition, the last one is partial).  r04: C5 first pass 910 -> 869 us, C2 49.0 -> 47.1 us.  Lines of a few tiles keep
        // the power of two (a partial last tile would be most of the work).
        const int H1 = (F - 2) * ((1 << (NL - 1)) - 1);
        const int fit = ((a.TS - 4 * H1) / 64) * 64;
        if (env_int("WL_TS_FIT", 1) && n >= 8 * (int64_t)a.TS && fit >= a.TS / 2) a.TS = fit;
    }
    while (a.TS > (int)(4096 / sizeof(T)) && ((n + a.TS - 1) / a.TS) * nlines < 512) a.TS >>= 1;
    a.tp = shrink<T, F>(taps);
    const int H0 = (F - 2) * ((1 << NL) - 1), H1 = (F - 2) * ((1 << (NL - 1)) - 1);
    const size_t elems = (size_t)((a.TS + 2 * H0 + 7) & ~7) + 16 + (size_t)(a.TS / 2 + 2 * H1 + 8) + 32;   // (+ room for the last groups' windows)
    const size_t shmem = elems * sizeof(T);
    const unsigned ntiles = (unsigned)((n + a.TS - 1) / a.TS);
    const int64_t slab = (env_int_raw("WL_SLAB_LINES", 32768) > 0) ? env_int_raw("WL_SLAB_LINES", 32768) : 32768;
    for (int64_t l0 = 0; l0 < nlines; l0 += slab) {
        const int64_t nl = (nlines - l0 < slab) ? (nlines - l0) : slab;
        Multi1DArgs<T, F> b = a;
        b.src = a.src + l0 * a.src_ls; b.y = a.y + l0 * a.y_ls; b.sdst = a.sdst + l0 * a.s_ls;
        if (lvl1) hipLaunchKernelGGL((k_fwd1d_multi<T, F, 1>), dim3(ntiles, (unsigned)nl), dim3(256), shmem, st, b);
        else hipLaunchKernelGGL((k_fwd1d_multi<T, F, 0>), dim3(ntiles, (unsigned)nl), dim3(256), shmem, st, b);
    }
    return hipGetLastError();
}


template <typename T, int F>
static hipError_t launch_fwd2d_multi(hipStream_t st, const Taps<T> &taps, const T *src, int64_t lds, T *y, int64_t ldy,
                                     T *ll, int64_t ldll, int M, int N, int NL)
{
    Multi2DArgs<T, F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.M = M; a.N = N; a.NL = NL;
    // owned tile: 64 x 64 inputs, smaller while there would be fewer than ~64 workgroups
    int OT = (NL >= 3) ? 32 : 64;          // (OT + 2*H0)^2 tile + two row-pass buffers must fit 160 KiB of LDS
    const int mn = M < N ? M : N;
    while (OT > 16 && ((int64_t)(M / OT) * (N / OT) < 64 || OT > mn)) OT >>= 1;
    a.OT = OT;
    const int H0 = (F - 2) * ((1 << NL) - 1);
    const int S0 = OT + 2 * H0;
    a.ld0 = (S0 + 3) & ~3;
    a.tp = shrink<T, F>(taps);
    const int L1 = (S0 - (F - 2)) / 2 + 1;
    const size_t elems = (size_t)a.ld0 * S0 + 2 * (size_t)a.ld0 * L1 + 16;
    const size_t shmem = elems * sizeof(T);
    static unsigned char done__[64]; static size_t cur__[64];
    hipError_t e = set_max_lds_once(reinterpret_cast<const void *>(&k_fwd2d_multi<T, F>), 160 * 1024, done__, cur__);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_fwd2d_multi<T, F>), dim3((unsigned)(M / OT), (unsigned)(N / OT)), dim3(512), shmem, st, a);
    return hipGetLastError();
}

#define WL_DISPATCH_F(F_, ...)                               \
    switch (F_) {                                            \
    case 2: { constexpr int FF = 2; __VA_ARGS__; } break;    \
    case 4: { constexpr int FF = 4; __VA_ARGS__; } break;    \
    case 6: { constexpr int FF = 6; __VA_ARGS__; } break;    \
    case 8: { constexpr int FF = 8; __VA_ARGS__; } break;    \
    case 10: { constexpr int FF = 10; __VA_ARGS__; } break;  \
    default: break;                                          \
    }

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

thread_local SrcView tl_srcview = {0, 0, 0, -1, 0.0, 0.0, nullptr, 0, 0};

template <typename T>
int filter_fwd_levels(void *ws, bool ws_gen, int cu_count, int path, hipStream_t st, const BoxSpec &b,
                      T *y, const T *x, const Taps<T> &taps, int L, const char **kernel_name, int *hip_err)
{
#define WL_TRY(expr)                                                  \
    do {                                                              \
        hipError_t e__ = (expr);                                      \
        if (e__ != hipSuccess) { if (hip_err) *hip_err = (int)e__; return WL_EHIP; } \
    } while (0)
    const int64_t N = b.dims[0] * b.dims[1] * b.dims[2];
    Work<T> w = carve<T>(ws, N, b.nt, ws_gen);
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    const bool fastF = (path == 0) && (F % 2 == 0) && (F <= 10);
    const bool two_d = (b.nd == 2 && b.nt == 2);
    const bool lines = (b.nt == 1 && b.nd <= 2);          // 1-D vector or batched columns
    const int64_t nlines = lines ? b.dims[1] : 1;
    const char *dominant = nullptr;

    const T *cur = x;
    Strides3 cur_st = b.full;
    int pp = 0;
    int lstep = 1;
    for (int l = 1; l <= L; l += lstep) {
        lstep = 1;
        int64_t n[3];
        level_box(b, l, n);
        Extent3 ext = {{n[0], n[1], n[2]}};
        Extent3 lo = low_corner(b, n);
        const bool last = (l == L);
        T *llbuf = pp ? w.B : w.A;
        int64_t hn[3] = {lo.n[0], lo.n[1], lo.n[2]};
        Strides3 ll_st = dense_strides(hn);
        Strides3 box_st = dense_strides(n);

        // ---- batch of independent 2-D blocks (box n0 x n1 x B, first two axes transformed: the shifted copies of a
        //      translation-invariant denoise): one launch of the LDS-exchange kernel per level, planes over blockIdx.y ----
        // ... two levels per launch while the planes are big enough for the fused pair kernel to pay (many planes keep the chip full
        //     where a single 2048^2 block would not)
        if constexpr (sizeof(T) == 4) {
            if (fastF && b.nd == 3 && b.nt == 2 && (L - l + 1) >= 2 && env_int("WL_PAIR_BATCH", 1) != 0 && n[2] >= 2 && n[2] <= 65535 &&
                n[0] * n[1] >= (int64_t)env_int_raw("WL_PAIR_BATCH_MIN", 1 << 22) && fwd2d_pair_ok(F, n[0], n[1]) && cur_st.s[0] == 1 &&
                (cur_st.s[1] % VEC) == 0 && (cur_st.s[2] % VEC) == 0 && aligned16(cur) && (b.full.s[1] % VEC) == 0 &&
                (b.full.s[2] % VEC) == 0 && aligned16(y) && aligned16(llbuf)) {
                const bool view = (l == 1 && tl_srcview.mod > 0);
                const bool thr = tl_srcview.th >= 0 && (view || (l > 1 && tl_srcview.used != 0 && tl_srcview.corner0 == n[0] && tl_srcview.corner1 == n[1]));
                const bool lastp = (l + 1 == L);
                int64_t hn2[3] = {n[0] >> 2, n[1] >> 2, n[2]};
                Strides3 ll2_st = dense_strides(hn2);
                WL_TRY(fwd2d_pair_launch(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], lastp ? (T *)nullptr : llbuf, hn2[0], n[0], n[1],
                                         cu_count, n[2], cur_st.s[2], b.full.s[2], ll2_st.s[2], view ? tl_srcview.mod : 0,
                                         view ? tl_srcview.spin0 : 0, thr ? &tl_srcview : nullptr));
                if (view) tl_srcview.used = 1;
                if (thr) { tl_srcview.corner0 = lastp ? 0 : hn2[0]; tl_srcview.corner1 = lastp ? 0 : hn2[1]; }
                if (!dominant) dominant = "k_fwd2d_pair";
                lstep = 2;
                cur = llbuf; cur_st = ll2_st; pp ^= 1;
                continue;
            }
        }
        if constexpr (sizeof(T) == 4) {
            if (fastF && b.nd == 3 && b.nt == 2 && fwd2d_lds_ok(F, 1, n[0], n[1]) && n[2] <= 65535 && cur_st.s[0] == 1 &&
                (cur_st.s[1] % VEC) == 0 && (cur_st.s[2] % VEC) == 0 && aligned16(cur) && (b.full.s[1] % VEC) == 0 &&
                (b.full.s[2] % VEC) == 0 && aligned16(y) && aligned16(llbuf)) {
                const bool view = (l == 1 && tl_srcview.mod > 0);      // level 1 of a translation-invariant batch: virtual shifted planes
                // ... whose launches also threshold the coefficients they store (every level that runs here, once level 1 did)
                const bool thr = tl_srcview.th >= 0 && (view || (l > 1 && tl_srcview.used != 0 && tl_srcview.corner0 == n[0] && tl_srcview.corner1 == n[1]));
                WL_TRY(fwd2d_lds_launch(st, taps, 1, l == 1, cur, cur_st.s[1], y, b.full.s[1], last ? (T *)nullptr : llbuf, hn[0],
                                        n[0], n[1], cu_count, n[2], cur_st.s[2], b.full.s[2], ll_st.s[2], (int)n[2],
                                        view ? tl_srcview.mod : 0, view ? tl_srcview.spin0 : 0, thr ? &tl_srcview : nullptr));
                if (thr) { tl_srcview.corner0 = last ? 0 : hn[0]; tl_srcview.corner1 = last ? 0 : hn[1]; }
                if (view) tl_srcview.used = 1;
                if (!dominant) dominant = "k_fwd2d_lds";
                cur = llbuf; cur_st = ll_st; pp ^= 1;
                continue;
            }
        }
        // a virtually shifted source is only valid for the two batch tiers above: anything else would read planes it does not hold
        if (l == 1 && tl_srcview.mod > 0) return WL_RETRY_NOVIEW;
        // ---- two levels of a block too big for one resident round of the staging tile kernel (2048 rows): tiles without staging ----
        if constexpr (sizeof(T) == 4) {
            if (fastF && two_d && env_int("WL_TILEB", 1) && (L - l + 1) >= 2 && n[0] <= env_int("WL_TILEB_MAX", 2048) &&
                n[1] <= env_int("WL_TILEB_MAX", 2048) && n[0] * n[1] >= (int64_t)env_int_raw("WL_TILEB_MIN", 1 << 21) && cur_st.s[0] == 1 &&
                (cur_st.s[1] % 4) == 0 && aligned16(cur) && (b.full.s[1] % 4) == 0 && aligned16(y) && aligned16(llbuf) &&
                fwd2d_tile_ok(F, 2, n[0], n[1])) {
                const bool lastt = (l + 1 == L);
                T *lld = lastt ? y : llbuf;
                const int64_t ldd = lastt ? b.full.s[1] : (n[0] >> 2);
                WL_TRY(fwd2d_tileB_launch(st, taps, cur, cur_st.s[1], y, b.full.s[1], lld, ldd, (int)n[0], (int)n[1]));
                if (!dominant) dominant = "k_fwd2d_tileB";
                lstep = 2;
                int64_t hn2[3] = {n[0] >> 2, n[1] >> 2, n[2]};
                cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                continue;
            }
        }
        // ---- tile kernel: 1..3 fused levels of a cache-resident block (wl_tile.hip; Float64: up to 2 levels) ----
        {
            auto al4 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) % (4 * sizeof(T))) == 0; };     // 4-element vectors
            if (fastF && two_d && env_int("WL_TILE", 1) && n[0] <= env_int("WL_TILE_MAX", 1024) && n[1] <= env_int("WL_TILE_MAX", 1024) &&
                cur_st.s[0] == 1 && (cur_st.s[1] % 4) == 0 && al4(cur) && (b.full.s[1] % 4) == 0 && al4(y) && al4(llbuf)) {
                int NL = L - l + 1;
                if (NL > 3) NL = 3;
                const int64_t nmax = n[0] > n[1] ? n[0] : n[1];
                // three levels per tile launch from 512^2 down (round 5, with the approximation handed over through L2: 2048^2 full depth 34.7 -> 32.2 us,
                // 512^2 20.1 -> 17.7, 8192^2 -2 %; round 4 measured it 4 % slower and kept two)
                if (NL == 3 && (nmax > env_int("WL_TILE_NL3_MAX", 512) || sizeof(T) == 8)) NL = 2;
                while (NL > 1 && !fwd2d_tile_ok(F, NL, n[0], n[1])) --NL;
                if (fwd2d_tile_ok(F, NL, n[0], n[1])) {
                    const bool lastt = (l + NL - 1 == L);
                    T *lld = lastt ? y : llbuf;
                    const int64_t ldd = lastt ? b.full.s[1] : (n[0] >> NL);
                    WL_TRY(fwd2d_tile_launch<T>(st, taps, NL, cur, cur_st.s[1], y, b.full.s[1], lld, ldd, (int)n[0], (int)n[1]));
                    if (!dominant) dominant = "k_fwd2d_tile";
                    lstep = NL;
                    int64_t hn2[3] = {n[0] >> NL, n[1] >> NL, n[2]};
                    cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                    continue;
                }
            }
        }
        // ---- 12..20 taps, Float32, blocks of 128 .. 1024 rows: two levels per launch through the LDS tile kernel (round 5).  These
        //      levels ran one streaming launch each (12-13 us apiece, short of workgroups) and the 128^2 level as two line passes:
        //      8192^2 db6 246 -> 224 us, sym8 286 -> 261, db10 329 -> 303; 1024^2 sym8 full depth 60.5 -> 36.8, 512^2 48.5 -> 29.8.
        //      Not above 1024 rows (2048^2: four rounds of one 512-thread workgroup per CU: sym8 8192^2 280 us, 4096^2 too: 356).
        if constexpr (sizeof(T) == 4) {
            auto al4 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; };
            if (path == 0 && F >= 12 && F <= 20 && (F % 2) == 0 && two_d && env_int("WL_TILE_LONG", 1) && n[0] <= env_int("WL_TILE_LONG_MAX", 1024) &&
                n[1] <= env_int("WL_TILE_LONG_MAX", 1024) && cur_st.s[0] == 1 && (cur_st.s[1] % 4) == 0 && al4(cur) && (b.full.s[1] % 4) == 0 && al4(y) &&
                al4(llbuf) && b.full.s[0] == 1) {
                int NL = (L - l + 1) >= 2 ? 2 : 1;
                while (NL > 1 && !fwd2d_tile_ok(F, NL, n[0], n[1])) --NL;
                // (a single remaining level of >= 256 rows stays with the streaming kernel below)
                if (fwd2d_tile_ok(F, NL, n[0], n[1]) && (NL == 2 || n[0] < 256)) {
                    const bool lastt = (l + NL - 1 == L);
                    T *lld = lastt ? y : llbuf;
                    const int64_t ldd = lastt ? b.full.s[1] : (n[0] >> NL);
                    WL_TRY(fwd2d_tile_launch<T>(st, taps, NL, cur, cur_st.s[1], y, b.full.s[1], lld, ldd, (int)n[0], (int)n[1]));
                    if (!dominant) dominant = "k_fwd2d_tile";
                    lstep = NL;
                    int64_t hn2[3] = {n[0] >> NL, n[1] >> NL, n[2]};
                    cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                    continue;
                }
            }
        }
        // ---- 2-D multi-level tiles for the cache-resident levels (two levels per launch) ----
        if (fastF && F <= 8 && two_d && env_int("WL_NO_MULTI2D", 0) == 0 && n[0] <= env_int("WL_M2D_MAX", 1024) &&
            n[1] <= env_int("WL_M2D_MAX", 1024) && n[0] >= env_int("WL_M2D_MIN", 128) && n[1] >= env_int("WL_M2D_MIN", 128) && (n[0] % 64) == 0 && (n[1] % 64) == 0 &&
            cur_st.s[0] == 1) {
            int NL = (L - l + 1);
            const int nl2max = env_int("WL_M2D_NL", 2);
            if (NL > nl2max) NL = nl2max;
            while (NL > 1 && ((n[0] >> NL) < 1 || (n[1] >> NL) < 1)) --NL;
            const bool lastm = (l + NL - 1 == L);
            T *lld = lastm ? y : llbuf;
            const int64_t ldd = lastm ? b.full.s[1] : (n[0] >> NL);
            bool ok = false;
            WL_DISPATCH_F(F, WL_TRY((launch_fwd2d_multi<T, FF>(st, taps, cur, cur_st.s[1], y, b.full.s[1], lld, ldd,
                                                               (int)n[0], (int)n[1], NL)));
                          ok = true);
            if (ok) {
                if (!dominant) dominant = "k_fwd2d_multi";
                lstep = NL;
                int64_t hn2[3] = {n[0] >> NL, n[1] >> NL, n[2]};
                cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                continue;
            }
        }
        // ---- 3-D boxes of <= 4096 elements: every remaining level in one workgroup (wl_tail.hip) ----
        if (path == 0 && b.nd == 3 && b.nt == 3 && env_int("WL_TAIL3", 1) && b.full.s[0] == 1 && cur_st.s[0] == 1 &&
            tail3_ok<T>(F, n[0], n[1], n[2], L - l + 1)) {
            WL_TRY(launch_tail3<T>(st, taps, 1, cur, cur_st.s[1], cur_st.s[2], y, b.full.s[1], b.full.s[2], (int)n[0], (int)n[1], (int)n[2],
                                   L - l + 1));
            if (!dominant) dominant = "k_tail3";
            break;
        }
        // ---- LDS-resident tail: finishes every remaining level in one launch ----
        // (batches of many lines: one workgroup per line is only efficient for short lines -- longer ones take
        //  another pass of the multi-level tile kernel first)
        // single lines: the latency-optimised tail (wl_tail.hip) takes 4096 samples; longer lines get another multi-level pass first
        const int64_t line_cap = (lines && nlines >= 32 && fastF) ? env_int("WL_TAIL_LINES_CAP", 512)
                                 : ((lines && fastF && env_int("WL_TAIL2", 1)) ? (int64_t)4096 : (int64_t)tail_cap<T>());
        if (path == 0 && (two_d || lines) && b.full.s[0] == 1) {
            const int64_t blk = two_d ? n[0] * n[1] : n[0];
            // filters beyond 24 taps: one workgroup is slow at 2 F multiply-adds per sample -- the chip-wide line / axis
            // kernels of wl_vlong.hip take every level down to 16 samples per dimension first
            // ... except 12..20 taps on power-of-two blocks of <= 4096 elements: the LDS-resident tail is instantiated for them
            const bool long_tail = F >= 12 && env_int("WL_LONG_TAIL", 1) && tail2_ok<T>(F, two_d ? 2 : 1, n[0], two_d ? n[1] : 1, L - l + 1, 20);
            const int64_t vl_cap = long_tail ? (int64_t)4096
                                   : ((vlong_filter_ok(F) && env_int("WL_NO_LONGF", 0) == 0) ? (two_d ? 64 : 16) : ((int64_t)1 << 40));
            if (blk <= (two_d ? (int64_t)tail_cap<T>() : line_cap) && blk <= vl_cap && n[0] < (1 << 20) && (!two_d || n[1] <= 256)) {
                // power-of-two blocks / lines of <= 16 KiB: the latency-optimised tail (wl_tail.hip)
                const bool t2 = env_int("WL_TAIL2", 1) && tail2_ok<T>(F, two_d ? 2 : 1, n[0], two_d ? n[1] : 1, L - l + 1, 20);
                if (two_d) {
                    if (t2) WL_TRY(launch_tail2<T>(st, taps, cur, cur_st.s[1], y, b.full.s[1], 0, 0, 1, (int)n[0], (int)n[1], 2, L - l + 1));
                    else WL_TRY(launch_tail<T>(st, taps, cur, cur_st.s[1], y, b.full.s[1], 0, 0, 1, (int)n[0], (int)n[1], 2, L - l + 1));
                } else {   // one workgroup per line
                    if (t2) WL_TRY(launch_tail2<T>(st, taps, cur, 0, y, 0, cur_st.s[1], b.full.s[1], (int)nlines, (int)n[0], 1, 1, L - l + 1));
                    else WL_TRY(launch_tail<T>(st, taps, cur, 0, y, 0, cur_st.s[1], b.full.s[1], (int)nlines, (int)n[0], 1, 1, L - l + 1));
                }
                if (!dominant) dominant = t2 ? "k_tail2_fwd" : "k_tail_fwd";
                break;
            }
        }
        bool done = false;
        // ---- 1-D multi-level tile kernel: up to 4 levels per pass over HBM ----
        if (fastF && lines && env_int("WL_NO_MULTI", 0) == 0 && n[0] > line_cap && (n[0] % (8 * VEC)) == 0 &&
            cur_st.s[0] == 1 && aligned16(cur) && aligned16(y) &&
            (nlines == 1 || ((cur_st.s[1] % VEC) == 0 && (b.full.s[1] % VEC) == 0))) {
            int NL = L - l + 1;
            // big stages are bandwidth-bound: 4 levels keep the halo (F-2)(2^NL - 1) small; small stages are launch-latency
            // bound: up to 8 levels per launch saves whole launches (1-D db2 2^20 f64: 23.8 -> 17.3 us with 6, 15.9 with 8)
            int nl_auto = 4;
            if (n[0] * nlines < ((int64_t)1 << 22)) {
                // levels left before the one-workgroup tail can take the line: split them evenly over the fewest launches
                int lg_n = 0, lg_cap = 0;
                while (((int64_t)2 << lg_n) <= n[0]) ++lg_n;
                while (((int64_t)2 << lg_cap) <= line_cap) ++lg_cap;
                int r = lg_n - lg_cap;
                if (r > L - l + 1) r = L - l + 1;
                if (r < 1) r = 1;
                // (up to 8 levels per launch: the halo (F-2)(2^NL - 1) is then a multiple of the tile, but these lines are a few MiB
                //  in L2 and a launch saved is ~1 us net -- r04: 2^20 db4 19.0 -> 18.0 us, L = 8: 12.4 -> 11.3)
                int per = env_int("WL_NL_SMALL_MAX", 8);
                if (per < 1) per = 1;
                if (per > 8) per = 8;
                const int stages = (r + per - 1) / per;
                nl_auto = (r + stages - 1) / stages;
            }
            const int nlmax = env_int("WL_NLMAX", nl_auto);
            if (NL > nlmax) NL = nlmax;
            while (NL > 1 && (n[0] % ((int64_t)(1 << NL) * 4 * VEC)) != 0) --NL;   // alignment of every level's stores
            const bool lastm = (l + NL - 1 == L);
            // after NL levels the approximation has n >> NL samples per line
            T *sd = lastm ? y : llbuf;
            const int64_t sls = lastm ? b.full.s[1] : (n[0] >> NL);
            WL_DISPATCH_F(F, WL_TRY((launch_fwd1d_multi<T, FF>(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], sd, sls,
                                                               n[0], nlines, NL)));
                          done = true);
            if (done) {
                if (!dominant) dominant = "k_fwd1d_multi";
                lstep = NL;
                int64_t hn2[3] = {n[0] >> NL, n[1], n[2]};
                cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                continue;
            }
        }
        // ---- LDS-exchange streaming kernel: one or two fused 2-D levels (f32, wl_fwd2d.hip) ----
        if constexpr (sizeof(T) == 4) {
            if (fastF && two_d && env_int("WL_LDS2D", 1) && n[0] >= env_int("WL_LDS2D_MIN_ROWS", 256) && cur_st.s[0] == 1 &&
                (cur_st.s[1] % VEC) == 0 && aligned16(cur) && (b.full.s[1] % VEC) == 0 && aligned16(y) && aligned16(llbuf)) {
                int nlev = 0;
                if ((L - l + 1) >= 2 && env_int("WL_FUSE2", 1) && fwd2d_lds_ok(F, 2, n[0], n[1]) &&
                    n[0] * n[1] >= (int64_t)opt("WL_LDS_PAIR_MIN", (long long)1 << 24))
                    nlev = 2;
                else if (fwd2d_lds_ok(F, 1, n[0], n[1]))
                    nlev = 1;
                // ... and FOUR on request (WL_FUSE4 = 1) when the two levels after the pair would be the tiles of k_fwd2d_tileB: the same tiles run
                //     inside the pair's launch, each as soon as the pair workgroups it reads from have published their columns.  Built and
                //     measured in round 6 (profiles/r06_c3_timeline.md): the tile work does hide under the pair (the tiles end 5-7 us after
                //     the last pair workgroup instead of 19-20), but the pair is HBM-bound and its own end slips by as much -- 8192^2 L = 13
                //     within +-1 us of the two-launch chain in five interleaved A/B runs.  Off by default: no gain to pay for in-launch hand-over.
                if (nlev == 2 && (L - l + 1) >= 4 && opt("WL_FUSE4", 0) != 0 && tl_sync != nullptr && env_int("WL_TILEB", 1) &&
                    (n[0] >> 2) <= env_int("WL_TILEB_MAX", 2048) && (n[1] >> 2) <= env_int("WL_TILEB_MAX", 2048) &&
                    (n[0] >> 2) * (n[1] >> 2) >= (int64_t)env_int_raw("WL_TILEB_MIN", 1 << 21) && fwd2d_pair_tile_ok(F, n[0], n[1], cu_count)) {
                    const bool last4 = (l + 3 == L);
                    T *ll2buf = llbuf, *ll4buf = pp ? w.A : w.B;
                    if (aligned16(ll4buf)) {
                        WL_TRY(fwd2d_pair_tile_launch(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], ll2buf, n[0] >> 2, last4 ? (T *)nullptr : ll4buf,
                                                      n[0] >> 4, n[0], n[1], cu_count, tl_sync));
                        if (!dominant) dominant = "k_fwd2d_pair_tile";
                        lstep = 4;
                        int64_t hn4[3] = {n[0] >> 4, n[1] >> 4, n[2]};
                        cur = ll4buf; cur_st = dense_strides(hn4);
                        continue;
                    }
                }
                if (nlev) {
                    const bool lastp = (l + nlev - 1 == L);
                    T *lld = lastp ? y : llbuf;
                    const int64_t ldd = lastp ? b.full.s[1] : (n[0] >> nlev);
                    WL_TRY(fwd2d_lds_launch(st, taps, nlev, l == 1, cur, cur_st.s[1], y, b.full.s[1], lld, ldd, n[0], n[1], cu_count));
                    if (!dominant) dominant = (nlev == 2) ? "k_fwd2d_pair" : "k_fwd2d_lds";
                    lstep = nlev;
                    int64_t hn2[3] = {n[0] >> nlev, n[1] >> nlev, n[2]};
                    cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                    continue;
                }
            }
        }
        // ---- Float64: the LDS-exchange level kernel with two rows per lane (wl_fwd2d64.hip) ----
        if constexpr (sizeof(T) == 8) {
            if (fastF && two_d && env_int("WL_LDS2D", 1) && n[0] >= env_int("WL_LDS2D_MIN_ROWS", 256) && cur_st.s[0] == 1 &&
                (cur_st.s[1] % VEC) == 0 && aligned16(cur) && (b.full.s[1] % VEC) == 0 && aligned16(y) && aligned16(llbuf) &&
                fwd2d_lds64_ok(F, n[0], n[1])) {
                // two levels per launch from 2^22 elements upwards (the Float64 pair kernel, wl_pair2d64.hip)
                if ((L - l + 1) >= 2 && env_int("WL_FUSE2", 1) && fwd2d_pair64_ok(F, n[0], n[1]) &&
                    n[0] * n[1] >= (int64_t)opt("WL_LDS_PAIR_MIN64", (long long)1 << 23)) {
                    const bool lastp = (l + 1 == L);
                    T *lld2 = lastp ? y : llbuf;
                    const int64_t ldd2 = lastp ? b.full.s[1] : (n[0] >> 2);
                    WL_TRY(fwd2d_pair64_launch(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], lld2, ldd2, n[0], n[1], cu_count));
                    if (!dominant) dominant = "k_fwd2d_pair64";
                    lstep = 2;
                    int64_t hn4[3] = {n[0] >> 2, n[1] >> 2, n[2]};
                    cur = llbuf; cur_st = dense_strides(hn4); pp ^= 1;
                    continue;
                }
                T *lld = last ? y : llbuf;
                const int64_t ldd = last ? b.full.s[1] : (n[0] >> 1);
                WL_TRY(fwd2d_lds64_launch(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], lld, ldd, n[0], n[1], cu_count));
                if (!dominant) dominant = "k_fwd2d_lds64";
                lstep = 1;
                int64_t hn2[3] = {n[0] >> 1, n[1] >> 1, n[2]};
                cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                continue;
            }
        }
        // ---- streaming 2-D, two levels fused (f32) ----
        if constexpr (sizeof(T) == 4) {
            if (fastF && F <= 8 && two_d && env_int("WL_FUSE2", 1) && (L - l + 1) >= 2 && n[0] >= 512 && (n[0] % 16) == 0 &&
                n[0] * n[1] >= (int64_t)env_int_raw("WL_FUSE2_MIN", 1 << 24) &&
                n[1] >= 64 && (n[1] % 32) == 0 && cur_st.s[0] == 1 && (cur_st.s[1] % VEC) == 0 && aligned16(cur) &&
                (b.full.s[1] % VEC) == 0 && aligned16(y) && aligned16(llbuf)) {
                const bool lastp = (l + 1 == L);
                T *lld = lastp ? y : llbuf;
                const int64_t ldd = lastp ? b.full.s[1] : (n[0] >> 2);
                bool ok = false;
                WL_DISPATCH_F(F, WL_TRY((launch_fwd2d_pair<T, FF>(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], lld, ldd,
                                                                  n[0], n[1], cu_count)));
                              ok = true);
                if (ok) {
                    if (!dominant) dominant = "k_fwd2d_stream2";
                    lstep = 2;
                    int64_t hn2[3] = {n[0] >> 2, n[1] >> 2, n[2]};
                    cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                    continue;
                }
            }
        }
        // ---- streaming 2-D level ----
        if (fastF && two_d && n[0] >= 64 * VEC && (n[0] % 8) == 0 && n[1] >= 16 && (n[1] % 16) == 0 &&
            cur_st.s[0] == 1 && (cur_st.s[1] % VEC) == 0 && aligned16(cur) &&
            (b.full.s[1] % VEC) == 0 && aligned16(y) && aligned16(llbuf)) {
            WL_DISPATCH_F(F, WL_TRY((launch_fwd2d<T, FF>(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1],
                                                         last ? (T *)nullptr : llbuf, ll_st.s[1], n[0], n[1], cu_count)));
                          done = true);
            if (done && !dominant) dominant = "k_fwd2d_stream";
        }
        // ---- streaming 1-D level (vector, or every column of a matrix) ----
        if (!done && fastF && lines && n[0] >= 512 && (n[0] % 8) == 0 && cur_st.s[0] == 1 && aligned16(cur) &&
            aligned16(y) && aligned16(llbuf) && (nlines == 1 || ((cur_st.s[1] % VEC) == 0 && (b.full.s[1] % VEC) == 0)) &&
            true) {
            T *sd = last ? y : llbuf;
            int64_t sls = last ? b.full.s[1] : ll_st.s[1];
            WL_DISPATCH_F(F, WL_TRY((launch_fwd1d<T, FF>(st, taps, l == 1, cur, cur_st.s[1], sd, sls, y + (n[0] >> 1),
                                                         b.full.s[1], n[0], nlines, cu_count)));
                          done = true);
            if (done && !dominant) dominant = "k_fwd1d_stream";
        }
        // ---- 12..20 taps, Float32, blocks whose rows tile into strips of 256: one pass per level (wl_fwd2d_long.hip) ----
        if constexpr (sizeof(T) == 4) {
            if (!done && path == 0 && two_d && env_int("WL_LONG2D", 1) && n[0] >= env_int("WL_LONG2D_MIN_ROWS", 256) && b.full.s[0] == 1 &&
                cur_st.s[0] == 1 && fwd2d_long_ok(F, n[0], n[1]) && (cur_st.s[1] % VEC) == 0 && aligned16(cur) && (b.full.s[1] % VEC) == 0 &&
                aligned16(y) && aligned16(llbuf)) {
                T *lld = last ? y : llbuf;
                const int64_t ldd = last ? b.full.s[1] : (n[0] >> 1);
                WL_TRY(fwd2d_long_launch(st, taps, l == 1, cur, cur_st.s[1], y, b.full.s[1], lld, ldd, n[0], n[1], cu_count));
                if (!dominant) dominant = "k_fwd2d_lds_long";
                lstep = 1;
                int64_t hn2[3] = {n[0] >> 1, n[1] >> 1, n[2]};
                cur = llbuf; cur_st = dense_strides(hn2); pp ^= 1;
                continue;
            }
        }
        // ---- long filters (12..24 taps): line kernel with multi-lane halo, 2-D as axis pass + line pass (wl_axis.hip) ----
        if (!done && path == 0 && long_filter_ok(F) && env_int("WL_NO_LONGF", 0) == 0 && b.full.s[0] == 1 && cur_st.s[0] == 1) {
            hipError_t e = hipSuccess;
            const int64_t h0 = n[0] >> 1, h1 = n[1] >> 1, ldy = b.full.s[1];
            if (lines) {
                T *sd = last ? y : llbuf;
                const int64_t sls = last ? ldy : ll_st.s[1];
                done = long_lines_fwd_level<T>(st, taps, cur, cur_st.s[1], sd, sls, y + h0, ldy, n[0], nlines, cu_count, &e);
                WL_TRY(e);
            } else if (two_d && long_shape2d_ok(F, n[0], n[1]) && (ldy % VEC) == 0 &&
                       (cur_st.s[1] % VEC) == 0 && aligned16(cur) && aligned16(y) && aligned16(llbuf)) {
                // rows (dim 2) into T0 = [s-columns | d-columns], then the columns of T0 as lines with the LL quadrant routed on
                if (!w.T0) return WL_RETRY_GEN;
                done = long_axis_level<T>(st, taps, 1, cur, cur_st.s[1], w.T0, n[0], n[0], n[1], cu_count, &e);
                WL_TRY(e);
                if (done) {
                    T *lld = last ? y : llbuf;
                    const int64_t ldd = last ? ldy : h0;
                    bool ok = long_lines_fwd_level<T>(st, taps, w.T0, n[0], lld, ldd, y + h0, ldy, n[0], h1, cu_count, &e);
                    WL_TRY(e);
                    ok = ok && long_lines_fwd_level<T>(st, taps, w.T0 + h1 * n[0], n[0], y + h1 * ldy, ldy, y + h1 * ldy + h0, ldy, n[0], h1,
                                                       cu_count, &e);
                    WL_TRY(e);
                    if (!ok) return WL_EINVAL_ARG;      // (eligibility is identical for the three launches)
                }
            }
            if (done && !dominant) dominant = (vlong_only(F) || n[0] < 512) ? "k_vl_lines" : "k_long_lines";
        }
        // ---- 3-D level from three single-axis streaming passes (wl_axis.hip) ----
        if (!done && fastF && b.nd == 3 && b.nt == 3 && env_int("WL_NO_FAST3D", 0) == 0 && cur_st.s[0] == 1 && b.full.s[0] == 1) {
            hipError_t e3 = hipSuccess;
            const char *k3 = "k_fwd_axis_stream";
            if (!w.T0) return WL_RETRY_GEN;
            done = fast3d_fwd_level<T>(st, taps, cur, cur_st.s[1], cur_st.s[2], y, b.full.s[1], b.full.s[2],
                                       last ? (T *)nullptr : llbuf, n, w.T0, w.T1, cu_count, &e3, &k3);
            WL_TRY(e3);
            if (done && !dominant) dominant = k3;
        }
        // ---- 2-D level of any even extents, F <= 10: one LDS-tile launch instead of two generic passes (wl_gtile.hip) ----
        if (!done && fastF && two_d && path == 0 && env_int("WL_GTILE", 1) && b.full.s[0] == 1 && cur_st.s[0] == 1 && gtile_ok(F, n[0], n[1])) {
            WL_TRY(gtile_launch<T>(st, taps, 1, cur, cur_st.s[1], y, b.full.s[1], last ? (T *)nullptr : llbuf, ll_st.s[1], (int)n[0], (int)n[1]));
            if (!dominant) dominant = "k_fwd2d_gtile";
            done = true;
        }
        // ---- generic level (any rank / size / filter) ----
        if (!done) {
            if (b.nt > 1 && !w.T0) return WL_RETRY_GEN;
            const T *in = cur;
            Strides3 in_st = cur_st;
            int tog = 0;
            for (int a = b.nt - 1; a >= 0; --a) {
                // F <= 10: four pairs per thread from a register window, compile-time taps (wl_anyaxis.hip); else one output per thread
                const bool any = path == 0 && env_int("WL_ANYAXIS", 1) && any_axis_ok(F, ext, a);
                if (a != 0) {
                    T *out = tog ? w.T1 : w.T0;
                    if (any) WL_TRY(any_axis_pass<T>(st, taps, 1, in, in_st, out, box_st, (T *)nullptr, box_st, ext, a, lo));
                    else WL_TRY(generic_fwd_filter_pass<T>(st, taps, in, in_st, out, box_st, (T *)nullptr, box_st, ext, a, lo));
                    in = out; in_st = box_st; tog ^= 1;
                } else {
                    if (any) WL_TRY(any_axis_pass<T>(st, taps, 1, in, in_st, y, b.full, last ? (T *)nullptr : llbuf, ll_st, ext, a, lo));
                    else WL_TRY(generic_fwd_filter_pass<T>(st, taps, in, in_st, y, b.full, last ? (T *)nullptr : llbuf, ll_st, ext, a, lo));
                    if (!dominant) dominant = any ? "k_fwd_any" : "k_generic_fwd_filter";
                }
            }
        }
        cur = llbuf; cur_st = ll_st; pp ^= 1;
    }
    if (kernel_name) *kernel_name = dominant ? dominant : "none";
    return WL_OK;
#undef WL_TRY
}

// One forward level of `nlines` independent lines with the streaming kernel (used by the packet
// transform for fully split depths).  Returns false when the shape/filter is not eligible.
template <typename T>
bool fast_lines_fwd_level(hipStream_t st, const Taps<T> &taps, const T *src, int64_t src_ls, T *sdst, int64_t s_ls,
                          T *ddst, int64_t d_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err)
{
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    *err = hipSuccess;
    if ((F % 2) != 0 || F > 10 || n < 512 || (n % 8) != 0 || !aligned16(src) || !aligned16(sdst) ||
        !aligned16(ddst) || (src_ls % VEC) != 0 || (s_ls % VEC) != 0 || (d_ls % VEC) != 0)
        return false;
    bool done = false;
    WL_DISPATCH_F(F, *err = launch_fwd1d<T, FF>(st, taps, false, src, src_ls, sdst, s_ls, ddst, d_ls, n, nlines, cu_count);
                  done = true);
    return done;
}
template bool fast_lines_fwd_level<float>(hipStream_t, const Taps<float> &, const float *, int64_t, float *, int64_t, float *,
                                          int64_t, int64_t, int64_t, int, hipError_t *);
template bool fast_lines_fwd_level<double>(hipStream_t, const Taps<double> &, const double *, int64_t, double *, int64_t,
                                           double *, int64_t, int64_t, int64_t, int, hipError_t *);

// The fused 2-D level kernel on a batch of independent planes (the dim-2 + dim-1 passes of a 3-D level): plane p of
// `src` (n0 x n1, dense, stride n0*n1) -> plane p of y (strides y1, y2); the first nll planes send their approximation
// quadrant to plane p of ll (dense h0 x h1), the others write it into y.
template <typename T>
bool fwd2d_planes(hipStream_t st, const Taps<T> &taps, const T *src, T *y, int64_t y1, int64_t y2, T *ll, int64_t n0, int64_t n1,
                  int64_t nplanes, int nll, int cu_count, hipError_t *err)
{
    constexpr int VEC = 16 / sizeof(T);
    const int F = taps.F;
    *err = hipSuccess;
    if ((F % 2) != 0 || F > 10 || n0 < 64 * VEC || (n0 % 8) != 0 || n1 < 16 || (n1 % 16) != 0 || (y1 % VEC) != 0 || (y2 % VEC) != 0 ||
        !aligned16(src) || !aligned16(y) || (ll && !aligned16(ll)) || nplanes > 65535)
        return false;
    if constexpr (sizeof(T) == 4) {
        // Float32 planes of >= 256 rows: the LDS-exchange level kernel, batched over blockIdx.y (wl_fwd2d.hip)
        if (opt("WL_LDS2D", 1) != 0 && n0 >= opt("WL_LDS2D_MIN_ROWS", 256) && fwd2d_lds_ok(F, 1, n0, n1)) {
            *err = fwd2d_lds_launch(st, taps, 1, false, src, n0, y, y1, ll, n0 >> 1, n0, n1, cu_count, nplanes, n0 * n1, y2,
                                    (n0 >> 1) * (n1 >> 1), nll);
            return true;
        }
    }
    bool ok = false;
    WL_DISPATCH_F(F, *err = launch_fwd2d_r<T, FF, VEC>(st, taps, false, src, n0, y, y1, ll, n0 >> 1, n0, n1, cu_count, nplanes, n0 * n1, y2,
                                                        (n0 >> 1) * (n1 >> 1), nll);
                  ok = true);
    return ok;
}
template bool fwd2d_planes<float>(hipStream_t, const Taps<float> &, const float *, float *, int64_t, int64_t, float *, int64_t, int64_t,
                                  int64_t, int, int, hipError_t *);
template bool fwd2d_planes<double>(hipStream_t, const Taps<double> &, const double *, double *, int64_t, int64_t, double *, int64_t,
                                   int64_t, int64_t, int, int, hipError_t *);

template int filter_fwd_levels<float>(void *, bool, int, int, hipStream_t, const BoxSpec &, float *, const float *,
                                      const Taps<float> &, int, const char **, int *);
template int filter_fwd_levels<double>(void *, bool, int, int, hipStream_t, const BoxSpec &, double *, const double *,
                                       const Taps<double> &, int, const char **, int *);

}  // namespace wl
