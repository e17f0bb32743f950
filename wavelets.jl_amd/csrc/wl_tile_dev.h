// wl_tile_dev.h -- device code of the cache-resident forward tiles (wl_tile.hip) that the fused pair + tile launch (wl_pair2d.hip)
// shares: tile geometry, one level inside a tile, and the body of k_fwd2d_tileB (two levels of a 64 x 64 piece, first level straight
// from global memory).  See wl_tile.hip for the design.
#pragma once
#include "wl_fast.h"
#include "wl_dev.h"

// (the including translation unit names the stamp family of tileB_body: WL_TB_STAMP(workgroup, slot), see WL_STAMP_AT)
#ifndef WL_TB_STAMP
#define WL_TB_STAMP(wg, k) do { } while (0)
#endif

namespace wl {

template <typename T, int F>
struct TileArgs {
    const T *src; int64_t lds;          // input block M x N
    T *y; int64_t ldy;
    T *ll; int64_t ldll;                // approximation after NL levels (dense buffer or y itself)
    int M, N;
    TapsF<T, F> tp;
};

// dim-1 pass of a thread: s rows 4q .. 4q+3 and d rows 4q+DS .. 4q+DS+3 from window rows 8q .. 8q+WINR-1.  DS = the shift of the stored
// d rows, a multiple of 4 >= (F-2)/2 (a group of four d rows never straddles the periodic wrap): 4 up to 10 taps (WINR = 16, the
// layout of rounds 2-4), 8 for 12..18 taps (24 rows), 12 for 20 taps (32 rows) -- round 5: the tiles also serve the 12..20-tap filters
template <int F>
struct TileGeomF {
    static constexpr int SHD = (F - 2) / 2;
    static constexpr int DS = (F <= 10) ? 4 : ((SHD + 3) / 4) * 4;
    static constexpr int WINR = (((F + 6 > 2 * DS + 8) ? F + 6 : 2 * DS + 8) + 3) & ~3;
    static constexpr int HR = WINR - 8;
};
// extents of the tile at level l (l = 0: the launch's input), OT owned input samples per side (64; 32 for the three-level tiles of
// blocks with too few 64-sample tiles to fill the chip, round 6)
template <int F, int NL, int l, int OT = 64>
struct TileDim {
    static constexpr int HR = TileGeomF<F>::HR, HC = F - 2;        // one-sided halos per level: rows (d rows shifted by DS), columns
    static constexpr int R = 2 * TileDim<F, NL, l + 1, OT>::R + HR;    // rows / columns of this level's input that the tile needs
    static constexpr int C = 2 * TileDim<F, NL, l + 1, OT>::C + HC;
};
template <int F, int NL, int OT>
struct TileDim<F, NL, NL, OT> {
    static constexpr int R = OT >> NL, C = OT >> NL;
};

template <int F, int NL, int OT = 64>
struct TileLds {
    // X0 | T | X1 | X2 (float offsets); leading dimensions padded to a multiple of 4 rows plus 4 (bank spread, 16-byte aligned)
    static constexpr int ldx(int r) { return ((r + 3) & ~3) + 4; }
    static constexpr int R0 = TileDim<F, NL, 0, OT>::R, C0 = TileDim<F, NL, 0, OT>::C;
    static constexpr int R1 = TileDim<F, NL, (NL >= 1 ? 1 : 0), OT>::R, C1 = TileDim<F, NL, (NL >= 1 ? 1 : 0), OT>::C;
    static constexpr int R2 = TileDim<F, NL, (NL >= 2 ? 2 : NL), OT>::R, C2 = TileDim<F, NL, (NL >= 2 ? 2 : NL), OT>::C;
    static constexpr int X0 = 0;
    static constexpr int T = X0 + ldx(R0) * C0;
    static constexpr int X1 = T + ldx(R0) * (C1 + OT / 2);         // T: R0 rows x (C1 s-columns + OT / 2 owned d-columns)
    static constexpr int X2 = X1 + ldx(R1) * C1;
    static constexpr int TOTAL = X2 + ldx(R2) * C2 + 16;
};


// One level inside the tile.  X: input R x C (leading dimension ldx), T: scratch, XN: next level's input (RN x CN) in LDS.
// OWN = owned outputs per side at this level (32, 16, 8); (r0h, c0h) = tile origin in this level's OUTPUT coordinates;
// hm, hn = half extents of this level's block; LAST: the approximation goes to global memory (ll) instead of XN.
// The taps travel BY VALUE (round 6).  By reference, hipcc kept the kernel argument's tap array addressable -- the SLP vectoriser loads
// runs of consecutive taps as vectors -- and parked it in scratch: 28-48 bytes per lane, reloaded before every pass, in every
// k_fwd2d_tile instance (k_fwd2d_tile<float, 8, 3>: 125 VGPRs + 32 B scratch -> 68 VGPRs, none; the 20-tap instances 56 B -> 0).
#ifdef WL_TILE_TAPS_BYREF
#define WL_TILE_TAPS const TapsF<TT, F> &
#else
#define WL_TILE_TAPS const TapsF<TT, F>
#endif
template <typename TT, int F, int R, int C, int RN, int CN, int OWN, bool LAST>
__device__ __forceinline__ void tile_level(const TT *X, int ldX, TT *T, int ldT, TT *XN, int ldN, WL_TILE_TAPS tp,
                                           TT *y, int64_t ldy, TT *ll, int64_t ldll, int r0h, int c0h, int hm, int hn, int tid,
                                           int nthr)
{
    typedef TT F4t __attribute__((ext_vector_type(4)));
    constexpr int SH = (F - 2) / 2;
    constexpr int RQ = (R + 3) / 4;                 // row quads of the input
    // g[m] = (-1)^m h[m] exactly (make_taps): only the scaling taps occupy SGPRs, a detail term multiplies by the negated tap
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -tp.h[m] : tp.h[m]; };
    // ---- dim-2 pass: X (R x C) -> T: columns [0, CN) = s (window columns 2k .. 2k+F-1), columns [CN, CN+OWN) = d[k + SH]
    for (int it = tid; it < RQ * CN; it += nthr) {
        const int iq = it % RQ, k = it / RQ;
        const TT *p = X + 4 * iq + (2 * k) * ldX;
        F4t x0 = *reinterpret_cast<const F4t *>(p);
        F4t s = tp.h[0] * x0, d = gq(F - 1) * x0;
#pragma unroll
        for (int m = 1; m < F; ++m) {
            const F4t xm = *reinterpret_cast<const F4t *>(p + m * ldX);
            s = s + tp.h[m] * xm;
            d = d + gq(F - 1 - m) * xm;
        }
        *reinterpret_cast<F4t *>(T + 4 * iq + k * ldT) = s;
        if (k < OWN) *reinterpret_cast<F4t *>(T + 4 * iq + (CN + k) * ldT) = d;
    }
    lds_barrier();
    // ---- dim-1 pass: column c of T, rows 8q .. 8q+WINR-1 -> s rows 4q .. 4q+3, d rows 4q+DS .. 4q+DS+3
    constexpr int QG = (RN + 3) / 4;                // groups of four output rows (covers the RN approximation rows needed below)
    constexpr int DS = TileGeomF<F>::DS, WINR = TileGeomF<F>::WINR;
    for (int it = tid; it < QG * (CN + OWN); it += nthr) {
        const int q = it % QG, c = it / QG;
        const TT *p = T + 8 * q + c * ldT;
        TT E[WINR];
#pragma unroll
        for (int v = 0; v < WINR / 4; ++v) {
            const F4t t = *reinterpret_cast<const F4t *>(p + 4 * v);
            E[4 * v] = t.x; E[4 * v + 1] = t.y; E[4 * v + 2] = t.z; E[4 * v + 3] = t.w;
        }
        F4t so, dO;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            TT s = tp.h[0] * E[2 * j];
#pragma unroll
            for (int m = 1; m < F; ++m) s = s + tp.h[m] * E[2 * j + m];
            TT d = gq(F - 1) * E[2 * j + 2 * DS + 2 - F];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + gq(m) * E[2 * j + 2 * DS + 1 - m];
            so[j] = s; dO[j] = d;
        }
        const bool is_s = c < CN;
        // approximation of an s-column: next level's input (all RN rows), or global when this is the launch's last level
        if (is_s) {
            if (!LAST) *reinterpret_cast<F4t *>(XN + 4 * q + c * ldN) = so;
            else if (c < OWN && 4 * q < OWN) store_pol<WL_P_TILE_LL>(reinterpret_cast<F4t *>(ll + (r0h + 4 * q) + (int64_t)(c0h + c) * ldll), so);
        }
        // details: owned rows / columns only
        const int cc = is_s ? c : c - CN;
        if (cc < OWN && 4 * q < OWN) {
            int64_t col;
            if (is_s) col = c0h + cc;                                   // s along dim 2
            else { int kd = c0h + cc + SH; if (kd >= hn) kd -= hn; col = hn + kd; }
            int rd = r0h + 4 * q + DS;
            if (rd >= hm) rd -= hm;
            TT *yc = y + col * ldy;
            store_pol<WL_P_TILE3_ST>(reinterpret_cast<F4t *>(yc + hm + rd), dO);                // ds or dd
            if (!is_s) store_pol<WL_P_TILE3_ST>(reinterpret_cast<F4t *>(yc + (r0h + 4 * q)), so);  // sd
        }
    }
    lds_barrier();
}

// k_fwd2d_tileB<F>: TWO fused levels of a block that is too big for one resident round of the kernel above (2048^2: 1024 tiles).
// Same tiles, same one-sided halos, same arithmetic -- but the input tile is NOT staged: the dim-2 pass of the first level
// reads its window columns straight from global memory (the overlapping windows of neighbouring threads are L1 / L2 hits), so a
// workgroup holds only the dim-2 results (26 KB) and the first approximation (7 KB) in LDS: 33 KB and 256 threads instead of
// 62 KB and 1024.  Four workgroups per CU stay resident -- all 1024 tiles of a 2048^2 block at once -- where the staging
// kernel kept two, each waiting on its own loads, barriers and partially idle passes (PMC, 2048^2: 64 % of the wave cycles
// waiting, 41 % of the LDS cycles bank conflicts).
// LDPOL: 0 = the policy of WL_P_TILE_LD (loads the compiler counts); 2 = `sc1` loads (L1 bypassed, served coherently: the input was
// written with write-through stores by workgroups of the SAME launch -- the fused pair + tile launch of wl_pair2d.hip)
template <int F, int LDPOL>
__device__ __forceinline__ void tileB_body(const TileArgs<float, F> &a, const TapsF<float, F> &tp, float *Ts, float *X1s, const int tbx, const int tby, const int tid,
                                           const int nthr, [[maybe_unused]] const int wgid)
{
    typedef float T;
    typedef TileLds<F, 2> L;
    typedef T F4t __attribute__((ext_vector_type(4)));
    constexpr int ldT = L::ldx(L::R0), ld1 = L::ldx(L::R1);
    constexpr int R0 = L::R0, R1 = L::R1, C1 = L::C1;
    constexpr int RQ = (R0 + 3) / 4;
    const int r0 = tbx * 64, c0 = tby * 64;
    const int hm = a.M >> 1, hn = a.N >> 1;
    if (tid == 0) WL_TB_STAMP(wgid, 0);
    // ---- level 1, dim-2 pass from global: T columns [0, C1) = s (window columns 2k .. 2k+F-1), [C1, C1+32) = d[k + SH] ----
    // A thread owns four rows and KG = 4 consecutive output columns: F + 6 column loads (all in flight together) instead of 4 F,
    // and the whole pass is one round of the workgroup (22 row quads x 10 column groups = 220 of 256 threads).
    {
        constexpr int KG = 4, NG = (C1 + KG - 1) / KG, NC = F + 2 * (KG - 1);
        for (int it = tid; it < RQ * NG; it += nthr) {
            const int iq = it % RQ, k0 = (it / RQ) * KG;
            int gr = r0 + 4 * iq;
            if (gr >= a.M) gr -= a.M;
            int gc = c0 + 2 * k0;
            if (gc >= a.N) gc -= a.N;
            F4t xw[NC];
#pragma unroll
            for (int m = 0; m < NC; ++m) {
                int c = gc + m;
                if (c >= a.N) c -= a.N;
                if constexpr (LDPOL == 2) gload16_sc1(xw[m], a.src + gr + (int64_t)c * a.lds);
                else xw[m] = load_pol<WL_P_TILE_LD != 0>(reinterpret_cast<const F4t *>(a.src + gr + (int64_t)c * a.lds));
            }
            if constexpr (LDPOL == 2) drain_ring(xw);
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                const int k = k0 + kk;
                F4t sv = tp.h[0] * xw[2 * kk], dv = tp.g[F - 1] * xw[2 * kk];
#pragma unroll
                for (int m = 1; m < F; ++m) {
                    sv = sv + tp.h[m] * xw[2 * kk + m];
                    dv = dv + tp.g[F - 1 - m] * xw[2 * kk + m];
                }
                if (k < C1) *reinterpret_cast<F4t *>(Ts + 4 * iq + k * ldT) = sv;
                if (k < 32) *reinterpret_cast<F4t *>(Ts + 4 * iq + (C1 + k) * ldT) = dv;
            }
        }
    }
    if (tid == 0) WL_TB_STAMP(wgid, 1);
    lds_barrier();
    if (tid == 0) WL_TB_STAMP(wgid, 2);
    // ---- level 1, dim-1 pass (the second half of tile_level) ----
    {
        constexpr int SH = (F - 2) / 2;
        constexpr int QG = (R1 + 3) / 4, OWN = 32;
        const int r0h = r0 >> 1, c0h = c0 >> 1;
        for (int it = tid; it < QG * (C1 + OWN); it += nthr) {
            const int q = it % QG, c = it / QG;
            const T *p = Ts + 8 * q + c * ldT;
            T E[16];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const F4t t = *reinterpret_cast<const F4t *>(p + 4 * v);
                E[4 * v] = t.x; E[4 * v + 1] = t.y; E[4 * v + 2] = t.z; E[4 * v + 3] = t.w;
            }
            F4t so, dO;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                T sv = tp.h[0] * E[2 * j];
#pragma unroll
                for (int m = 1; m < F; ++m) sv = sv + tp.h[m] * E[2 * j + m];
                T dv = tp.g[F - 1] * E[2 * j + 10 - F];
#pragma unroll
                for (int m = F - 2; m >= 0; --m) dv = dv + tp.g[m] * E[2 * j + 9 - m];
                so[j] = sv; dO[j] = dv;
            }
            const bool is_s = c < C1;
            if (is_s) *reinterpret_cast<F4t *>(X1s + 4 * q + c * ld1) = so;
            const int cc = is_s ? c : c - C1;
            if (cc < OWN && 4 * q < OWN) {
                int64_t col;
                if (is_s) col = c0h + cc;
                else { int kd = c0h + cc + SH; if (kd >= hn) kd -= hn; col = hn + kd; }
                int rd = r0h + 4 * q + 4;
                if (rd >= hm) rd -= hm;
                T *yc = a.y + col * a.ldy;
                store_pol<WL_P_TILE_ST>(reinterpret_cast<F4t *>(yc + hm + rd), dO);                  // ds or dd
                if (!is_s) store_pol<WL_P_TILE_ST>(reinterpret_cast<F4t *>(yc + (r0h + 4 * q)), so);  // sd
            }
        }
    }
    lds_barrier();
    if (tid == 0) WL_TB_STAMP(wgid, 3);
    // ---- level 2: LDS -> LDS as in the staging kernel ----
    tile_level<T, F, L::R1, L::C1, 16, 16, 16, true>(X1s, ld1, Ts, ld1, nullptr, 0, tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 2, c0 >> 2, hm >> 1,
                                                  hn >> 1, tid, nthr);
    if (tid == 0) WL_TB_STAMP(wgid, 4);
}


}  // namespace wl
