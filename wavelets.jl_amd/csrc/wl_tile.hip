// wl_tile.hip -- forward 2-D filter-bank levels of the cache-resident blocks (128^2 .. 2048^2): NL = 1..3 (Float64: 1..2) fused
// levels per launch, one 64 x 64 piece of the block per workgroup, everything after the first read in LDS.
//
// Replaces k_fwd2d_multi (wl_fwd.hip) on these sizes.  Same idea -- recompute halos instead of exchanging them, so that
// workgroups are independent -- but:
//   * ONE-SIDED halos.  Along dim 2 the detail column computed from window columns 2k .. 2k+F-1 is d[k + (F-2)/2] (the
//     window of s[k]); along dim 1 a thread produces s rows 4q .. 4q+3 and d rows 4q+4 .. 4q+7 from window rows
//     8q .. 8q+15.  Every window starts at the tile origin: the tile with halo is (64 + 24)(64 + 18) samples for two levels
//     instead of (64 + 36)^2 -- 1.76x the payload instead of 2.44x, and the loads / stores stay 16-byte aligned.
//   * register-blocked passes: a thread owns four rows (one ds_read_b128 per window column) in the dim-2 pass and a
//     16-row window (four aligned ds_read_b128) in the dim-1 pass, instead of 2F-2 scalar LDS reads per output pair;
//   * all extents are compile-time constants (no integer division at run time).
// Arithmetic: the closed forms of wl_internal.h in the reference's order, no FMA -- bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"
WL_STAMP_DECL(tile)
WL_STAMP_DECL(tileB)
#define WL_TB_STAMP(wg, k) WL_STAMP_AT(tileB, wg, k)
#include "wl_tile_dev.h"

namespace wl {

// (12..20 taps: 512 threads -- with 1024 the 128-VGPR budget spilled 28 registers of the 24- / 32-row dim-1 window to scratch)
// OT = 32 (round 6): a 512^2 block is 64 tiles of 64^2 -- a quarter of the CUs, each running two-round passes (1500 / 1148 work items on
// 1024 threads); 256 tiles of 32^2 carry 2.1x the halo work in total but 0.51x per workgroup, on every CU
template <typename T, int F, int NL, int OT = 64>
__global__ void __launch_bounds__((F > 10) ? 512 : 1024) k_fwd2d_tile(TileArgs<T, F> a)
{
    typedef TileLds<F, NL, OT> L;
    typedef T F4t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(32))) unsigned char smem_raw[];
    T *S = reinterpret_cast<T *>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int r0 = blockIdx.x * OT, c0 = blockIdx.y * OT;
    constexpr int ld0 = L::ldx(L::R0), ld1 = L::ldx(L::R1), ld2 = L::ldx(L::R2);
    [[maybe_unused]] const int wgid = (int)(blockIdx.x + gridDim.x * blockIdx.y);
    if (tid == 0) WL_STAMP_AT(tile, wgid, 0);
    // ---- stage X0[i + c*ld0] = src[(r0 + i) mod M, (c0 + c) mod N], 16-byte loads along dim 1 ----
    {
        constexpr int RQ0 = (L::R0 + 3) / 4;
        for (int it = tid; it < RQ0 * L::C0; it += nthr) {
            const int iq = it % RQ0, c = it / RQ0;
            int gr = r0 + 4 * iq, gc = c0 + c;
            if (gr >= a.M) gr -= a.M;
            if (gc >= a.N) gc -= a.N;
            *reinterpret_cast<F4t *>(S + L::X0 + 4 * iq + c * ld0) = *reinterpret_cast<const F4t *>(a.src + gr + (int64_t)gc * a.lds);
        }
    }
    lds_barrier_vm();
    if (tid == 0) WL_STAMP_AT(tile, wgid, 1);
    const int hm = a.M >> 1, hn = a.N >> 1;
    if constexpr (NL == 1) {
        tile_level<T, F, L::R0, L::C0, OT / 2, OT / 2, OT / 2, true>(S + L::X0, ld0, S + L::T, ld0, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 1,
                                                      c0 >> 1, hm, hn, tid, nthr);
    } else {
        tile_level<T, F, L::R0, L::C0, L::R1, L::C1, OT / 2, false>(S + L::X0, ld0, S + L::T, ld0, S + L::X1, ld1, a.tp, a.y, a.ldy, a.ll, a.ldll,
                                                             r0 >> 1, c0 >> 1, hm, hn, tid, nthr);
        if (tid == 0) WL_STAMP_AT(tile, wgid, 2);
        if constexpr (NL == 2) {
            tile_level<T, F, L::R1, L::C1, OT / 4, OT / 4, OT / 4, true>(S + L::X1, ld1, S + L::T, ld1, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 2,
                                                          c0 >> 2, hm >> 1, hn >> 1, tid, nthr);
        } else {
            tile_level<T, F, L::R1, L::C1, L::R2, L::C2, OT / 4, false>(S + L::X1, ld1, S + L::T, ld1, S + L::X2, ld2, a.tp, a.y, a.ldy, a.ll,
                                                                 a.ldll, r0 >> 2, c0 >> 2, hm >> 1, hn >> 1, tid, nthr);
            if (tid == 0) WL_STAMP_AT(tile, wgid, 3);
            tile_level<T, F, L::R2, L::C2, OT / 8, OT / 8, OT / 8, true>(S + L::X2, ld2, S + L::T, ld2, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 3,
                                                       c0 >> 3, hm >> 2, hn >> 2, tid, nthr);
        }
    }
    if (tid == 0) WL_STAMP_AT(tile, wgid, 4);
}

// ---------------------------------------------------------------------------------------------------
// k_fwd2d_tileB<F>: TWO fused levels of a block that is too big for one resident round of the kernel above (2048^2: 1024 tiles).
// Same tiles, same one-sided halos, same arithmetic -- but the input tile is NOT staged: the dim-2 pass of the first level
// reads its window columns straight from global memory (the overlapping windows of neighbouring threads are L1 / L2 hits), so a
// workgroup holds only the dim-2 results (26 KB) and the first approximation (7 KB) in LDS: 33 KB and 256 threads instead of
// 62 KB and 1024.  Four workgroups per CU stay resident -- all 1024 tiles of a 2048^2 block at once -- where the staging
// kernel kept two, each waiting on its own loads, barriers and partially idle passes (PMC, 2048^2: 64 % of the wave cycles
// waiting, 41 % of the LDS cycles bank conflicts).
// (The body below is repeated as tileB_body in wl_tile_dev.h for the fused pair + tile launch of wl_pair2d.hip.  Routing THIS kernel
//  through that function -- the taps reached through a reference to a reference -- made hipcc park them in scratch (32 B per lane,
//  94 VGPRs instead of 70); the two copies are pinned to each other by tests/test_gpu_parity.py::test_fused_pair_tile_launch.)
#ifndef WL_TILEB_REMAP
#define WL_TILEB_REMAP 1      // (r06: 2048^2 two levels 14.6 -> 13.7 us)
#endif
template <int F>
__global__ void __launch_bounds__(256, 4) k_fwd2d_tileB(TileArgs<float, F> a)
{
    typedef float T;
    typedef TileLds<F, 2> L;
    typedef T F4t __attribute__((ext_vector_type(4)));
    constexpr int ldT = L::ldx(L::R0), ld1 = L::ldx(L::R1);
    constexpr int R0 = L::R0, R1 = L::R1, C1 = L::C1;
    constexpr int RQ = (R0 + 3) / 4;
    __shared__ __attribute__((aligned(16))) T Ts[ldT * (C1 + 32)];
    __shared__ __attribute__((aligned(16))) T X1s[ld1 * C1 + 16];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int wgid = (int)(blockIdx.x + gridDim.x * blockIdx.y);
    // XCD-aware tile map (workgroup w runs on XCD w % 8): each XCD owns a compact gx/2 x gy/4 block of tiles, so that the halo rows /
    // columns a tile shares with its neighbours are hits in the XCD's own L2 instead of a second fetch over the fabric
    int tbx = (int)blockIdx.x, tby = (int)blockIdx.y;
#if WL_TILEB_REMAP
    if ((gridDim.x & 1) == 0 && (gridDim.y & 3) == 0) {
        const int xcd = wgid & 7, i = wgid >> 3, rx = (int)gridDim.x >> 1;
        tbx = (xcd & 1) * rx + i % rx;
        tby = (xcd >> 1) * ((int)gridDim.y >> 2) + i / rx;
    }
#endif
    const int r0 = tbx * 64, c0 = tby * 64;
    const int hm = a.M >> 1, hn = a.N >> 1;
    if (tid == 0) WL_STAMP_AT(tileB, wgid, 0);
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
                xw[m] = load_pol<WL_P_TILE_LD != 0>(reinterpret_cast<const F4t *>(a.src + gr + (int64_t)c * a.lds));
            }
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                const int k = k0 + kk;
                F4t sv = a.tp.h[0] * xw[2 * kk], dv = a.tp.g[F - 1] * xw[2 * kk];
#pragma unroll
                for (int m = 1; m < F; ++m) {
                    sv = sv + a.tp.h[m] * xw[2 * kk + m];
                    dv = dv + a.tp.g[F - 1 - m] * xw[2 * kk + m];
                }
                if (k < C1) *reinterpret_cast<F4t *>(Ts + 4 * iq + k * ldT) = sv;
                if (k < 32) *reinterpret_cast<F4t *>(Ts + 4 * iq + (C1 + k) * ldT) = dv;
            }
        }
    }
    if (tid == 0) WL_STAMP_AT(tileB, wgid, 1);
    lds_barrier();
    if (tid == 0) WL_STAMP_AT(tileB, wgid, 2);
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
                T sv = a.tp.h[0] * E[2 * j];
#pragma unroll
                for (int m = 1; m < F; ++m) sv = sv + a.tp.h[m] * E[2 * j + m];
                T dv = a.tp.g[F - 1] * E[2 * j + 10 - F];
#pragma unroll
                for (int m = F - 2; m >= 0; --m) dv = dv + a.tp.g[m] * E[2 * j + 9 - m];
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
    if (tid == 0) WL_STAMP_AT(tileB, wgid, 3);
    // ---- level 2: LDS -> LDS as in the staging kernel ----
    tile_level<T, F, L::R1, L::C1, 16, 16, 16, true>(X1s, ld1, Ts, ld1, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 2, c0 >> 2, hm >> 1,
                                                  hn >> 1, tid, nthr);
    if (tid == 0) WL_STAMP_AT(tileB, wgid, 4);
}

template <int F>
static hipError_t launch_tileB_f(hipStream_t st, const Taps<float> &taps, const float *src, int64_t lds, float *y, int64_t ldy, float *ll,
                                 int64_t ldll, int M, int N)
{
    TileArgs<float, F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.M = M; a.N = N;
    a.tp = shrink<float, F>(taps);
    hipLaunchKernelGGL((k_fwd2d_tileB<F>), dim3((unsigned)(M / 64), (unsigned)(N / 64)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t fwd2d_tileB_launch(hipStream_t st, const Taps<float> &taps, const float *src, int64_t lds, float *y, int64_t ldy, float *ll,
                              int64_t ldll, int M, int N)
{
    switch (taps.F) {
    case 2: return launch_tileB_f<2>(st, taps, src, lds, y, ldy, ll, ldll, M, N);
    case 4: return launch_tileB_f<4>(st, taps, src, lds, y, ldy, ll, ldll, M, N);
    case 6: return launch_tileB_f<6>(st, taps, src, lds, y, ldy, ll, ldll, M, N);
    case 8: return launch_tileB_f<8>(st, taps, src, lds, y, ldy, ll, ldll, M, N);
    case 10: return launch_tileB_f<10>(st, taps, src, lds, y, ldy, ll, ldll, M, N);
    default: return hipErrorInvalidValue;
    }
}

bool fwd2d_tile_ok(int F, int NL, int64_t M, int64_t N)
{
    if (F < 2 || F > 20 || (F & 1) || NL < 1 || NL > 3) return false;
    if (F > 10 && NL > 2) return false;            // (12..20 taps: Float32, at most two levels -- the staged tile grows with (F-2)(2^NL - 1))
    // tiles of 64 x 64; the d rows / columns wrap in groups of four at every level; the tile with halo must not wrap twice
    return M >= 128 && N >= 128 && (M % 64) == 0 && (N % 64) == 0 && M <= 4096 && N <= 4096 && (M >> NL) % 4 == 0 && (N >> NL) >= 1;
}

template <typename T, int F, int NL, int OT = 64>
static hipError_t launch_tile_fn(hipStream_t st, const TileArgs<T, F> &a)
{
    constexpr size_t shmem = (size_t)TileLds<F, NL, OT>::TOTAL * sizeof(T);
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd2d_tile<T, F, NL, OT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    unsigned nthr = (unsigned)opt("WL_TILE_THREADS", OT == 32 ? 512 : 1024);
    if (F > 10 && nthr > 512) nthr = 512;
    hipLaunchKernelGGL((k_fwd2d_tile<T, F, NL, OT>), dim3((unsigned)(a.M / OT), (unsigned)(a.N / OT)), dim3(nthr), shmem, st, a);
    return hipGetLastError();
}

template <typename T, int F>
static hipError_t launch_tile_f(hipStream_t st, const Taps<T> &taps, int NL, const T *src, int64_t lds, T *y, int64_t ldy,
                                T *ll, int64_t ldll, int M, int N)
{
    TileArgs<T, F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.M = M; a.N = N;
    a.tp = shrink<T, F>(taps);
    switch (NL) {
    case 1: return launch_tile_fn<T, F, 1>(st, a);
    case 2: return launch_tile_fn<T, F, 2>(st, a);
    default:
        if constexpr (sizeof(T) == 4 && F <= 10) {
            // three levels: tiles of 32 x 32 while 64 x 64 ones would leave CUs idle (a 512^2 block: 64 tiles) -- (M >> 3) % 4 == 0 holds
            // for them as for the 64-sample tiles (fwd2d_tile_ok)
            if ((int64_t)(M / 64) * (N / 64) < (int64_t)opt("WL_TILE32_BELOW", 128) && (M % 32) == 0 && (N % 32) == 0) return launch_tile_fn<T, F, 3, 32>(st, a);
            return launch_tile_fn<T, F, 3>(st, a);
        }
        else return hipErrorInvalidValue;          // (three levels of Float64 / of a long filter do not fit the 160 KiB of LDS)
    }
}

template <typename T>
hipError_t fwd2d_tile_launch(hipStream_t st, const Taps<T> &taps, int NL, const T *src, int64_t lds, T *y, int64_t ldy,
                             T *ll, int64_t ldll, int M, int N)
{
    switch (taps.F) {
    case 2: return launch_tile_f<T, 2>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 4: return launch_tile_f<T, 4>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 6: return launch_tile_f<T, 6>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 8: return launch_tile_f<T, 8>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 10: return launch_tile_f<T, 10>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    default: break;
    }
    if constexpr (sizeof(T) == 4) {                // 12..20 taps: Float32 only (wl_fwd.hip sends only those here)
        switch (taps.F) {
        case 12: return launch_tile_f<T, 12>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
        case 14: return launch_tile_f<T, 14>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
        case 16: return launch_tile_f<T, 16>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
        case 18: return launch_tile_f<T, 18>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
        case 20: return launch_tile_f<T, 20>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
        default: break;
        }
    }
    return hipErrorInvalidValue;
}
template hipError_t fwd2d_tile_launch<float>(hipStream_t, const Taps<float> &, int, const float *, int64_t, float *, int64_t, float *, int64_t, int,
                                             int);
template hipError_t fwd2d_tile_launch<double>(hipStream_t, const Taps<double> &, int, const double *, int64_t, double *, int64_t, double *, int64_t,
                                              int, int);

// ---------------------------------------------------------------------------------------------------
// The inverse of the same sizes: TWO reconstruction levels (output M/2 x N/2, then M x N, M, N in 128 .. 1024) per launch,
// one 64 x 64 piece of the final output per workgroup.  Everything a tile needs -- the (16 + SH2 + SH)^2 corner of the deeper
// approximation and the matching pieces of the six detail quadrants, one-sided halos: s coefficients reach SH pairs back,
// d coefficients SH pairs ahead -- is staged to LDS with all loads in flight, then four LDS -> LDS passes (dim 1 / dim 2 of
// the coarse level, dim 1 / dim 2 of the fine level; reference order transforms_filter.jl:173-186) produce the tile; halos
// are recomputed, workgroups are independent.  Replaces two ~7 us launches of k_inv2d_stream by one of ~5 us.
template <typename T, int F>
struct TileInvArgs {
    const T *x; int64_t ldx;            // coefficient array
    const T *ll; int64_t ldl;           // reconstruction of the deeper level (M/4 x N/4, dense)
    T *dst; int64_t ldd;                // M x N result
    int M, N;
    TapsF<T, F> tp;
};

template <typename T, int F>
__global__ void __launch_bounds__(1024) k_inv2d_tile2(TileInvArgs<T, F> a)
{
    constexpr int SH = (F - 2) / 2, SH2 = (SH + 1) / 2, OFF1 = 2 * SH2 - SH;
    constexpr int NP1 = 32, NP2 = 16 + SH2;                 // output pairs per dimension: fine level, coarse level
    constexpr int E1 = NP1 + SH, E2 = NP2 + SH;             // staged s / d extents per dimension
    constexpr int LD2 = E2 | 1, LT2 = (2 * NP2) | 1, LD1 = E1 | 1, LT1 = 65;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *SS2 = reinterpret_cast<T *>(smem_raw);               // [E2 x E2] s rows, s cols (deeper approximation)
    T *DS2 = SS2 + LD2 * E2, *SD2 = DS2 + LD2 * E2, *DD2 = SD2 + LD2 * E2;
    T *T2 = DD2 + LD2 * E2;                                 // [2 NP2 rows x (E2 s-cols | E2 d-cols)]
    T *A1 = T2 + LT2 * 2 * E2;                              // [2 NP2 x 2 NP2] coarse-level output around the tile
    T *DS1 = A1 + LT2 * 2 * NP2, *SD1 = DS1 + LD1 * E1, *DD1 = SD1 + LD1 * E1;
    T *T1 = DD1 + LD1 * E1;                                 // [64 rows x (E1 s-cols | E1 d-cols)]
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int h0 = a.M >> 1, h1 = a.N >> 1, q0 = a.M >> 2, q1 = a.N >> 2;       // half extents: fine level, coarse level
    const int P1r = r0 >> 1, P1c = c0 >> 1, P2r = (P1r >> 1) - SH2, P2c = (P1c >> 1) - SH2;
    auto wrap = [](int i, int n) __attribute__((always_inline)) { if (i < 0) i += n; if (i >= n) i -= n; return i; };
    // ---- stage (scalar loads, lanes along the rows) ----
    for (int it = tid; it < E2 * E2; it += nthr) {
        const int ai = it % E2, bi = it / E2;
        const int rs = wrap(P2r - SH + ai, q0), rd = wrap(P2r + ai, q0), cs = wrap(P2c - SH + bi, q1), cd = wrap(P2c + bi, q1);
        SS2[ai + bi * LD2] = a.ll[rs + (int64_t)cs * a.ldl];
        DS2[ai + bi * LD2] = a.x[q0 + rd + (int64_t)cs * a.ldx];
        SD2[ai + bi * LD2] = a.x[rs + (int64_t)(q1 + cd) * a.ldx];
        DD2[ai + bi * LD2] = a.x[q0 + rd + (int64_t)(q1 + cd) * a.ldx];
    }
    for (int it = tid; it < E1 * E1; it += nthr) {
        const int ai = it % E1, bi = it / E1;
        const int rs = wrap(P1r - SH + ai, h0), rd = wrap(P1r + ai, h0), cs = wrap(P1c - SH + bi, h1), cd = wrap(P1c + bi, h1);
        DS1[ai + bi * LD1] = a.x[h0 + rd + (int64_t)cs * a.ldx];
        SD1[ai + bi * LD1] = a.x[rs + (int64_t)(h1 + cd) * a.ldx];
        DD1[ai + bi * LD1] = a.x[h0 + rd + (int64_t)(h1 + cd) * a.ldx];
    }
    lds_barrier_vm();
    // ---- coarse level, dim 1: columns b of (s-cols | d-cols), pairs p -> T2 rows 2p, 2p+1 ----
    for (int it = tid; it < NP2 * 2 * E2; it += nthr) {
        const int p = it % NP2, b = it / NP2;
        const T *sp = (b < E2) ? SS2 + b * LD2 : SD2 + (b - E2) * LD2;
        const T *dp = (b < E2) ? DS2 + b * LD2 : DD2 + (b - E2) * LD2;
        T sw[SH + 1], dw[SH + 1];
#pragma unroll
        for (int q = 0; q <= SH; ++q) { sw[q] = sp[p + q]; dw[q] = dp[p + q]; }
        T xe, xo;
        window_inv<T, F>(sw, dw, a.tp, xe, xo);
        T2[2 * p + b * LT2] = xe;
        T2[2 * p + 1 + b * LT2] = xo;
    }
    lds_barrier();
    // ---- coarse level, dim 2: rows i, column pairs p -> A1 ----
    for (int it = tid; it < 2 * NP2 * NP2; it += nthr) {
        const int i = it % (2 * NP2), p = it / (2 * NP2);
        T sw[SH + 1], dw[SH + 1];
#pragma unroll
        for (int q = 0; q <= SH; ++q) { sw[q] = T2[i + (p + q) * LT2]; dw[q] = T2[i + (E2 + p + q) * LT2]; }
        T xe, xo;
        window_inv<T, F>(sw, dw, a.tp, xe, xo);
        A1[i + (2 * p) * LT2] = xe;
        A1[i + (2 * p + 1) * LT2] = xo;
    }
    lds_barrier();
    // ---- fine level, dim 1: s rows come from A1 (offset OFF1: its region starts at an even row) ----
    for (int it = tid; it < NP1 * 2 * E1; it += nthr) {
        const int p = it % NP1, b = it / NP1;
        T sw[SH + 1], dw[SH + 1];
        if (b < E1) {
#pragma unroll
            for (int q = 0; q <= SH; ++q) { sw[q] = A1[OFF1 + p + q + (OFF1 + b) * LT2]; dw[q] = DS1[p + q + b * LD1]; }
        } else {
#pragma unroll
            for (int q = 0; q <= SH; ++q) { sw[q] = SD1[p + q + (b - E1) * LD1]; dw[q] = DD1[p + q + (b - E1) * LD1]; }
        }
        T xe, xo;
        window_inv<T, F>(sw, dw, a.tp, xe, xo);
        T1[2 * p + b * LT1] = xe;
        T1[2 * p + 1 + b * LT1] = xo;
    }
    lds_barrier();
    // ---- fine level, dim 2: rows i, column pairs p -> the result tile ----
    T *out = a.dst + r0 + (int64_t)c0 * a.ldd;
    for (int it = tid; it < 64 * NP1; it += nthr) {
        const int i = it & 63, p = it >> 6;
        T sw[SH + 1], dw[SH + 1];
#pragma unroll
        for (int q = 0; q <= SH; ++q) { sw[q] = T1[i + (p + q) * LT1]; dw[q] = T1[i + (E1 + p + q) * LT1]; }
        T xe, xo;
        window_inv<T, F>(sw, dw, a.tp, xe, xo);
        out[i + (int64_t)(2 * p) * a.ldd] = xe;
        out[i + (int64_t)(2 * p + 1) * a.ldd] = xo;
    }
}

template <typename T>
bool inv2d_tile2_ok(int F, int64_t M, int64_t N)
{
    if (F < 2 || F > 20 || (F & 1)) return false;      // (round 5: 12..20 taps too -- the staged pieces grow with (F-2)/2, 66 KB at 16 taps Float32)
    return M >= 128 && N >= 128 && M <= 1024 && N <= 1024 && (M % 64) == 0 && (N % 64) == 0;
}

template <typename T, int F>
static hipError_t launch_inv_tile2_f(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl, T *dst,
                                     int64_t ldd, int M, int N)
{
    constexpr int SH = (F - 2) / 2, SH2 = (SH + 1) / 2, NP1 = 32, NP2 = 16 + SH2, E1 = NP1 + SH, E2 = NP2 + SH;
    constexpr int LD2 = E2 | 1, LT2 = (2 * NP2) | 1, LD1 = E1 | 1, LT1 = 65;
    constexpr size_t elems = 4 * LD2 * E2 + LT2 * 2 * E2 + LT2 * 2 * NP2 + 3 * LD1 * E1 + LT1 * 2 * E1 + 16;
    TileInvArgs<T, F> a;
    a.x = x; a.ldx = ldx; a.ll = ll; a.ldl = ldl; a.dst = dst; a.ldd = ldd; a.M = M; a.N = N;
    a.tp = shrink<T, F>(taps);
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_inv2d_tile2<T, F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    hipLaunchKernelGGL((k_inv2d_tile2<T, F>), dim3((unsigned)(M / 64), (unsigned)(N / 64)), dim3((unsigned)opt("WL_TILE_THREADS", 1024)),
                       elems * sizeof(T), st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t inv2d_tile2_launch(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl, T *dst, int64_t ldd,
                              int M, int N)
{
    switch (taps.F) {
    case 2: return launch_inv_tile2_f<T, 2>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 4: return launch_inv_tile2_f<T, 4>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 6: return launch_inv_tile2_f<T, 6>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 8: return launch_inv_tile2_f<T, 8>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 10: return launch_inv_tile2_f<T, 10>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 12: return launch_inv_tile2_f<T, 12>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 14: return launch_inv_tile2_f<T, 14>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 16: return launch_inv_tile2_f<T, 16>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 18: return launch_inv_tile2_f<T, 18>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    case 20: return launch_inv_tile2_f<T, 20>(st, taps, x, ldx, ll, ldl, dst, ldd, M, N);
    default: return hipErrorInvalidValue;
    }
}
template bool inv2d_tile2_ok<float>(int, int64_t, int64_t);
template bool inv2d_tile2_ok<double>(int, int64_t, int64_t);
template hipError_t inv2d_tile2_launch<float>(hipStream_t, const Taps<float> &, const float *, int64_t, const float *, int64_t, float *, int64_t,
                                              int, int);
template hipError_t inv2d_tile2_launch<double>(hipStream_t, const Taps<double> &, const double *, int64_t, const double *, int64_t, double *,
                                               int64_t, int, int);

}  // namespace wl
