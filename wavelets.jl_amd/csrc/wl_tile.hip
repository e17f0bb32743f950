// wl_tile.hip -- forward 2-D filter-bank levels of the cache-resident blocks (128^2 .. 2048^2, Float32): NL = 1..3 fused
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

namespace wl {

template <int F>
struct TileArgs {
    const float *src; int64_t lds;      // input block M x N
    float *y; int64_t ldy;
    float *ll; int64_t ldll;            // approximation after NL levels (dense buffer or y itself)
    int M, N;
    TapsF<float, F> tp;
};

// extents of the tile at level l (l = 0: the launch's input), OT = 64 owned input samples per side
template <int F, int NL, int l>
struct TileDim {
    static constexpr int HR = 8, HC = F - 2;                       // one-sided halos per level: rows (d rows shifted by 4), columns
    static constexpr int R = 2 * TileDim<F, NL, l + 1>::R + HR;    // rows / columns of this level's input that the tile needs
    static constexpr int C = 2 * TileDim<F, NL, l + 1>::C + HC;
};
template <int F, int NL>
struct TileDim<F, NL, NL> {
    static constexpr int R = 64 >> NL, C = 64 >> NL;
};

template <int F, int NL>
struct TileLds {
    // X0 | T | X1 | X2 (float offsets); leading dimensions padded to a multiple of 4 rows plus 4 (bank spread, 16-byte aligned)
    static constexpr int ldx(int r) { return ((r + 3) & ~3) + 4; }
    static constexpr int R0 = TileDim<F, NL, 0>::R, C0 = TileDim<F, NL, 0>::C;
    static constexpr int R1 = TileDim<F, NL, (NL >= 1 ? 1 : 0)>::R, C1 = TileDim<F, NL, (NL >= 1 ? 1 : 0)>::C;
    static constexpr int R2 = TileDim<F, NL, (NL >= 2 ? 2 : NL)>::R, C2 = TileDim<F, NL, (NL >= 2 ? 2 : NL)>::C;
    static constexpr int X0 = 0;
    static constexpr int T = X0 + ldx(R0) * C0;
    static constexpr int X1 = T + ldx(R0) * (C1 + 32);             // T: R0 rows x (C1 s-columns + 32 owned d-columns)
    static constexpr int X2 = X1 + ldx(R1) * C1;
    static constexpr int TOTAL = X2 + ldx(R2) * C2 + 16;
};

typedef float F4t __attribute__((ext_vector_type(4)));

// One level inside the tile.  X: input R x C (leading dimension ldx), T: scratch, XN: next level's input (RN x CN) in LDS.
// OWN = owned outputs per side at this level (32, 16, 8); (r0h, c0h) = tile origin in this level's OUTPUT coordinates;
// hm, hn = half extents of this level's block; LAST: the approximation goes to global memory (ll) instead of XN.
template <int F, int R, int C, int RN, int CN, int OWN, bool LAST>
__device__ __forceinline__ void tile_level(const float *X, int ldX, float *T, int ldT, float *XN, int ldN, const TapsF<float, F> &tp,
                                           float *y, int64_t ldy, float *ll, int64_t ldll, int r0h, int c0h, int hm, int hn, int tid,
                                           int nthr)
{
    constexpr int SH = (F - 2) / 2;
    constexpr int RQ = (R + 3) / 4;                 // row quads of the input
    // ---- dim-2 pass: X (R x C) -> T: columns [0, CN) = s (window columns 2k .. 2k+F-1), columns [CN, CN+OWN) = d[k + SH]
    for (int it = tid; it < RQ * CN; it += nthr) {
        const int iq = it % RQ, k = it / RQ;
        const float *p = X + 4 * iq + (2 * k) * ldX;
        F4t x0 = *reinterpret_cast<const F4t *>(p);
        F4t s = tp.h[0] * x0, d = tp.g[F - 1] * x0;
#pragma unroll
        for (int m = 1; m < F; ++m) {
            const F4t xm = *reinterpret_cast<const F4t *>(p + m * ldX);
            s = s + tp.h[m] * xm;
            d = d + tp.g[F - 1 - m] * xm;
        }
        *reinterpret_cast<F4t *>(T + 4 * iq + k * ldT) = s;
        if (k < OWN) *reinterpret_cast<F4t *>(T + 4 * iq + (CN + k) * ldT) = d;
    }
    lds_barrier();
    // ---- dim-1 pass: column c of T, rows 8q .. 8q+15 -> s rows 4q .. 4q+3, d rows 4q+4 .. 4q+7
    constexpr int QG = (RN + 3) / 4;                // groups of four output rows (covers the RN approximation rows needed below)
    for (int it = tid; it < QG * (CN + OWN); it += nthr) {
        const int q = it % QG, c = it / QG;
        const float *p = T + 8 * q + c * ldT;
        float E[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const F4t t = *reinterpret_cast<const F4t *>(p + 4 * v);
            E[4 * v] = t.x; E[4 * v + 1] = t.y; E[4 * v + 2] = t.z; E[4 * v + 3] = t.w;
        }
        F4t so, dO;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = tp.h[0] * E[2 * j];
#pragma unroll
            for (int m = 1; m < F; ++m) s = s + tp.h[m] * E[2 * j + m];
            float d = tp.g[F - 1] * E[2 * j + 10 - F];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + tp.g[m] * E[2 * j + 9 - m];
            so[j] = s; dO[j] = d;
        }
        const bool is_s = c < CN;
        // approximation of an s-column: next level's input (all RN rows), or global when this is the launch's last level
        if (is_s) {
            if (!LAST) *reinterpret_cast<F4t *>(XN + 4 * q + c * ldN) = so;
            else if (c < OWN && 4 * q < OWN) *reinterpret_cast<F4t *>(ll + (r0h + 4 * q) + (int64_t)(c0h + c) * ldll) = so;
        }
        // details: owned rows / columns only
        const int cc = is_s ? c : c - CN;
        if (cc < OWN && 4 * q < OWN) {
            int64_t col;
            if (is_s) col = c0h + cc;                                   // s along dim 2
            else { int kd = c0h + cc + SH; if (kd >= hn) kd -= hn; col = hn + kd; }
            int rd = r0h + 4 * q + 4;
            if (rd >= hm) rd -= hm;
            float *yc = y + col * ldy;
            *reinterpret_cast<F4t *>(yc + hm + rd) = dO;                // ds or dd
            if (!is_s) *reinterpret_cast<F4t *>(yc + (r0h + 4 * q)) = so;  // sd
        }
    }
    lds_barrier();
}

template <int F, int NL>
__global__ void __launch_bounds__(1024) k_fwd2d_tile(TileArgs<F> a)
{
    typedef TileLds<F, NL> L;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *S = reinterpret_cast<float *>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    constexpr int ld0 = L::ldx(L::R0), ld1 = L::ldx(L::R1), ld2 = L::ldx(L::R2);
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
    const int hm = a.M >> 1, hn = a.N >> 1;
    if constexpr (NL == 1) {
        tile_level<F, L::R0, L::C0, 32, 32, 32, true>(S + L::X0, ld0, S + L::T, ld0, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 1,
                                                      c0 >> 1, hm, hn, tid, nthr);
    } else {
        tile_level<F, L::R0, L::C0, L::R1, L::C1, 32, false>(S + L::X0, ld0, S + L::T, ld0, S + L::X1, ld1, a.tp, a.y, a.ldy, a.ll, a.ldll,
                                                             r0 >> 1, c0 >> 1, hm, hn, tid, nthr);
        if constexpr (NL == 2) {
            tile_level<F, L::R1, L::C1, 16, 16, 16, true>(S + L::X1, ld1, S + L::T, ld1, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 2,
                                                          c0 >> 2, hm >> 1, hn >> 1, tid, nthr);
        } else {
            tile_level<F, L::R1, L::C1, L::R2, L::C2, 16, false>(S + L::X1, ld1, S + L::T, ld1, S + L::X2, ld2, a.tp, a.y, a.ldy, a.ll,
                                                                 a.ldll, r0 >> 2, c0 >> 2, hm >> 1, hn >> 1, tid, nthr);
            tile_level<F, L::R2, L::C2, 8, 8, 8, true>(S + L::X2, ld2, S + L::T, ld2, nullptr, 0, a.tp, a.y, a.ldy, a.ll, a.ldll, r0 >> 3,
                                                       c0 >> 3, hm >> 2, hn >> 2, tid, nthr);
        }
    }
}

bool fwd2d_tile_ok(int F, int NL, int64_t M, int64_t N)
{
    if (F < 2 || F > 10 || (F & 1) || NL < 1 || NL > 3) return false;
    // tiles of 64 x 64; the d rows / columns wrap in groups of four at every level; the tile with halo must not wrap twice
    return M >= 128 && N >= 128 && (M % 64) == 0 && (N % 64) == 0 && M <= 4096 && N <= 4096 && (M >> NL) % 4 == 0 && (N >> NL) >= 1;
}

template <int F, int NL>
static hipError_t launch_tile_fn(hipStream_t st, const TileArgs<F> &a)
{
    constexpr size_t shmem = (size_t)TileLds<F, NL>::TOTAL * sizeof(float);
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd2d_tile<F, NL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    hipLaunchKernelGGL((k_fwd2d_tile<F, NL>), dim3((unsigned)(a.M / 64), (unsigned)(a.N / 64)), dim3((unsigned)opt("WL_TILE_THREADS", 1024)), shmem, st, a);
    return hipGetLastError();
}

template <int F>
static hipError_t launch_tile_f(hipStream_t st, const Taps<float> &taps, int NL, const float *src, int64_t lds, float *y, int64_t ldy,
                                float *ll, int64_t ldll, int M, int N)
{
    TileArgs<F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.M = M; a.N = N;
    a.tp = shrink<float, F>(taps);
    switch (NL) {
    case 1: return launch_tile_fn<F, 1>(st, a);
    case 2: return launch_tile_fn<F, 2>(st, a);
    default: return launch_tile_fn<F, 3>(st, a);
    }
}

hipError_t fwd2d_tile_launch(hipStream_t st, const Taps<float> &taps, int NL, const float *src, int64_t lds, float *y, int64_t ldy,
                             float *ll, int64_t ldll, int M, int N)
{
    switch (taps.F) {
    case 2: return launch_tile_f<2>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 4: return launch_tile_f<4>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 6: return launch_tile_f<6>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 8: return launch_tile_f<8>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    case 10: return launch_tile_f<10>(st, taps, NL, src, lds, y, ldy, ll, ldll, M, N);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
