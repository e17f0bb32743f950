// wl_level3.hip -- ONE launch per level for the small 3-D filter-bank levels (boxes of 32^3 ... 64^3 and their non-cubic relatives),
// forward and inverse, both element types, even F <= 10.
//
//   k_level3_lds<T, F, P, FW>     reference: the planes / rows / columns passes of one 3-D level, transforms_filter.jl:246-287
//
// Between the streaming sizes (k_fwd3d_one, the plane kernels) and the one-workgroup tail (k_tail3: <= 4096 elements) a level took
// three dependent single-axis launches (wl_axis.hip) on a box that lives in L2: 4-5 us each, almost all of it launch latency.  Here a
// workgroup owns a block of P x P x P coefficient pairs, stages the part of the box its windows cover in LDS -- (2 P + F - 2)^3
// samples forward (one-sided windows: s[p] and d[p + SH] share x[2p .. 2p + F - 1]), (2 (P + SH))^3 coefficients inverse (s[p - SH .. p]
// and d[p .. p + SH] per axis) -- runs the three passes LDS -> LDS in the reference's order (forward: dim 3, dim 2, dim 1; inverse:
// dim 1, dim 2, dim 3) and writes its share of the level.  The halo is recomputed per workgroup ((22 / 16)^3 = 2.6 x the arithmetic
// for P = 8, 8 taps): the data is cache-resident and the launch count is what these levels cost.
// Arithmetic: the closed forms of wl_internal.h in the reference's summation order -- bit-identical to the axis kernels.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <typename T, int F>
struct Level3Args {
    const T *src; int64_t s1, s2;      // forward: the level-l box;  inverse: the coefficient array (full strides)
    T *dst; int64_t d1, d2;            // forward: the coefficient array (full strides);  inverse: the reconstructed box
    const T *llr; T *llw;              // approximation octant, dense (h0, h1, h2): forward writes llw (or dst when null), inverse reads llr (or src)
    int n0, n1, n2;                    // extents of the level (inverse: of its OUTPUT)
    int nb0, nb1;                      // blocks along dim 1 / dim 2 (blockIdx.x = b0 + nb0 * (b1 + nb1 * b2))
    TapsF<T, F> tp;
};

template <typename T, int F, int P, int FW>
__global__ void __launch_bounds__(256) k_level3_lds(Level3Args<T, F> a)
{
    constexpr int SH = (F - 2) / 2;
    constexpr int E = FW ? (2 * P + F - 2) : 2 * (P + SH);        // staged extent per axis
    constexpr int Q = 2 * P;                                      // produced extent per axis
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *const A = reinterpret_cast<T *>(smem_raw);                 // [E][E][E], later [E | Q][Q][Q] ...
    T *const B = A + E * E * E;                                   // [.][E][E] intermediate
    const int tid = threadIdx.x, nthr = blockDim.x;
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };

    const int b = blockIdx.x;
    const int b0 = b % a.nb0, b1 = (b / a.nb0) % a.nb1, b2 = b / (a.nb0 * a.nb1);
    const int n0 = a.n0, n1 = a.n1, n2 = a.n2, h0 = n0 >> 1, h1 = n1 >> 1, h2 = n2 >> 1;
    // first pair of the block per axis; extents that are not multiples of 2 P: the last block is moved back to end at the edge and
    // recomputes what it shares with its neighbour (same values, same addresses)
    const int pa0 = (P * b0 + P <= h0) ? P * b0 : h0 - P, pa1 = (P * b1 + P <= h1) ? P * b1 : h1 - P, pa2 = (P * b2 + P <= h2) ? P * b2 : h2 - P;

    if constexpr (FW) {
        // ---- stage x[2 pa + e] (periodic) ----
        for (int idx = tid; idx < E * E * E; idx += nthr) {
            const int i = idx % E, j = (idx / E) % E, k = idx / (E * E);
            int gi = 2 * pa0 + i, gj = 2 * pa1 + j, gk = 2 * pa2 + k;
            if (gi >= n0) gi -= n0;
            if (gj >= n1) gj -= n1;
            if (gk >= n2) gk -= n2;
            A[idx] = a.src[gi + (int64_t)gj * a.s1 + (int64_t)gk * a.s2];
        }
        lds_barrier_vm();
        // one-sided window at w[0 .. F-1] (stride st): s = sum h[m] w[m], d (index + SH) = sum g[F-1-m] w[m], m ascending
        auto win = [&](const T *w, const int st, T &s, T &d) __attribute__((always_inline)) {
            T x = w[0];
            s = a.tp.h[0] * x;
            d = gq(F - 1) * x;
#pragma unroll
            for (int m = 1; m < F; ++m) {
                x = w[m * st];
                s = s + a.tp.h[m] * x;
                d = d + gq(F - 1 - m) * x;
            }
        };
        // dim 3: A[E][E][E] -> B[Q][E][E]   (plane p: scaling, plane P + p: detail)
        for (int idx = tid; idx < E * E * P; idx += nthr) {
            const int ij = idx % (E * E), p = idx / (E * E);
            T s, d;
            win(A + ij + (2 * p) * E * E, E * E, s, d);
            B[ij + p * E * E] = s;
            B[ij + (P + p) * E * E] = d;
        }
        lds_barrier();
        // dim 2: B[Q][E][E] -> A[Q][Q][E]
        for (int idx = tid; idx < E * P * Q; idx += nthr) {
            const int i = idx % E, p = (idx / E) % P, k = idx / (E * P);
            T s, d;
            win(B + i + (2 * p) * E + k * E * E, E, s, d);
            A[i + p * E + k * Q * E] = s;
            A[i + (P + p) * E + k * Q * E] = d;
        }
        lds_barrier();
        // dim 1: A[Q][Q][E] -> the coefficient array; the all-scaling octant to llw (the next level's input) when given
        for (int idx = tid; idx < P * Q * Q; idx += nthr) {
            const int p = idx % P, j = (idx / P) % Q, k = idx / (P * Q);
            T s, d;
            win(A + 2 * p + j * E + k * Q * E, 1, s, d);
            int oj, ok, pd = pa0 + p + SH;
            if (pd >= h0) pd -= h0;
            if (j < P) oj = pa1 + j; else { oj = pa1 + (j - P) + SH; if (oj >= h1) oj -= h1; oj += h1; }
            if (k < P) ok = pa2 + k; else { ok = pa2 + (k - P) + SH; if (ok >= h2) ok -= h2; ok += h2; }
            T *const yc = a.dst + (int64_t)oj * a.d1 + (int64_t)ok * a.d2;
            yc[h0 + pd] = d;
            if (a.llw != nullptr && j < P && k < P) a.llw[(pa0 + p) + (int64_t)oj * h0 + (int64_t)ok * h0 * h1] = s;
            else yc[pa0 + p] = s;
        }
    } else {
        constexpr int H = P + SH;                                  // staged scaling (and detail) coefficients per axis
        // ---- stage: per axis entries [0, H) = s[pa - SH + e], [H, 2H) = d[pa + e - H] (periodic) ----
        for (int idx = tid; idx < E * E * E; idx += nthr) {
            const int i = idx % E, j = (idx / E) % E, k = idx / (E * E);
            int gi, gj, gk;
            bool lo = true;
            if (i < H) { gi = pa0 - SH + i; if (gi < 0) gi += h0; } else { gi = pa0 + i - H; if (gi >= h0) gi -= h0; gi += h0; lo = false; }
            if (j < H) { gj = pa1 - SH + j; if (gj < 0) gj += h1; } else { gj = pa1 + j - H; if (gj >= h1) gj -= h1; gj += h1; lo = false; }
            if (k < H) { gk = pa2 - SH + k; if (gk < 0) gk += h2; } else { gk = pa2 + k - H; if (gk >= h2) gk -= h2; gk += h2; lo = false; }
            A[idx] = (lo && a.llr != nullptr) ? a.llr[gi + (int64_t)gj * h0 + (int64_t)gk * h0 * h1]
                                              : a.src[gi + (int64_t)gj * a.s1 + (int64_t)gk * a.s2];
        }
        lds_barrier_vm();
        auto inv = [&](const T *w, const int st, T &xe, T &xo) __attribute__((always_inline)) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int q = 0; q <= SH; ++q) { sw[q] = w[q * st]; dw[q] = w[(H + q) * st]; }
            window_inv<T, F>(sw, dw, a.tp, xe, xo);
        };
        // dim 1: A[E][E][E] -> B[E][E][Q]
        for (int idx = tid; idx < P * E * E; idx += nthr) {
            const int p = idx % P, jk = idx / P;
            T xe, xo;
            inv(A + p + jk * E, 1, xe, xo);
            B[2 * p + jk * Q] = xe;
            B[2 * p + 1 + jk * Q] = xo;
        }
        lds_barrier();
        // dim 2: B[E][E][Q] -> A[E][Q][Q]
        for (int idx = tid; idx < Q * P * E; idx += nthr) {
            const int i = idx % Q, p = (idx / Q) % P, k = idx / (Q * P);
            T xe, xo;
            inv(B + i + p * Q + k * E * Q, Q, xe, xo);
            A[i + (2 * p) * Q + k * Q * Q] = xe;
            A[i + (2 * p + 1) * Q + k * Q * Q] = xo;
        }
        lds_barrier();
        // dim 3: A[E][Q][Q] -> the reconstructed box
        for (int idx = tid; idx < Q * Q * P; idx += nthr) {
            const int i = idx % Q, j = (idx / Q) % Q, p = idx / (Q * Q);
            T xe, xo;
            inv(A + i + j * Q + p * Q * Q, Q * Q, xe, xo);
            T *const o = a.dst + (2 * pa0 + i) + (int64_t)(2 * pa1 + j) * a.d1 + (int64_t)(2 * (pa2 + p)) * a.d2;
            o[0] = xe;
            o[a.d2] = xo;
        }
    }
}

// any_tier: the box is not a shape of the axis / plane kernels (extents that are not powers of two: the any-extent kernels of
// wl_anyaxis.hip would take it, three launches per level) -- there the blocks stay ahead up to about 2^20 elements (96^3 dwt L = 2
// 32.7 -> 22.5 us, idwt 30.4 -> 24.3; 200^3 and 240 x 240 x 160 lose: 80 -> 95, 107 -> 125)
template <typename T>
bool level3_lds_ok(int F, const int64_t n[3], bool any_tier)
{
    if (opt("WL_LEVEL3", 1) == 0) return false;
    if (F < 2 || F > 10 || (F & 1)) return false;
    for (int a = 0; a < 3; ++a)
        if (n[a] < 16 || (n[a] % 2) != 0 || n[a] > 4096) return false;
    const int64_t tot = n[0] * n[1] * n[2];
    return tot > 4096 && tot <= (any_tier ? opt("WL_LEVEL3_MAX_ANY", (long long)1 << 20) : opt("WL_LEVEL3_MAX", (long long)1 << 18));
}
template bool level3_lds_ok<float>(int, const int64_t[3], bool);
template bool level3_lds_ok<double>(int, const int64_t[3], bool);

template <typename T, int F, int P, int FW>
static hipError_t launch_level3_inst(hipStream_t st, const Level3Args<T, F> &a0, const int64_t n[3])
{
    constexpr int SH = (F - 2) / 2, E = FW ? (2 * P + F - 2) : 2 * (P + SH), Q = 2 * P;
    Level3Args<T, F> a = a0;
    a.nb0 = (int)((n[0] / 2 + P - 1) / P); a.nb1 = (int)((n[1] / 2 + P - 1) / P);
    const unsigned nwg = (unsigned)(a.nb0 * a.nb1 * ((n[2] / 2 + P - 1) / P));
    const size_t shmem = (size_t)(E * E * E + Q * E * E) * sizeof(T);
    static thread_local int attr_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool done = false;
    for (int i = 0; i < 8; ++i) done = done || attr_dev[i] == dev;
    if (!done && shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_level3_lds<T, F, P, FW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        for (int i = 0; i < 8; ++i) if (attr_dev[i] < 0) { attr_dev[i] = dev; break; }
    }
    hipLaunchKernelGGL((k_level3_lds<T, F, P, FW>), dim3(nwg), dim3(256), shmem, st, a);
    return hipGetLastError();
}

template <typename T, int F, int FW>
static hipError_t launch_level3_f(hipStream_t st, const Taps<T> &taps, const T *src, int64_t s1, int64_t s2, T *dst, int64_t d1, int64_t d2,
                                  const T *llr, T *llw, const int64_t n[3])
{
    Level3Args<T, F> a;
    a.src = src; a.s1 = s1; a.s2 = s2; a.dst = dst; a.d1 = d1; a.d2 = d2; a.llr = llr; a.llw = llw;
    a.n0 = (int)n[0]; a.n1 = (int)n[1]; a.n2 = (int)n[2]; a.nb0 = a.nb1 = 0;
    a.tp = shrink<T, F>(taps);
    // blocks of 4^3 pairs (measured: 64^3 full depth 26.6 us against 36.2 with 8^3 blocks and 37.4 with three launches per level: the
    // smaller block is the shorter dependent chain); 8^3 on request
    const bool p8 = (n[0] % 16) == 0 && (n[1] % 16) == 0 && (n[2] % 16) == 0 && (n[0] / 16) * (n[1] / 16) * (n[2] / 16) >= opt("WL_LEVEL3_P8_MIN", (long long)1 << 30);
    constexpr int E8 = FW ? (16 + F - 2) : 2 * (8 + (F - 2) / 2);
    constexpr bool fits8 = (size_t)(E8 * E8 * E8 + 16 * E8 * E8) * sizeof(T) <= 160 * 1024;      // (Float64, 10 taps: 184 KB)
    if constexpr (fits8) {
        if (p8) return launch_level3_inst<T, F, 8, FW>(st, a, n);
    }
    return launch_level3_inst<T, F, 4, FW>(st, a, n);
}

template <typename T>
hipError_t level3_lds_launch(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t s1, int64_t s2, T *dst, int64_t d1, int64_t d2,
                             const T *llr, T *llw, const int64_t n[3])
{
#define WL_L3F(F_) case F_: return fw ? launch_level3_f<T, F_, 1>(st, taps, src, s1, s2, dst, d1, d2, llr, llw, n) \
                                      : launch_level3_f<T, F_, 0>(st, taps, src, s1, s2, dst, d1, d2, llr, llw, n)
    switch (taps.F) {
    WL_L3F(2); WL_L3F(4); WL_L3F(6); WL_L3F(8); WL_L3F(10);
    default: return hipErrorInvalidValue;
    }
#undef WL_L3F
}
template hipError_t level3_lds_launch<float>(hipStream_t, const Taps<float> &, int, const float *, int64_t, int64_t, float *, int64_t, int64_t,
                                             const float *, float *, const int64_t[3]);
template hipError_t level3_lds_launch<double>(hipStream_t, const Taps<double> &, int, const double *, int64_t, int64_t, double *, int64_t, int64_t,
                                              const double *, double *, const int64_t[3]);

}  // namespace wl
