// wl_dev.h -- device-side helpers shared by the fast-path translation units (vector loads/stores, LDS-only barriers,
// DPP lane shifts, compile-time tap blocks).
#pragma once
#include "wl_internal.h"

namespace wl {

// ------------------------------------------------------------------------------------------
template <typename T, int F>
struct TapsF {
    T h[F];
    T g[F];
};
template <typename T, int F>
inline TapsF<T, F> shrink(const Taps<T> &t)
{
    TapsF<T, F> r;
    for (int i = 0; i < F; ++i) { r.h[i] = t.h[i]; r.g[i] = t.g[i]; }
    return r;
}

template <typename T, int N>
struct VecOf { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T>
struct VecOf<T, 1> { typedef T type; };

template <typename T, int N>
__device__ __forceinline__ void vload(const T *p, T (&v)[N])
{
    typedef typename VecOf<T, N>::type V;
    V t = *reinterpret_cast<const V *>(p);
    if constexpr (N == 1) v[0] = t;
    else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void vstore(T *p, const T (&v)[N])
{
    typedef typename VecOf<T, N>::type V;
    if constexpr (N == 1) *p = v[0];
    else {
        V t;
#pragma unroll
        for (int i = 0; i < N; ++i) t[i] = v[i];
        *reinterpret_cast<V *>(p) = t;
    }
}
// 16-byte-granular load/store of N elements (N*sizeof(T) may exceed 16 bytes)
template <typename T, int N>
__device__ __forceinline__ void vload16(const T *p, T (&v)[N])
{
    constexpr int C = 16 / sizeof(T);
    static_assert(N % C == 0, "chunking");
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        T t[C];
        vload<T, C>(p + c * C, t);
#pragma unroll
        for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void vstore16(T *p, const T (&v)[N])
{
    constexpr int C = 16 / sizeof(T);
    static_assert(N % C == 0, "chunking");
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        T t[C];
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        vstore<T, C>(p + c * C, t);
    }
}

// N elements in chunks of at most 16 bytes, with a cache policy (NT: non-temporal, see "cache policy" below)
template <bool NT, typename T, int N>
__device__ __forceinline__ void ldg_pol(const T *p, T (&v)[N])
{
    constexpr int C = ((int)(16 / sizeof(T)) < N) ? (int)(16 / sizeof(T)) : N;
    typedef typename VecOf<T, C>::type V;
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
        if constexpr (NT) t = __builtin_nontemporal_load(reinterpret_cast<const V *>(p + c * C));
        else t = *reinterpret_cast<const V *>(p + c * C);
        if constexpr (C == 1) v[c] = t;
        else {
#pragma unroll
            for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
        }
    }
}
template <bool NT, typename T, int N>
__device__ __forceinline__ void stg_pol(T *p, const T (&v)[N])
{
    constexpr int C = ((int)(16 / sizeof(T)) < N) ? (int)(16 / sizeof(T)) : N;
    typedef typename VecOf<T, C>::type V;
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
        if constexpr (C == 1) t = v[c];
        else {
#pragma unroll
            for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        }
        if constexpr (NT) __builtin_nontemporal_store(t, reinterpret_cast<V *>(p + c * C));
        else *reinterpret_cast<V *>(p + c * C) = t;
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also carries a workgroup-scope
// release fence, which on gfx9 waits for every outstanding GLOBAL store (vmcnt(0)); kernels that
// stream results to HBM between barriers and never read them back do not need that.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// ... and one that also drains this wave's global loads (after staging HBM data into LDS)
__device__ __forceinline__ void lds_barrier_vm() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Experiment builds only (tools/mkvariant.sh NAME "-DWL_WGTIME" file.hip): eight 100 MHz stamps per workgroup of the LAST launch of a
// kernel family, read back through wl_debug_wgtimes_<family> (tools/wlbench.cpp wgtime=FILE wgsym=<family>; tools/wgtime_stats.py).
#ifdef WL_WGTIME
#define WL_STAMP_DECL(NAME)                                                                                              \
    __device__ unsigned long long wl_dbg_##NAME[8 * 8192];                                                               \
    extern "C" __attribute__((visibility("default"))) int wl_debug_wgtimes_##NAME(unsigned long long *host, size_t n)   \
    {                                                                                                                    \
        return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(wl_dbg_##NAME), n * sizeof(unsigned long long));               \
    }
#define WL_STAMP_AT(NAME, wg, k) do { if ((threadIdx.x & 63) == 0 && (wg) < 8192) wl_dbg_##NAME[8 * (wg) + (k)] = wall_clock64(); } while (0)
#else
#define WL_STAMP_DECL(NAME)
#define WL_STAMP_AT(NAME, wg, k) do { } while (0)
#endif

constexpr __host__ __device__ int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

// Whole-wave lane shifts on the VALU (DPP wave_shl:1 / wave_shr:1 -- gfx9-family controls, valid
// on gfx950): lane i receives the value of lane i+1 (shl) or lane i-1 (shr).  No LDS round trip,
// unlike __shfl (ds_bpermute).  Lanes shifted in from outside the wave get an unspecified value;
// callers never store results that depend on them.
__device__ __forceinline__ int dpp_from_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_from_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ float from_next(float v) { return __int_as_float(dpp_from_next(__float_as_int(v))); }
__device__ __forceinline__ float from_prev(float v) { return __int_as_float(dpp_from_prev(__float_as_int(v))); }
__device__ __forceinline__ double from_next(double v)
{
    int lo = dpp_from_next(__double2loint(v)), hi = dpp_from_next(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_prev(double v)
{
    int lo = dpp_from_prev(__double2loint(v)), hi = dpp_from_prev(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// swap with the partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float from_partner(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ double from_partner(double v)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xf, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// (x[2p], x[2p+1]) from sw[q] = s[p - SH + q], dw[q] = d[p + q], q = 0..SH (wl_internal.h closed form, even F)
template <typename T, int F>
__device__ __forceinline__ void window_inv(const T (&sw)[(F - 2) / 2 + 1], const T (&dw)[(F - 2) / 2 + 1], const TapsF<T, F> &tp, T &xe, T &xo)
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


// ---- threshold! (threshold_main.jl:21-80) ----
// T: element type, C: the type Julia's promotion computes `x[i] op t` in
template <typename T, typename C>
__device__ __forceinline__ T threshold_one(T xr, int th, C t)
{
    const C xi = (C)xr;
    const C ax = xi < 0 ? -xi : xi;
    const C sg = (C)((xi > 0) - (xi < 0));
    T out = xr;
    switch (th) {
    case WL_TH_HARD: if (ax <= t) out = (T)0; break;
    case WL_TH_SOFT: { const C sh = ax - t; out = (sh < 0) ? (T)0 : (T)(sg * sh); } break;
    case WL_TH_SEMISOFT:
        if (xi <= 2 * t) {
            const C sh = ax - t;
            if (sh < 0) out = (T)0;
            else if (sh - t < 0) out = (T)(sg * sh * 2);
        }
        break;
    case WL_TH_STEIN: { const C sh = 1 - t * t / (xi * xi); out = (sh < 0) ? (T)0 : (T)(xi * sh); } break;
    case WL_TH_POS: if (xi > 0) out = (T)0; break;
    case WL_TH_NEG: if (xi < 0) out = (T)0; break;
    }
    return out;
}

// ---- threshold! fused into the stores of a Float32 level kernel (the translation-invariant batch) ----
// The reference computes `x[i] op t` in Float64 (t = sigma * dnt.t is a Float64).  Every kind zeroes the small coefficients, and
// "small" has a Float32 cut: a compare and a select per coefficient decide most of them without leaving single precision.
//   hard:            |x| <= t            <=>  |x| <= the largest Float32 <= t           (the cut is the whole threshold)
//   soft, semisoft:  |x| - t < 0         <=>  |x| <= the largest Float32 <  t           (semisoft: then x <= |x| < t <= 2t holds too)
//   Stein:           1 - t^2 / x^2 < 0   <==  |x| <= the largest Float32 <= 0.999 t     (a sufficient cut: the quotient's roundings are
//                                                                                       1e-16 relative; x = 0 gives -Inf < 0 as well)
// Coefficients above the cut of the last three kinds take threshold_one in Float64 -- inside a wave-uniform branch that a wave
// whose coefficients were all cut (most waves of a denoising problem) skips.  t <= 0 disables the cut (tf < 0): soft with t = 0
// keeps -0.0, Stein with t = 0 produces the reference's NaN at x = 0.
struct ThCut {
    float tf;       // |x| <= tf: zero
    double t;
    int th;         // wl_thtype; < 0: none
};
__device__ __forceinline__ ThCut th_make_cut(int th, double t)
{
    ThCut c;
    c.th = th; c.t = t; c.tf = -1.f;
    if (th < 0 || !(t > 0)) {
        if (th == WL_TH_HARD && t == 0) c.tf = 0.f;                  // (hard: |x| <= 0 zeroes -0.0)
        return c;
    }
    const double lim = (th == WL_TH_STEIN) ? 0.999 * t : t;
    float f = (float)lim;
    const bool strict = (th == WL_TH_SOFT || th == WL_TH_SEMISOFT);
    if ((double)f > lim || (strict && (double)f == lim)) f = (f > 0.f) ? __uint_as_float(__float_as_uint(f) - 1u) : -1.f;
    if (th != WL_TH_HARD && f > 3.4028234663852886e38f) f = 3.4028234663852886e38f;      // (t = Inf: an infinite x still takes the exact path -> NaN)
    c.tf = f;
    return c;
}
__device__ __forceinline__ float th_cut(const ThCut &c, float v) { return (__builtin_fabsf(v) <= c.tf) ? 0.f : v; }
// after th_cut on a group of coefficients: `above` = this lane still holds a coefficient above the cut
__device__ __forceinline__ bool th_needs_exact(const ThCut &c, bool above)
{
    return c.th > WL_TH_HARD && __builtin_amdgcn_ballot_w64(above) != 0;
}
__device__ __forceinline__ float th_exact(const ThCut &c, float v)
{
    return (__builtin_fabsf(v) <= c.tf) ? v : threshold_one<float, double>(v, c.th, c.t);     // (cut values are already 0)
}

// ---- hand-placed vector-memory loads / waits of the 2-D marching kernels (wl_fwd2d.hip, wl_pair2d.hip) ----
__device__ __forceinline__ void wg_lds_sync(bool multi)
{
    if (multi) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// LVL1 only gives the launch that consumes the full-size input its own symbol (rocprofv3 --stats then reports the dominant
// kernel separately from the same code running on the smaller levels).
// Column loads and their waits are written by hand.  hipcc's own wait-count insertion, given the rotating 16-slot ring, puts
// vmcnt(1) / vmcnt(0) in front of two of every eight steps (checked in the ISA: tools/probes/waitcnt_probe.hip has the small
// reproduction): the wave then waits for the loads it issued a few instructions earlier AND for all of its stores, twice per
// iteration -- the four-step prefetch distance never exists.  Here the load is opaque to the compiler and the wait names the
// two ring slots it guards, so every consumer depends on the wait through its data.
typedef float F4 __attribute__((ext_vector_type(4)));
typedef double D2 __attribute__((ext_vector_type(2)));
// ---- cache policy of the streaming kernels (round 5, profiles/r05_cache_policy.md) ----
// `nt` on a global load / store marks the line non-temporal: it streams through the XCD's L2 (and the Infinity Cache) instead of
// displacing what is resident.  A level kernel reads every input sample once and writes every detail coefficient once, while the
// approximation it writes is the NEXT launch's input: with the input loads (and the deeper level's detail stores) non-temporal, the
// approximation stays in L2 for the tail kernels, and the level kernel itself no longer evicts its own halo re-reads.
// Measured on 8192^2 f32 db4, three inputs in rotation: levels 1-2 114.5 -> 107.0 us, the whole 13-level transform 149.4 -> 128.2 us;
// one level (k_fwd2d_lds) 108.8 -> 96.2 us.  Float64 pair kernel: +4 % slower with it, so not there.  Per call site; the WL_P_*
// macros are the shipped choice (overridable with -D in tools/mkvariant.sh experiment builds).  Results are unaffected.
#ifndef WL_P_PAIR_LD
#define WL_P_PAIR_LD 1      // k_fwd2d_pair: main waves' column loads
#endif
#ifndef WL_P_PAIR_LDH
#define WL_P_PAIR_LDH 0     // ... the halo wave's loads
#endif
#ifndef WL_P_PAIR_ST1
#define WL_P_PAIR_ST1 0     // ... level-l detail stores
#endif
#ifndef WL_P_PAIR_ST2
#define WL_P_PAIR_ST2 1     // ... level-(l+1) detail stores
#endif
#ifndef WL_P_PAIR_LL
#define WL_P_PAIR_LL 2      // ... the level-(l+1) approximation (the next launch's input): write-through, see store_pol (r06)
#endif
#ifndef WL_P_LDS_LD
#define WL_P_LDS_LD 1       // k_fwd2d_lds (Float32): column loads
#endif
#ifndef WL_P_LDS_ST
#define WL_P_LDS_ST 0       // ... detail stores
#endif
#ifndef WL_P_LDS64_LD
#define WL_P_LDS64_LD 0     // k_fwd2d_lds64
#endif
#ifndef WL_P_PAIR64_LD
#define WL_P_PAIR64_LD 0    // k_fwd2d_pair64
#endif
#ifndef WL_P_LONG_LD
#define WL_P_LONG_LD 1      // k_fwd2d_lds_long (12..20 taps): 8192^2 sym8 293.1 -> 288.9 us
#endif
#ifndef WL_P_TILE_LD
#define WL_P_TILE_LD 0      // k_fwd2d_tileB: window loads
#endif
#ifndef WL_P_TILE_ST
#define WL_P_TILE_ST 2      // ... detail stores: write-through (r06: 2048^2 two levels 14.6 -> 12.7 us; with the approximation stores and the
                            //     pair's, 8192^2 L = 13 133.7 -> 131.2: the dependent launch no longer waits for 16 MB of dirty lines)
#endif
#ifndef WL_P_TILE_LL
#define WL_P_TILE_LL 2      // ... its approximation stores (k_fwd2d_tileB and k_fwd2d_tile)
#endif
#ifndef WL_P_TILE3_ST
#define WL_P_TILE3_ST 2     // k_fwd2d_tile: detail stores
#endif
#ifndef WL_P_M1D_LD
#define WL_P_M1D_LD 0       // k_fwd1d_multi: staging loads of the input tile
#endif
#ifndef WL_P_M1D_ST
#define WL_P_M1D_ST 1       // ... detail stores (C5 shard 939 -> 926 us, C2 49.6 -> 49.0; loads: slower)
#endif
#ifndef WL_P_LIFT3_LD
#define WL_P_LIFT3_LD 0     // k_lift1d_fwd3: input loads
#endif
#ifndef WL_P_LIFT3_ST
#define WL_P_LIFT3_ST 0     // ... level-1 detail stores
#endif
#ifndef WL_P_LIFTI3_LD
#define WL_P_LIFTI3_LD 1    // k_lift1d_inv3: detail loads (2^24 cdf9/7 inverse 50.9 -> 49.6 us)
#endif
#ifndef WL_P_LIFTI3_ST
#define WL_P_LIFTI3_ST 0    // ... output stores
#endif
#ifndef WL_P_IPAIR_LD
#define WL_P_IPAIR_LD 1     // k_inv2d_pair: coefficient loads of the level-l waves (with _ST: 8192^2 db4 idwt 155.9 -> 142.1 us)
#endif
#ifndef WL_P_IPAIR_LD2
#define WL_P_IPAIR_LD2 0    // ... of the level-(l+1) wave
#endif
#ifndef WL_P_IPAIR_ST
#define WL_P_IPAIR_ST 1     // ... output stores
#endif
#ifndef WL_P_I1D2_LD
#define WL_P_I1D2_LD 1      // k_inv1d_stream2: detail loads (2^24 db4 inverse 56.5 -> 55.8 us)
#endif
#ifndef WL_P_I1D2_ST
#define WL_P_I1D2_ST 0      // ... output stores
#endif
#ifndef WL_P_I2DS_LD
#define WL_P_I2DS_LD 0      // k_inv2d_stream: coefficient loads
#endif
#ifndef WL_P_I2DS_ST
#define WL_P_I2DS_ST 0      // ... output stores
#endif
#ifndef WL_P_ILONG_LD
#define WL_P_ILONG_LD 1     // k_inv2d_lds_long, Float32 only: coefficient loads (with _ST: sym5 idwt 171.4 -> 161.4, sym8 246.4 -> 237.5 us;
                            // Float64 +7 % slower with it)
#endif
#ifndef WL_P_ILONG_ST
#define WL_P_ILONG_ST 1     // ... output stores
#endif
#ifndef WL_P_LIFT2D_LD
#define WL_P_LIFT2D_LD 0    // k_lift2d_inv: coefficient loads
#endif
#ifndef WL_P_LIFT2D_ST
#define WL_P_LIFT2D_ST 1    // ... output stores (8192^2 cdf9/7 idwt 235.1 -> 231.9 us; loads: slower)
#endif
#ifndef WL_P_LIFT2DF_LD
#define WL_P_LIFT2DF_LD 0   // k_lift2d_fwd: input loads
#endif
#ifndef WL_P_LIFT2DF_ST
#define WL_P_LIFT2DF_ST 1   // ... detail stores (8192^2 cdf9/7 dwt 230.7 -> 220.2 us; loads: slower)
#endif
#ifndef WL_P_PAIR64_ST1
#define WL_P_PAIR64_ST1 1   // k_fwd2d_pair64: level-l detail stores (with _ST2: 8192^2 Float64 dwt 286.2 -> 282.4 us; loads: slower)
#endif
#ifndef WL_P_PAIR64_ST2
#define WL_P_PAIR64_ST2 1   // ... level-(l+1) detail stores
#endif
#ifndef WL_P_LDS64_ST
#define WL_P_LDS64_ST 0     // k_fwd2d_lds64: detail stores
#endif
#ifndef WL_P_LONG_ST
#define WL_P_LONG_ST 1      // k_fwd2d_lds_long: detail stores (sym8 dwt 291.2 -> 286.1 us)
#endif
// POL: 0 plain, 1 non-temporal, 2 write-through (`sc1`: the bytes leave the XCD's L2 at once and the line is dropped -- nothing of it
// is left dirty for the end-of-kernel write-back that the next dependent launch waits for; 16-byte and 8-byte vectors only)
template <int POL, typename V>
__device__ __forceinline__ void store_pol(V *p, const V v)
{
    if constexpr (POL == 2 && sizeof(V) == 16) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 2 && sizeof(V) == 8) asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory");
    else if constexpr (POL == 1) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT, typename V>
__device__ __forceinline__ V load_pol(const V *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
// V = any 16-byte vector type (F4: four Float32 rows, D2: two Float64 rows)
template <bool NT = false, typename V, typename T>
__device__ __forceinline__ void gload16(V &dst, const T *p)
{
    static_assert(sizeof(V) == 16, "one global_load_dwordx4");
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// ... with `sc1`: the vector L1 is bypassed and the line is served coherently at agent scope -- for data that workgroups of the SAME
// launch published with write-through (`sc1`) stores (the consumer side of store_pol<2>; guide: "sc1 stores AND sc1 loads")
template <typename V, typename T>
__device__ __forceinline__ void gload16_sc1(V &dst, const T *p)
{
    static_assert(sizeof(V) == 16, "one global_load_dwordx4");
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
// one row per lane (the halo wave of the fused pair kernel)
template <bool NT = false, typename T>
__device__ __forceinline__ void gload4(float &dst, const T *p)
{
    if constexpr (NT) asm volatile("global_load_dword %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// The same load, skipped when the wave-uniform flag is 0.  The destination is read-write for the compiler (the old contents
// survive a skipped load), so there is no control flow -- and no phi / register copy -- around the asynchronous load.
template <bool NT = false, typename V, typename T>
__device__ __forceinline__ void gload16_if(V &dst, const T *p, int flag)
{
    static_assert(sizeof(V) == 16, "one global_load_dwordx4");
    if constexpr (NT)
        asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 .Lwl_skip%=\n\tglobal_load_dwordx4 %0, %1, off nt\n.Lwl_skip%=:"
                     : "+v"(dst) : "v"(p), "s"(flag) : "memory", "scc");
    else
        asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 .Lwl_skip%=\n\tglobal_load_dwordx4 %0, %1, off\n.Lwl_skip%=:"
                     : "+v"(dst) : "v"(p), "s"(flag) : "memory", "scc");
}
// "at most N vector-memory operations outstanding".  Loads return in issue order among themselves, so this covers every load
// that has at least N younger LOADS behind it.  Do not count younger stores towards N: round 3 measured (one transform in
// about a thousand, some boxes only) that a store can be acknowledged while an older load is still in flight.
template <int N, typename V>
__device__ __forceinline__ void wait_vm(V &a, V &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
// The same wait with the count chosen by a wave-uniform flag at run time: vmcnt(N) when flag != 0, vmcnt(0) otherwise.  The
// branch lives INSIDE the asm statement for the reason gload16_if exists: a C++ `if` around two wait_vm calls makes the ring
// registers phis, and hipcc placed the phi copies of the drain side BEFORE its s_waitcnt (k_fwd2d_lds_long up to round 3:
// found by tools/isa_check.py) -- a copy of a register whose load may still be in flight.
template <int N, typename V>
__device__ __forceinline__ void wait_vm_sel(V &a, V &b, int flag)
{
    asm volatile("s_cmp_eq_u32 %3, 0\n\ts_cbranch_scc1 .Lwl_drain%=\n\ts_waitcnt vmcnt(%2)\n\ts_branch .Lwl_waited%=\n"
                 ".Lwl_drain%=:\n\ts_waitcnt vmcnt(0)\n.Lwl_waited%=:"
                 : "+v"(a), "+v"(b) : "n"(N), "s"(flag) : "memory", "scc");
}
// Every outstanding load has landed before any of the R registers of a ring is read, copied or reused.  For the kernels whose
// march ends without a consuming step (2 taps: the steps past the chunk neither load nor produce), so that the last prefetches
// -- columns nobody needs -- cannot land in registers the compiler has meanwhile given to something else.
template <int R, typename V>
__device__ __forceinline__ void drain_ring(V (&ring)[R])
{
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);
}


}  // namespace wl
