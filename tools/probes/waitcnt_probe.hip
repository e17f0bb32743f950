// Compile-only probe (hipcc -S): what s_waitcnt vmcnt values does the compiler place in a rotating-ring prefetch loop?
// Variants: -DCOND=0/1 (loads under a divergent condition), -DSTORES=0/1, -DDRAIN (vmcnt(0) after the prologue).
#include <hip/hip_runtime.h>
typedef float T4 __attribute__((ext_vector_type(4)));
#ifndef COND
#define COND 1
#endif
#ifndef STORES
#define STORES 1
#endif
__global__ void __launch_bounds__(64) k(const float *src, float *dst, long ld, int S, int nload, float h0, float h1)
{
    constexpr int R = 16, U = 8;
    const int lp = threadIdx.x;
    const bool loader = lp < nload;
    const float *base = src + 4 * lp;
    T4 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T4{0, 0, 0, 0};
#if COND
    if (loader)
#endif
    {
#pragma unroll
        for (int c = 0; c < R - 2; ++c) ring[c] = *reinterpret_cast<const T4 *>(base + c * ld);
    }
#ifdef DRAIN
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    auto step = [&](const int t, const int u) __attribute__((always_inline)) {
#if COND
        if (loader)
#endif
        {
#pragma unroll
            for (int e = 0; e < 2; ++e) ring[(2 * u + R - 2 + e) % R] = *reinterpret_cast<const T4 *>(base + (2 * t + R - 2 + e) * ld);
        }
        T4 acc = h0 * ring[(2 * u) % R];
#pragma unroll
        for (int m = 1; m < 8; ++m) acc = acc + h1 * ring[(2 * u + m) % R];
#if STORES
        *reinterpret_cast<T4 *>(dst + 4 * lp + t * ld) = acc;
        *reinterpret_cast<T4 *>(dst + 4 * lp + (t + 4096) * ld) = acc * h0;
#else
        if (acc.x == 123.f) dst[lp] = acc.y;
#endif
    };
    for (int t0 = 0; t0 < S; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u);
    }
}
