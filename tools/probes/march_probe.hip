// Scratch microbenchmark (not product code): pure data movement in the access geometry of the fused 2-D DWT level
// kernels -- what is the floor each geometry allows before any arithmetic?  8192 x 8192 f32 in, same bytes out.
//
// A wave owns a strip of rows (64 lanes x RPL rows, 16-byte loads) and marches along the columns of a chunk with
// the production kernels' 16-slot column ring (loads 4 steps ahead, loop unrolled x8).  Variants:
//   MODE 0  level-1 outputs only: four quadrant streams, even/odd-lane 16-byte stores        (k_fwd2d_stream)
//   MODE 1  fused pair: three level-1 detail streams + four level-2 streams, 8-byte stores   (k_fwd2d_stream2)
//   MODE 2  fused pair with the level-2 outputs regrouped over lane quads into 16-byte stores
//   MODE 3  single output stream (plain transposed-free copy in the marching order)
// Parameters: valid lanes [lo, hi), chunk length TJ, waves per workgroup (adjacent strips), leading-dimension pad.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

struct Args {
    const float *in; float *out;
    long n, ldi, ldo;
    int TJ, nstrips, nchunks, lo, hi, wpb, extra;   // extra: steps beyond the owned range (pair kernels run 8 more)
};

template <int MODE>
__global__ void __launch_bounds__(256) march(Args a)
{
    constexpr int RPL = 4, R = 16, U = 8, PFD = 4;
    const int lane = threadIdx.x & 63;
    const unsigned b = blockIdx.x, nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const unsigned lwg = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    const unsigned logical = lwg * a.wpb + (threadIdx.x >> 6);
    if (logical >= (unsigned)(a.nstrips * a.nchunks)) return;
    const int strip = logical % a.nstrips, chunk = logical / a.nstrips;
    const long n = a.n, hm = n >> 1, nxj = n >> 1, hm2 = n >> 2, nxj2 = n >> 2;
    const int VR = (a.hi - a.lo) * RPL;
    const long gi = (long)strip * VR + (long)(lane - a.lo) * RPL;
    long row = gi; if (row < 0) row += n; if (row >= n) row -= n;
    const bool own = lane >= a.lo && lane < a.hi && gi < n;
    const long ko = gi >> 1, ko2 = gi >> 2;
    const bool odd = lane & 1;
    const long j0 = (long)chunk * a.TJ;
    const int S_own = a.TJ >> 1, S = S_own + a.extra;
    const float *base = a.in + row;
    float4 ring[R];
#pragma unroll
    for (int c = 0; c < R - 2; ++c) { long jc = j0 + c; if (jc >= n) jc -= n; ring[c] = *(const float4 *)(base + jc * a.ldi); }
    const long kbase = j0 >> 1, kbase2 = j0 >> 2;
    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                long jc = j0 + 2 * t + (R - 2) + e; if (jc >= n) jc -= n; if (jc >= n) jc -= n;
                ring[(2 * u + R - 2 + e) % R] = *(const float4 *)(base + jc * a.ldi);
            }
        }
        const float4 v0 = ring[(2 * u) % R], v1 = ring[(2 * u + 1) % R];
        const long k = kbase + t;
        long kd = k + 3; if (kd >= nxj) kd -= nxj;
        if (MODE == 0) {
            if (own) {
                float *pP = odd ? (a.out + ko - 2 + (nxj + kd) * a.ldo) : (a.out + ko + k * a.ldo);
                float *pQ = odd ? (a.out + ko - 2 + (nxj + kd) * a.ldo + hm) : (a.out + ko + k * a.ldo + hm);
                *(float4 *)pP = v0; *(float4 *)pQ = v1;
            }
        } else if (MODE == 3) {
            if (own) { *(float4 *)(a.out + gi + (j0 + 2 * t) * a.ldo) = v0; *(float4 *)(a.out + gi + (j0 + 2 * t + 1) * a.ldo) = v1; }
        } else {
            if (t < S_own && own) {
                float *p0 = odd ? (a.out + ko - 2 + (nxj + kd) * a.ldo) : (a.out + ko + k * a.ldo + hm);
                *(float4 *)p0 = v0;
                if (odd) *(float4 *)(a.out + ko - 2 + (nxj + kd) * a.ldo + hm) = v1;
            }
            if (MODE == 1) {
                if ((u & 1) && t >= U - 1 && t < S_own + U - 1 && own) {
                    const long k2 = kbase2 + ((t - (U - 1)) >> 1);
                    long kd2 = k2 + 3; if (kd2 >= nxj2) kd2 -= nxj2;
                    float *p0 = odd ? (a.out + (ko2 - 1) + (nxj2 + kd2) * a.ldo) : (a.out + ko2 + k2 * a.ldo);
                    float *p1 = odd ? (a.out + (ko2 - 1) + (nxj2 + kd2) * a.ldo + hm2) : (a.out + ko2 + k2 * a.ldo + hm2);
                    *(float2 *)p0 = make_float2(v1.x, v1.y); *(float2 *)p1 = make_float2(v1.z, v1.w);
                }
            } else {
                if ((u & 1) && t >= U - 1 && t < S_own + U - 1 && own) {
                    // lane quad (4 level-2 rows): lane q stores the 4 rows of subband q at column k2 / kd2
                    const int q = lane & 3;
                    const long k2 = kbase2 + ((t - (U - 1)) >> 1);
                    long kd2 = k2 + 3; if (kd2 >= nxj2) kd2 -= nxj2;
                    const long r2 = ko2 - q;       // first row of the quad
                    float *p = a.out + r2 + ((q & 2) ? (nxj2 + kd2) : k2) * a.ldo + ((q & 1) ? hm2 : 0);
                    *(float4 *)p = v1;
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

template <typename F> float timeit(F f, int reps = 25) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> v;
    for (int i = 0; i < reps; ++i) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); v.push_back(ms); }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2] * 1e3f;
}

int main() {
    const long n = 8192;
    const long padmax = 256;
    float *in, *out;
    CK(hipMalloc(&in, (n + padmax) * n * 4)); CK(hipMalloc(&out, (n + padmax) * n * 4));
    CK(hipMemset(in, 1, (n + padmax) * n * 4)); CK(hipMemset(out, 0, (n + padmax) * n * 4));
    struct Cfg { int mode, lo, hi, TJ, wpb, pad; };
    std::vector<Cfg> cfgs;
    // production geometries
    cfgs.push_back({0, 2, 62, 128, 1, 0});
    cfgs.push_back({1, 6, 58, 128, 1, 0});
    // one-sided halo geometries
    cfgs.push_back({0, 0, 62, 128, 1, 0});
    cfgs.push_back({1, 0, 59, 128, 1, 0});
    cfgs.push_back({1, 0, 56, 128, 1, 0});
    cfgs.push_back({2, 0, 56, 128, 1, 0});
    cfgs.push_back({2, 4, 60, 128, 1, 0});
    // all lanes valid (an LDS halo exchange would allow this)
    cfgs.push_back({0, 0, 64, 128, 1, 0});
    cfgs.push_back({1, 0, 64, 128, 1, 0});
    cfgs.push_back({2, 0, 64, 128, 1, 0});
    cfgs.push_back({3, 0, 64, 128, 1, 0});
    // chunk length
    for (int tj : {64, 256, 512}) { cfgs.push_back({0, 0, 64, tj, 1, 0}); cfgs.push_back({2, 0, 64, tj, 1, 0}); cfgs.push_back({2, 0, 56, tj, 1, 0}); }
    // waves per workgroup
    for (int w : {2, 4}) { cfgs.push_back({0, 0, 64, 128, w, 0}); cfgs.push_back({2, 0, 64, 128, w, 0}); cfgs.push_back({2, 0, 56, 128, w, 0}); cfgs.push_back({1, 6, 58, 128, w, 0}); }
    // leading-dimension padding (both arrays)
    for (int p : {16, 64, 256}) { cfgs.push_back({0, 2, 62, 128, 1, p}); cfgs.push_back({1, 6, 58, 128, 1, p}); cfgs.push_back({2, 0, 64, 128, 1, p}); cfgs.push_back({3, 0, 64, 128, 1, p}); }
    const double bytes = 2.0 * n * n * 4;
    for (const Cfg &c : cfgs) {
        Args a;
        a.in = in; a.out = out; a.n = n; a.ldi = n + c.pad; a.ldo = n + c.pad; a.TJ = c.TJ; a.lo = c.lo; a.hi = c.hi; a.wpb = c.wpb;
        const int VR = (c.hi - c.lo) * 4;
        a.nstrips = (int)((n + VR - 1) / VR); a.nchunks = (int)(n / c.TJ); a.extra = (c.mode == 1 || c.mode == 2) ? 8 : 0;
        const unsigned nwg = (a.nstrips * a.nchunks + c.wpb - 1) / c.wpb;
        float us = 0;
        switch (c.mode) {
        case 0: us = timeit([&] { march<0><<<nwg, 64 * c.wpb>>>(a); }); break;
        case 1: us = timeit([&] { march<1><<<nwg, 64 * c.wpb>>>(a); }); break;
        case 2: us = timeit([&] { march<2><<<nwg, 64 * c.wpb>>>(a); }); break;
        case 3: us = timeit([&] { march<3><<<nwg, 64 * c.wpb>>>(a); }); break;
        }
        CK(hipGetLastError());
        printf("mode %d lanes [%2d,%2d) TJ %3d wpb %d pad %3d waves %5d: %7.1f us  %5.2f TB/s  frac8 %.3f\n", c.mode, c.lo, c.hi, c.TJ, c.wpb, c.pad,
               a.nstrips * a.nchunks, us, bytes / us / 1e6, bytes / us / 1e6 / 8.0);
    }
    return 0;
}
