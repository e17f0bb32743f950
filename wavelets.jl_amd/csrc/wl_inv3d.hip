// wl_inv3d.hip -- one INVERSE 3-D filter-bank level in ONE pass over HBM (both element types, even F <= 8, lines of 32 ... 1024 rows).
//
//   k_inv3d_one<T, RPL, F, NW>    reference: columns -> rows -> planes of one level, transforms_filter.jl:264-287
//
// The mirror of k_fwd3d_one (wl_fwd3d.hip) as far as the reference's order of the passes allows.  The inverse reconstructs dim 1 first
// and dim 3 last, so the march runs along dim 3 (the ring holds planes that are already reconstructed along dims 1 and 2) and the
// tile is cut along dim 2:
//   * a workgroup owns WHOLE dim-1 lines (NW waves, RPL rows per lane), a TILE of 2 output column pairs (4 output columns) along
//     dim 2 and a SEGMENT of TK output plane pairs along dim 3;
//   * per step it takes one scaling and one detail coefficient plane: of each the SH + 2 scaling and SH + 2 detail coefficient
//     columns the tile's two dim-2 windows cover (s[p - SH .. p], d[p .. p + SH]) -- one 16-byte load per lane and column, in
//     rounds of SH + 2 columns: the lanes publish a round in LDS, and after the round's barrier every lane reads the dim-1 windows
//     of its own pairs back, reconstructs its RPL rows of each column (window_inv) and folds them into the running dim-2 sums of
//     the tile's four output columns (q ascending = the reference's order);
//   * the four columns of a finished plane go into a ring of SH + 1 scaling-plane and SH + 1 detail-plane slots; the dim-3
//     reconstruction reads the ring and stores two output planes (16 bytes per lane and column).
// Cost against the forward kernel: the dim-1 pass runs on every loaded column, i.e. on the (SH + 2) / 2 re-read columns too (the
// forward kernel only LOADS its shared planes twice); the re-reads themselves are L2 hits between neighbouring tiles of one XCD.
// Extents: lines of any multiple of 2 RPL rows, any even dim-2 / dim-3 extents (the last tile / segment is moved back to the edge and
// recomputes what it shares with its neighbour: same values, same addresses).
// Arithmetic: window_inv (wl_dev.h) term by term -- bit-identical to the axis kernels.
#include "wl_fast.h"
#include "wl_dev.h"

#ifndef WL_P_I3D1_ST
#define WL_P_I3D1_ST 0      // k_inv3d_one: output stores
#endif

namespace wl {

template <typename T, int F>
struct Inv3DArgs {
    const T *x; int64_t x1, x2;            // coefficient array, strides 1, x1, x2
    const T *ll;                           // approximation octant: dense (h0, h1, h2), or nullptr = in x
    T *out; int64_t o1, o2;                // reconstructed box, strides 1, o1, o2
    int n0, n1, n2;                        // output extents of the level
    int TK;                                // output plane pairs per segment (multiple of SH + 1)
    int nseg, ntile;
    TapsF<T, F> tp;
};

template <typename T, int N> struct VxI { typedef T type __attribute__((ext_vector_type(N))); };

template <int N, typename V>
__device__ __forceinline__ void wait_vm1i(V &a)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <typename V, typename T>
__device__ __forceinline__ void gload_v(V &dst, const T *p)
{
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "global_load_dwordx4 / x2");
    if constexpr (sizeof(V) == 16) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int POL, typename V, typename T>
__device__ __forceinline__ void gstore_si(T *sbase, uint32_t voff, const V v)
{
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "global_store_dwordx4 / x2");
    if constexpr (sizeof(V) == 16) {
        if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    } else {
        if constexpr (POL == 1) asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
        else asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    }
}

template <typename T, int RPL, int F, int NW>
__global__ void __launch_bounds__(64 * NW, 2) k_inv3d_one(Inv3DArgs<T, F> a)
{
    typedef typename VxI<T, RPL>::type V;                      // the lane's RPL rows of one column
    typedef typename VxI<T, 2>::type T2;
    constexpr int SH = (F - 2) / 2, NC = SH + 2, R = SH + 1;   // columns per round; ring slots per kind = steps per unrolled group
    constexpr int NPQ = RPL / 2;                               // coefficient pairs per lane along dim 1
    constexpr int CP = 4 + 64 * RPL * NW + 8;                  // column pitch in LDS: [4 wrapped s][s: h0][d: h0][8 wrapped d]
    static_assert(F >= 2 && F <= 8 && (F % 2) == 0, "ring of SH + 1 planes per kind");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *const lds = reinterpret_cast<T *>(smem_raw);            // [2][NC][CP]

    constexpr bool multi = NW > 1;
    const int lp = (int)threadIdx.x;
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8);
    const uint32_t logical = first + (b >> 3);
    const int tile = (int)(logical / (uint32_t)a.nseg);
    const int seg = (int)(logical % (uint32_t)a.nseg);

    const int n0 = a.n0, h0 = n0 >> 1, h1 = a.n1 >> 1, h2 = a.n2 >> 1;
    const int P0 = (2 * tile + 2 <= h1) ? 2 * tile : h1 - 2;           // first output column pair of the tile
    const int k0 = (seg * a.TK + a.TK <= h2) ? seg * a.TK : h2 - a.TK;  // first output plane pair of the segment
    const bool active = RPL * lp < n0;
    const bool lo = active && (RPL * lp < h0) && (a.ll != nullptr);   // this lane's rows of an all-scaling column come from ll
    const int64_t rofs = active ? (int64_t)RPL * lp : 0;
    const T *const xrow = a.x + rofs;
    const T *const lrow = lo ? (a.ll + rofs) : xrow;
    // LDS positions of this lane: its rows, the wrapped copies it owns, its windows
    const int wpos = 4 + RPL * lp;
    const int smir = (RPL * lp >= h0 - 4 && RPL * lp < h0) ? (RPL * lp - (h0 - 4)) : -1;
    const int dmir = (RPL * lp >= h0 && RPL * lp < h0 + 8) ? (4 + n0 + RPL * lp - h0) : -1;
    const int r0 = NPQ * lp;                                   // first pair of this lane
    const uint32_t vout = (uint32_t)sizeof(V) * (uint32_t)lp;

    // global column / plane indices (periodic)
    int P0l = P0;
    auto scol = [&](const int i) __attribute__((always_inline)) { int j = P0l - SH + i; if (j < 0) j += h1; if (j >= h1) j -= h1; return j; };
    auto dcol = [&](const int i) __attribute__((always_inline)) { int j = P0l + i; if (j >= h1) j -= h1; return j; };

    V L[2][NC];                                                // two rounds in flight
    V RS[R][4], RD[R][4];                                      // reconstructed planes: 4 output columns each
    V Se[2], So[2], De[2], Do[2];                              // running dim-2 sums of the two column pairs

    // request round `rnd` (0: scaling columns, 1: detail columns) of coefficient plane (kind, z) into landing half `half`
    auto request = [&](const int half, const int kind, const int z, const int rnd) __attribute__((always_inline)) {
        // (the tile origin is made opaque per request: hipcc otherwise hoists the per-lane 64-bit column addresses of every round out of
        //  the march -- 20 of them -- and spills: 256 VGPRs + scratch against 233, none)
        asm volatile("" : "+s"(P0l));
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const T *p;
            if (rnd == 0) {
                const int j = scol(i);
                if (kind == 0) p = lo ? (lrow + (int64_t)j * h0 + (int64_t)z * h0 * h1) : (xrow + (int64_t)j * a.x1 + (int64_t)z * a.x2);
                else p = xrow + (int64_t)j * a.x1 + (int64_t)(h2 + z) * a.x2;
            } else {
                const int j = h1 + dcol(i);
                p = xrow + (int64_t)j * a.x1 + (int64_t)(kind == 0 ? z : h2 + z) * a.x2;
            }
            gload_v(L[half][i], p);
        }
    };
    auto plane_s = [&](const int t) __attribute__((always_inline)) { int z = k0 + t; if (z < 0) z += h2; return z; };
    auto plane_d = [&](const int t) __attribute__((always_inline)) { int z = k0 + t + SH; if (z >= h2) z -= h2; return z; };

    // ---- the step, written out: four rounds ----
    // kind / rnd / half are compile-time at every call; `more` = the round two ahead exists (else nothing is requested and the
    // remaining waits drain)
    auto do_round = [&](const int half, const int rnd, const bool drain, auto &&next_request) __attribute__((always_inline)) {
        T *const buf = lds + half * NC * CP;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            // the NC loads of the other half were requested after this round's: every load of this round has at least NC younger loads
            if (drain) wait_vm1i<0>(L[half][i]);
            else wait_vm1i<NC>(L[half][i]);
            const V v = L[half][i];
            if (active) *reinterpret_cast<V *>(buf + i * CP + wpos) = v;
            if (smir >= 0) *reinterpret_cast<V *>(buf + i * CP + smir) = v;
            if (dmir >= 0) *reinterpret_cast<V *>(buf + i * CP + dmir) = v;
        }
        next_request();                                        // (asm volatile + "memory": stays behind the LDS stores above)
        wg_lds_sync(multi);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const T *const col = buf + i * CP;
            // the lane's pairs r0 .. r0 + NPQ - 1: scaling coefficients s[r0 - SH ..], detail coefficients d[r0 ..]
            T sv[NPQ + 4], dv[NPQ + 4];                        // s[r0 - 4 .. r0 + NPQ - 1], d[r0 .. r0 + NPQ + 3]
            if constexpr (NPQ == 2) {
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const T2 s2 = *reinterpret_cast<const T2 *>(col + r0 + 2 * e);             // position 4 + (r0 - 4) + 2 e
                    const T2 d2 = *reinterpret_cast<const T2 *>(col + 4 + h0 + r0 + 2 * e);
                    sv[2 * e] = s2.x; sv[2 * e + 1] = s2.y;
                    dv[2 * e] = d2.x; dv[2 * e + 1] = d2.y;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NPQ + 4; ++e) {
                    sv[e] = (e >= 4 - SH) ? col[r0 + e] : (T)0;                                // (only s[r0 - SH ..] is read)
                    dv[e] = (e <= SH) ? col[4 + h0 + r0 + e] : (T)0;
                }
            }
            V v;
#pragma unroll
            for (int p = 0; p < NPQ; ++p) {
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = sv[4 + p - SH + q]; dw[q] = dv[p + q]; }
                T xe, xo;
                window_inv<T, F>(sw, dw, a.tp, xe, xo);
                v[2 * p] = xe; v[2 * p + 1] = xo;
            }
            // fold column i into the dim-2 sums: scaling column i is sw[i] of pair 0 and sw[i - 1] of pair 1, detail likewise
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int q = i - pr;
                if (q < 0 || q > SH) continue;
                if (rnd == 0) {
#pragma unroll
                    for (int e = 0; e < RPL; ++e) {
                        Se[pr][e] = (q == 0) ? a.tp.h[F - 2] * v[e] : Se[pr][e] + a.tp.h[F - 2 - 2 * q] * v[e];
                        So[pr][e] = (q == 0) ? a.tp.h[F - 1] * v[e] : So[pr][e] + a.tp.h[F - 1 - 2 * q] * v[e];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < RPL; ++e) {
                        De[pr][e] = (q == 0) ? a.tp.g[1] * v[e] : De[pr][e] + a.tp.g[1 + 2 * q] * v[e];
                        Do[pr][e] = (q == 0) ? a.tp.g[0] * v[e] : Do[pr][e] + a.tp.g[2 * q] * v[e];
                    }
                }
            }
            // anchor: the sums are "used" here, so the column's arithmetic is complete before the next column's window is read (hipcc
            // otherwise read every window of the round, spilled them, and ran the arithmetic rounds later: 700 spilled VGPRs)
            if (rnd == 0) asm volatile("" : "+v"(Se[0]), "+v"(Se[1]), "+v"(So[0]), "+v"(So[1]));
            else asm volatile("" : "+v"(De[0]), "+v"(De[1]), "+v"(Do[0]), "+v"(Do[1]));
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto finish_plane = [&](V (&slot)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                slot[2 * pr][e] = Se[pr][e] + De[pr][e];
                slot[2 * pr + 1][e] = So[pr][e] + Do[pr][e];
            }
        }
        asm volatile("" : "+v"(slot[0]), "+v"(slot[1]), "+v"(slot[2]), "+v"(slot[3]));
    };

    // t: step (from -SH: the first SH steps only fill the rings); u: its ring position (t mod R, compile-time); last: nothing follows
    auto step = [&](const int t, const int u, const bool emit, const bool last) __attribute__((always_inline)) {
        const int zd = plane_d(t), zs1 = plane_s(t + 1);
        // rounds: (s-plane, s-cols) half 0, (s-plane, d-cols) half 1, (d-plane, s-cols) half 0, (d-plane, d-cols) half 1;
        // each requests the round two ahead into the half it has just published
        do_round(0, 0, false, [&]() __attribute__((always_inline)) { request(0, 1, zd, 0); });
        do_round(1, 1, false, [&]() __attribute__((always_inline)) { request(1, 1, zd, 1); });
        finish_plane(RS[u]);
        do_round(0, 0, last, [&]() __attribute__((always_inline)) { if (!last) request(0, 0, zs1, 0); });
        do_round(1, 1, last, [&]() __attribute__((always_inline)) { if (!last) request(1, 0, zs1, 1); });
        finish_plane(RD[(u + SH) % R]);
        if (!emit) return;
        // ---- dim 3: scaling planes kp - SH .. kp = ring slots (u + 1 + q) mod R, detail planes kp .. kp + SH = slots (u + q) mod R ----
        const int kp = k0 + t;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            V xe, xo;
#pragma unroll
            for (int e = 0; e < RPL; ++e) {
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int q = 0; q <= SH; ++q) { sw[q] = RS[(u + 1 + q) % R][c][e]; dw[q] = RD[(u + q) % R][c][e]; }
                T e0, o0;
                window_inv<T, F>(sw, dw, a.tp, e0, o0);
                xe[e] = e0; xo[e] = o0;
            }
            T *const ob = a.out + (int64_t)(2 * P0 + c) * a.o1 + (int64_t)(2 * kp) * a.o2;
            if (active) {
                gstore_si<WL_P_I3D1_ST>(ob, vout, xe);
                gstore_si<WL_P_I3D1_ST>(ob + a.o2, vout, xo);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // the first two rounds (the scaling plane of step -SH)
    request(0, 0, plane_s(-SH), 0);
    request(1, 0, plane_s(-SH), 1);
#pragma unroll
    for (int i = 0; i < SH; ++i) step(i - SH, (i + 1) % R, false, false);
    const int S = a.TK;
    int t0 = 0;
    for (; t0 < S - R; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u, true, false);
    }
#pragma unroll
    for (int u = 0; u < R; ++u) step(t0 + u, u, true, u == R - 1);
}

template <typename T>
static int inv3d_rpl(int64_t n0)
{
    if (n0 < 32 || n0 > 1024) return 0;
    if (sizeof(T) == 8 && n0 > 512) return 0;                  // (eight Float64 waves: 11 spilled VGPRs -- a scratch reload drains the prefetches)
    if (sizeof(T) == 4 && n0 > 128 && (n0 % 8) == 0) return 4;
    return (n0 % 4) == 0 ? 2 : 0;
}

template <typename T>
bool inv3d_one_ok(int F, const T *x, int64_t x1, int64_t x2, const T *ll, const T *out, int64_t o1, int64_t o2, const int64_t n[3], bool any_tier)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    if (opt("WL_I3D_ONE", 1) == 0) return false;
    if (F < 2 || F > 8 || (F & 1)) return false;
    const int64_t n0 = n[0], n1 = n[1], n2 = n[2];
    if (inv3d_rpl<T>(n0) == 0) return false;
    if (n1 < 16 || (n1 % 2) != 0 || n1 > (1 << 20) || n2 < 16 || (n2 % 2) != 0 || n2 > (1 << 20)) return false;
    if ((x1 % VEC) != 0 || (x2 % VEC) != 0 || (o1 % VEC) != 0 || (o2 % VEC) != 0 || x1 < n0 || o1 < n0) return false;
    if (((uintptr_t)x & 15) != 0 || ((uintptr_t)out & 15) != 0 || (ll && ((uintptr_t)ll & 15) != 0)) return false;
    if (x == out || ll == out) return false;
    // Where it pays (measured, level 1 alone, us): the dim-1 pass runs on (SH + 2) / 2 times the columns, so the gain shrinks with the
    // filter length --
    //   Float32 512^3: haar 406 -> 201, db2 418 -> 219, db3 424 -> 352, db4 446 -> 359;  300^3 db4 249 -> 180 (any-extent tier);
    //   256^3 db2 46 -> 33, haar 44 -> 22;  Float64 512^3 db2 839 -> 446, 256^3 db2 105 -> 70;
    //   not taken: 256^3 db4 60 -> 67, 512 x 512 x 256 db4 221 -> 272 (half the resident waves);
    //   Float64 6 / 8 taps: 512^3 db4 idwt L = 9 1001 -> 837, db3 level 859 -> 619, 300^3 db4 461 -> 297 (lost while the kernel still spilled:
    //   830 -> 1067 with 28 VGPRs in scratch -- every reload drains the prefetches)
    if (sizeof(T) == 8 && F > opt("WL_I3D_ONE_F64_FMAX", 8)) return false;
    // 2 / 4 taps: ahead down to about 10^6 elements on every tier (120^3 db2 23.6 -> 9.9, 128^3 19.4 -> 10.0, 200^3 54 -> 22, Float64 200^3
    // 96 -> 35); 6 / 8 taps: 2^27 elements against the plane + axis kernels, 2^23 against the any-extent passes (240 x 240 x 160 db4 67 -> 59,
    // 200^3 62 -> 59; 160^3 39 -> 51 not taken)
    const long long gate = (F <= 4) ? opt("WL_I3D_ONE_MIN", (long long)1 << 20)
                                    : (any_tier ? opt("WL_I3D_ONE_MIN_ANY", (long long)1 << 23) : opt("WL_I3D_ONE_MIN_LONG", (long long)1 << 27));
    if (n0 * n1 * n2 < gate) return false;
    return true;
}
template bool inv3d_one_ok<float>(int, const float *, int64_t, int64_t, const float *, const float *, int64_t, int64_t, const int64_t[3], bool);
template bool inv3d_one_ok<double>(int, const double *, int64_t, int64_t, const double *, const double *, int64_t, int64_t, const int64_t[3], bool);

template <typename T, int RPL, int F, int NW>
static hipError_t launch_inv3d_inst(hipStream_t st, unsigned nwg, const Inv3DArgs<T, F> &a)
{
    constexpr int SH = (F - 2) / 2, NC = SH + 2, CP = 4 + 64 * RPL * NW + 8;
    const size_t shmem = (size_t)2 * NC * CP * sizeof(T);
    static thread_local int attr_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool done = false;
    for (int i = 0; i < 8; ++i) done = done || attr_dev[i] == dev;
    if (!done && shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_inv3d_one<T, RPL, F, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        for (int i = 0; i < 8; ++i) if (attr_dev[i] < 0) { attr_dev[i] = dev; break; }
    }
    hipLaunchKernelGGL((k_inv3d_one<T, RPL, F, NW>), dim3(nwg), dim3(64 * NW), shmem, st, a);
    return hipGetLastError();
}

template <typename T, int F>
static hipError_t launch_inv3d_f(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *ll, T *out, int64_t o1,
                                 int64_t o2, const int64_t n[3], int cu_count)
{
    constexpr int R = (F - 2) / 2 + 1;
    Inv3DArgs<T, F> a;
    a.x = x; a.x1 = x1; a.x2 = x2; a.ll = ll; a.out = out; a.o1 = o1; a.o2 = o2;
    a.n0 = (int)n[0]; a.n1 = (int)n[1]; a.n2 = (int)n[2];
    const int rpl = inv3d_rpl<T>(n[0]);
    if (rpl == 0) return hipErrorInvalidValue;
    int W = 1;
    while ((int64_t)64 * rpl * W < n[0]) W <<= 1;
    const int h1 = a.n1 >> 1, h2 = a.n2 >> 1;
    a.ntile = (h1 + 1) / 2;
    // segment: TK output plane pairs, a multiple of the ring length, <= h2; shorter while fewer than 8 waves per CU would be resident
    int TK = (int)opt("WL_I3D_ONE_TK", 32);
    TK = (TK / R) * R;
    if (TK < R) TK = R;
    while (TK > R && (TK > h2 || (int64_t)a.ntile * ((h2 + TK - 1) / TK) * W < (int64_t)cu_count * opt("WL_I3D_ONE_WAVES", 4))) TK -= R;
    if (TK > h2) return hipErrorInvalidValue;
    a.TK = TK;
    a.nseg = (h2 + TK - 1) / TK;
    a.tp = shrink<T, F>(taps);
    const unsigned nwg = (unsigned)(a.ntile * a.nseg);
    if constexpr (sizeof(T) == 4) {
        if (rpl == 2 && W == 1) return launch_inv3d_inst<T, 2, F, 1>(st, nwg, a);
        if (rpl == 2 && W == 2) return launch_inv3d_inst<T, 2, F, 2>(st, nwg, a);
        if (rpl == 2 && W == 4) return launch_inv3d_inst<T, 2, F, 4>(st, nwg, a);
        if (rpl == 2) return launch_inv3d_inst<T, 2, F, 8>(st, nwg, a);
        if (W == 1) return launch_inv3d_inst<T, 4, F, 1>(st, nwg, a);
        if (W == 2) return launch_inv3d_inst<T, 4, F, 2>(st, nwg, a);
        return launch_inv3d_inst<T, 4, F, 4>(st, nwg, a);
    } else {
        if (W == 1) return launch_inv3d_inst<T, 2, F, 1>(st, nwg, a);
        if (W == 2) return launch_inv3d_inst<T, 2, F, 2>(st, nwg, a);
        return launch_inv3d_inst<T, 2, F, 4>(st, nwg, a);
    }
}

template <typename T>
hipError_t inv3d_one_launch(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *ll, T *out, int64_t o1, int64_t o2,
                            const int64_t n[3], int cu_count)
{
    switch (taps.F) {
    case 2: return launch_inv3d_f<T, 2>(st, taps, x, x1, x2, ll, out, o1, o2, n, cu_count);
    case 4: return launch_inv3d_f<T, 4>(st, taps, x, x1, x2, ll, out, o1, o2, n, cu_count);
    case 6: return launch_inv3d_f<T, 6>(st, taps, x, x1, x2, ll, out, o1, o2, n, cu_count);
    case 8: return launch_inv3d_f<T, 8>(st, taps, x, x1, x2, ll, out, o1, o2, n, cu_count);
    default: return hipErrorInvalidValue;
    }
}
template hipError_t inv3d_one_launch<float>(hipStream_t, const Taps<float> &, const float *, int64_t, int64_t, const float *, float *, int64_t, int64_t,
                                            const int64_t[3], int);
template hipError_t inv3d_one_launch<double>(hipStream_t, const Taps<double> &, const double *, int64_t, int64_t, const double *, double *, int64_t,
                                             int64_t, const int64_t[3], int);

}  // namespace wl
