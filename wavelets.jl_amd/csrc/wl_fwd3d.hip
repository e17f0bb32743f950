// wl_fwd3d.hip -- one forward 3-D filter-bank level in ONE pass over HBM (Float32, even F <= 8, lines of 256 / 512 / 1024).
//
//   k_fwd3d_one<F, PD>    reference: planes -> rows -> columns of one level, transforms_filter.jl:246-263
//
// The three-launch / two-launch levels of wl_axis.hip read and write the box twice (dim 3 into a scratch box, then dims 2 + 1 per
// plane).  Here the level-l box is read once and the level's coefficients are written once:
//
//   * a workgroup owns WHOLE dim-1 lines (n0 = 256 W rows, W waves, 4 rows per lane: every global access is a 16-byte vector and
//     the dim-1 window of the topmost lanes wraps inside the workgroup's own LDS exchange -- no halo rows, no helper wave),
//     a TILE of 4 raw planes along dim 3 (two scaling + two detail planes of the level) and a SEGMENT of TJ columns along dim 2;
//   * it marches along dim 2.  Per column the F + 2 raw planes the tile's windows cover arrive in a landing ring of VGPRs
//     (hand-placed global_load_dwordx4 + named s_waitcnt, as in wl_fwd2d.hip) and are folded into the four dim-3 sums as they
//     land (m ascending = the reference's order); the sums go into an 8-slot column ring;
//   * the dim-2 pass runs on that ring in registers, the dim-1 pass through the LDS exchange of k_fwd2d_lds (same window
//     algebra: lane L' makes s rows 2L', 2L'+1 and d rows 2L'+4, 2L'+5 of four output planes, lane pairs swap halves so that
//     every store is a 16-byte vector).
//
// Traffic.  A tile reads F + 2 planes for 4 planes of payload -- (F + 2) / 4 times the box -- but the tiles that share planes are
// neighbours along dim 3, mapped to the SAME XCD and marching in step: the re-reads are L2 hits (34 TB/s aggregate), HBM sees the
// box once plus the F - 2 prologue columns of every segment ((TJ + F - 2) / TJ).  Nothing of the level is written twice.
//
// Arithmetic: the closed form of wl_internal.h for every axis, bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

#ifndef WL_P_3D1_LD
#define WL_P_3D1_LD 0       // k_fwd3d_one: raw plane loads (0 plain: the neighbouring tiles' re-reads are meant to hit L2)
#endif
#ifndef WL_P_3D1_ST
#define WL_P_3D1_ST 0       // ... detail stores (nt: 262-268 -> 245-254 us without it on 512^3)
#endif
#ifndef WL_P_3D1_LL
#define WL_P_3D1_LL 0       // ... the approximation corner (the next level's input)
#endif

namespace wl {

template <int F>
struct Fwd3DArgs {
    const float *src; int64_t c1, c2;      // level-l box, strides 1, c1, c2
    float *y; int64_t y1, y2;              // full array, strides 1, y1, y2
    float *ll;                             // approximation corner: dense (h0, h1, h2), or nullptr = into y
    int n0, n1, n2;
    int TJ;                                // owned dim-2 columns per segment (multiple of 8)
    int nseg, ntile;
    TapsF<float, F> tp;
};

// element-wise forms (this file is built with -fno-slp-vectorize: scalar v_mul / v_add take a tap straight from its SGPR, the packed
// forms wanted the taps duplicated into aligned SGPR pairs and their operands in aligned VGPR pairs)
typedef float F2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ F4 smul(float h, const F4 x) { return F4{h * x.x, h * x.y, h * x.z, h * x.w}; }
__device__ __forceinline__ F4 smad(const F4 acc, float h, const F4 x) { return F4{acc.x + h * x.x, acc.y + h * x.y, acc.z + h * x.z, acc.w + h * x.w}; }
__device__ __forceinline__ F2 smul(float h, const F2 x) { return F2{h * x.x, h * x.y}; }
__device__ __forceinline__ F2 smad(const F2 acc, float h, const F2 x) { return F2{acc.x + h * x.x, acc.y + h * x.y}; }

template <int N, typename V>
__device__ __forceinline__ void wait_vm1(V &a)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}

// 16 bytes per lane from (64-bit scalar base) + (32-bit unsigned per-lane byte offset)
template <bool NT, typename V, typename T>
__device__ __forceinline__ void gload16_s(V &dst, const T *sbase, uint32_t voff)
{
    static_assert(sizeof(V) == 16, "one global_load_dwordx4");
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

// POL: 0 plain, 1 non-temporal, 2 write-through (store_pol, wl_dev.h)
template <int POL, typename V, typename T>
__device__ __forceinline__ void gstore16_s(T *sbase, uint32_t voff, const V v)
{
    static_assert(sizeof(V) == 16, "one global_store_dwordx4");
    if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
}

// The same load tied to the four running sums ("+v"): everything that feeds them -- every product of the plane that just left `dst` --
// is complete before the load is issued, so the register allocator can give the new load the old plane's registers (without the tie
// hipcc sank the products below the following loads and the landing ring took twice its registers: spills, and a compiler-placed
// vmcnt(0) per scratch reload)
template <bool NT, typename V, typename T>
__device__ __forceinline__ void gload16_s_tied(V &dst, const T *sbase, uint32_t voff, V &t0, V &t1, V &t2, V &t3)
{
    static_assert(sizeof(V) == 16, "one global_load_dwordx4");
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %5, %6 nt" : "=v"(dst), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(voff), "s"(sbase) : "memory");
    else asm volatile("global_load_dwordx4 %0, %5, %6" : "=v"(dst), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(voff), "s"(sbase) : "memory");
}

template <int F, int PD, int NW, int G>
__global__ void __launch_bounds__(64 * NW * G, 2) k_fwd3d_one(Fwd3DArgs<F> a)
{
    typedef float T;
    typedef float T2 __attribute__((ext_vector_type(2)));
    typedef float T4 __attribute__((ext_vector_type(4)));
    constexpr int SH = (F - 2) / 2, KR = F + 2, RS = 8, U = 4, D = PD * KR;
    static_assert(F >= 2 && F <= 8 && (F % 2) == 0, "8-slot column ring");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    constexpr bool multi = NW * G > 1;
    // G tiles (neighbours along dim 3) per workgroup: their waves meet at the step barrier, so the planes two tiles share are
    // requested within a fraction of a step of each other -- L2 hits by construction, not by luck of the dispatch order
    const int lp = (G == 1) ? (int)threadIdx.x : (int)(threadIdx.x % (64 * NW));
    const int grp = (G == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / (64 * NW)));
    // XCD b & 7 owns a contiguous range of tiles (all their segments): tiles that share raw planes share an L2
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8);
    const uint32_t logical = first + (b >> 3);
    const int tile = (int)(logical / (uint32_t)a.nseg) * G + grp;
    const int seg = (int)(logical % (uint32_t)a.nseg);

    constexpr int n0 = 256 * NW, h0 = n0 >> 1;
    const int n1 = a.n1, n2 = a.n2, h1 = n1 >> 1, h2 = n2 >> 1;
    constexpr int rows1 = n0 + 16;                                  // exchange rows per plane (T2 each): the line + a copy of its first rows
    T2 *const x1 = reinterpret_cast<T2 *>(smem_raw) + grp * (2 * 4 * rows1);      // [G][2][4][rows1]

    const int ko = 2 * lp;
    int kod = ko + 4;  if (kod >= h0) kod -= h0;
    const bool odd = (lp & 1) != 0;
    const int j0 = seg * a.TJ;
    const int S = a.TJ >> 1;                                    // steps (multiple of U)
    const int kbase = j0 >> 1;

    // raw planes of the tile: element offsets from the box origin (wave-uniform, 32-bit: the launcher requires a box of < 2^32
    // elements); column + plane make a 64-bit scalar base, the lane adds its row offset (global_load ... v_offset, s[base:base+1])
    uint32_t poff[KR];                                          // (elements)
#pragma unroll
    for (int m = 0; m < KR; ++m) {
        int p = 4 * tile + m;
        if (p >= n2) p -= n2;
        poff[m] = (uint32_t)((int64_t)p * a.c2);
    }
    const uint32_t rowb = 16u * (uint32_t)lp;
    const uint32_t vo_s = 4u * (uint32_t)(odd ? ko - 2 : ko), vo_d = 4u * (uint32_t)(h0 + (odd ? kod - 2 : kod));
    auto colptr = [&](const int c) __attribute__((always_inline)) {
        int jc = j0 + c;
        if (jc >= n1) jc -= n1;
        return a.src + (int64_t)jc * a.c1;
    };

    // output planes: z = 0, 1 scaling planes 2 tile + z;  z = 2, 3 detail planes h2 + (2 tile + z - 2 + SH) mod h2
    T *yb[4];
    T *lb[4];
    int64_t ldl[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        int pz = 2 * tile + (z & 1);
        if (z >= 2) { pz += SH; if (pz >= h2) pz -= h2; pz += h2; }
        yb[z] = a.y + (int64_t)pz * a.y2;
        const bool to_ll = (a.ll != nullptr) && z < 2;
        lb[z] = to_ll ? (a.ll + (int64_t)pz * h0 * h1) : yb[z];
        ldl[z] = to_ll ? (int64_t)h0 : a.y1;
    }

    T4 L[D];
    T4 ring[RS][4];
#pragma unroll
    for (int c = 0; c < PD; ++c) {
        const T *const cp = colptr(c);
#pragma unroll
        for (int m = 0; m < KR; ++m) gload16_s<WL_P_3D1_LD != 0>(L[c * KR + m], cp + poff[m], rowb);
    }

    // One column through the dim-3 pass: plane m of the landing ring is folded into the tile's four sums the moment it has landed
    // (D - 1 younger loads are behind it), and its register goes back out for the same plane of column c + PD.
    auto column = [&](const int c, const int slot, const int lbase, const bool prefetch) __attribute__((always_inline)) {
        T4 s0 = T4{0.f, 0.f, 0.f, 0.f}, s1 = s0, d0 = s0, d1 = s0;
        const T *const nxt = colptr(c + PD);
#pragma unroll
        for (int m = 0; m < KR; ++m) {
            if (prefetch) wait_vm1<D - 1>(L[lbase + m]);
            else wait_vm1<0>(L[lbase + m]);                      // (last columns of the segment: nothing left to overlap)
            const T4 x = L[lbase + m];
            if (m == 0) { s0 = smul(a.tp.h[0], x); d0 = smul(gq(F - 1), x); }
            else if (m < F) { s0 = smad(s0, a.tp.h[m], x); d0 = smad(d0, gq(F - 1 - m), x); }
            if (m == 2) { s1 = smul(a.tp.h[0], x); d1 = smul(gq(F - 1), x); }
            else if (m > 2) { s1 = smad(s1, a.tp.h[m - 2], x); d1 = smad(d1, gq(F + 1 - m), x); }
            if (prefetch) gload16_s_tied<WL_P_3D1_LD != 0>(L[lbase + m], nxt + poff[m], rowb, s0, s1, d0, d1);
            __builtin_amdgcn_sched_barrier(0);
        }
        ring[slot][0] = s0; ring[slot][1] = s1; ring[slot][2] = d0; ring[slot][3] = d1;
    };

#pragma unroll
    for (int c = 0; c < F - 2; ++c) column(c, c, (c % PD) * KR, true);

    auto step = [&](const int t, const int u, const bool pfa, const bool pfb) __attribute__((always_inline)) {
        column(2 * t + F - 2, (2 * u + F - 2) % RS, 0, pfa);
        column(2 * t + F - 1, (2 * u + F - 1) % RS, (PD == 2) ? KR : 0, pfb);
        // ---- dim-2 pass on the column ring, per output plane; {A, B}[r] = scaling / detail (column k / kd) of row r ----
        T2 *const wbuf = x1 + (u & 1) * 4 * rows1;                 // (t and u have the same parity: groups of U = 4 steps)
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            T4 sa = smul(a.tp.h[0], ring[(2 * u) % RS][z]);
            T4 da = smul(gq(F - 1), ring[(2 * u) % RS][z]);
#pragma unroll
            for (int m = 1; m < F; ++m) {
                const T4 xm = ring[(2 * u + m) % RS][z];
                sa = smad(sa, a.tp.h[m], xm);
                da = smad(da, gq(F - 1 - m), xm);
            }
            T2 *const w1 = wbuf + z * rows1;
            const T4 v0 = T4{sa.x, da.x, sa.y, da.y}, v1 = T4{sa.z, da.z, sa.w, da.w};
            *reinterpret_cast<T4 *>(w1 + 4 * lp) = v0;
            *reinterpret_cast<T4 *>(w1 + 4 * lp + 2) = v1;
            if (lp < 4) {                                        // the line's first rows again behind its end: the top windows wrap
                *reinterpret_cast<T4 *>(w1 + n0 + 4 * lp) = v0;
                *reinterpret_cast<T4 *>(w1 + n0 + 4 * lp + 2) = v1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wg_lds_sync(multi);
        __builtin_amdgcn_sched_barrier(0);
        const int k = kbase + t;
        int kd = k + SH;
        if (kd >= h1) kd -= h1;
        // ---- dim-1 pass: window rows 4L' .. 4L'+11 as {A, B} pairs ----
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const T2 *const w1 = wbuf + z * rows1;
            T2 E[12];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const T4 v = *reinterpret_cast<const T4 *>(w1 + 4 * lp + 2 * c);
                E[2 * c] = T2{v.x, v.y};
                E[2 * c + 1] = T2{v.z, v.w};
            }
            T2 P[2], Q[2];                                 // P[q] = {ss, sd} of row ko + q;  Q[q] = {ds, dd} of row kod + q
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                T2 s = smul(a.tp.h[0], E[2 * q]);
#pragma unroll
                for (int m = 1; m < F; ++m) s = smad(s, a.tp.h[m], E[2 * q + m]);
                T2 d = smul(gq(F - 1), E[2 * q + 10 - F]);
#pragma unroll
                for (int m = F - 2; m >= 0; --m) d = smad(d, gq(m), E[2 * q + 9 - m]);
                P[q] = s;
                Q[q] = d;
            }
            T rP[2], rQ[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rP[q] = from_partner(odd ? P[q].x : P[q].y);
                rQ[q] = from_partner(odd ? Q[q].x : Q[q].y);
            }
            // even lane: column k (dim-2 scaling), rows ko..ko+3 and h0+kod..;  odd lane: column h1 + kd (dim-2 detail)
            // (both lanes of a pair issue the same two store instructions: the column base is a select of two scalars per pair parity,
            //  kept out of the vector registers: vo_s / vo_d are this lane's two row offsets for the whole march)
            T *const ck = yb[z] + (int64_t)k * a.y1, *const ckd = yb[z] + (int64_t)(h1 + kd) * a.y1, *const cl = lb[z] + (int64_t)k * ldl[z];
            const T4 vs = odd ? T4{rP[0], rP[1], P[0].y, P[1].y} : T4{P[0].x, P[1].x, rP[0], rP[1]};
            const T4 vd = odd ? T4{rQ[0], rQ[1], Q[0].y, Q[1].y} : T4{Q[0].x, Q[1].x, rQ[0], rQ[1]};
            if (!odd) {
                if (z < 2) gstore16_s<WL_P_3D1_LL>(cl, vo_s, vs);
                else gstore16_s<WL_P_3D1_ST>(cl, vo_s, vs);
                gstore16_s<WL_P_3D1_ST>(ck, vo_d, vd);
            } else {
                gstore16_s<WL_P_3D1_ST>(ckd, vo_s, vs);
                gstore16_s<WL_P_3D1_ST>(ckd, vo_d, vd);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int t0 = 0;
    for (; t0 < S - U; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u, true, true);
    }
    // the last PD columns of the segment have no successor to request
#pragma unroll
    for (int u = 0; u < U; ++u) step(t0 + u, u, (u < U - 1) || PD == 1, u < U - 1);
}

bool fwd3d_one_ok(int F, const float *cur, int64_t c1, int64_t c2, const float *y, int64_t y1, int64_t y2, const float *ll, const int64_t n[3])
{
    if (opt("WL_3D_ONE", 1) == 0) return false;
    if (F < 2 || F > 8 || (F & 1)) return false;
    const int64_t n0 = n[0], n1 = n[1], n2 = n[2];
    if (n0 != 256 && n0 != 512 && n0 != 1024) return false;
    if (n1 < 16 || (n1 % 16) != 0 || n1 > (1 << 20) || n2 < 16 || (n2 % 4) != 0 || n2 > (1 << 20)) return false;
    if ((c1 % 4) != 0 || (c2 % 4) != 0 || (y1 % 4) != 0 || (y2 % 4) != 0 || c1 < n0 || y1 < n0) return false;
    if (((uintptr_t)cur & 15) != 0 || ((uintptr_t)y & 15) != 0 || (ll && ((uintptr_t)ll & 15) != 0)) return false;
    if ((uint64_t)c2 * (uint64_t)n2 >= ((uint64_t)1 << 32)) return false;              // (32-bit plane offsets inside the box)
    if (cur == y) return false;                                  // (the level reads its input while its output is being written)
    if (n0 * n1 * n2 < opt("WL_3D_ONE_MIN", (long long)1 << 24)) return false;
    return true;
}

// (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is sticky per (function, device): once)
template <int F, int PD, int NW, int G>
static hipError_t launch_fwd3d_inst(hipStream_t st, unsigned nwg, size_t shmem, const Fwd3DArgs<F> &a)
{
    static thread_local int attr_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool done = false;
    for (int i = 0; i < 8; ++i) done = done || attr_dev[i] == dev;
    if (!done && shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd3d_one<F, PD, NW, G>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        for (int i = 0; i < 8; ++i) if (attr_dev[i] < 0) { attr_dev[i] = dev; break; }
    }
    hipLaunchKernelGGL((k_fwd3d_one<F, PD, NW, G>), dim3(nwg), dim3(64 * NW * G), shmem, st, a);
    return hipGetLastError();
}

template <int F>
static hipError_t launch_fwd3d_f(hipStream_t st, const Taps<float> &taps, const float *cur, int64_t c1, int64_t c2, float *y, int64_t y1,
                                 int64_t y2, float *ll, const int64_t n[3], int cu_count)
{
    Fwd3DArgs<F> a;
    a.src = cur; a.c1 = c1; a.c2 = c2; a.y = y; a.y1 = y1; a.y2 = y2; a.ll = ll;
    a.n0 = (int)n[0]; a.n1 = (int)n[1]; a.n2 = (int)n[2];
    const int W = a.n0 / 256;
    a.ntile = a.n2 / 4;
    int TJ = (int)opt("WL_3D_ONE_TJ", 64);
    if (TJ < 8 || (TJ % 8) != 0) TJ = 64;
    while (TJ > 8 && ((a.n1 % TJ) != 0 || (int64_t)a.ntile * (a.n1 / TJ) * W < (int64_t)cu_count * opt("WL_3D_ONE_WAVES", 8))) TJ >>= 1;
    if ((a.n1 % TJ) != 0) return hipErrorInvalidValue;
    a.TJ = TJ;
    a.nseg = a.n1 / TJ;
    a.tp = shrink<float, F>(taps);
    int Gw = (int)opt("WL_3D_ONE_G", 1);
    if ((Gw != 2 && Gw != 4) || W * Gw > 8 || (a.ntile % Gw) != 0) Gw = 1;
    const unsigned nwg = (unsigned)((a.ntile / Gw) * a.nseg);
    const size_t shmem = (size_t)Gw * 2 * 4 * (a.n0 + 16) * 8;
    const bool pd2 = opt("WL_3D_ONE_PD", 1) == 2;
#define WL_L3(PD_, NW_, G_) return launch_fwd3d_inst<F, PD_, NW_, G_>(st, nwg, shmem, a)
    if (pd2 && Gw == 1) { if (W == 1) WL_L3(2, 1, 1); else if (W == 2) WL_L3(2, 2, 1); else WL_L3(2, 4, 1); }
    else if (Gw == 1) { if (W == 1) WL_L3(1, 1, 1); else if (W == 2) WL_L3(1, 2, 1); else WL_L3(1, 4, 1); }
    else if (Gw == 2) { if (W == 1) WL_L3(1, 1, 2); else if (W == 2) WL_L3(1, 2, 2); else WL_L3(1, 4, 2); }
    else { if (W == 1) WL_L3(1, 1, 4); else WL_L3(1, 2, 4); }
#undef WL_L3
}

hipError_t fwd3d_one_launch(hipStream_t st, const Taps<float> &taps, const float *cur, int64_t c1, int64_t c2, float *y, int64_t y1, int64_t y2,
                            float *ll, const int64_t n[3], int cu_count)
{
    switch (taps.F) {
    case 2: return launch_fwd3d_f<2>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 4: return launch_fwd3d_f<4>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 6: return launch_fwd3d_f<6>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 8: return launch_fwd3d_f<8>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
