// wl_fwd3d.hip -- one forward 3-D filter-bank level in ONE pass over HBM (both element types, even F <= 8 -- 10 taps in Float32 --, lines of 32 ... 1024).
//
//   k_fwd3d_one<T, RPL, F, NW>    reference: planes -> rows -> columns of one level, transforms_filter.jl:246-263
//
// The three-launch / two-launch levels of wl_axis.hip read and write the box twice (dim 3 into a scratch box, then dims 2 + 1 per
// plane).  Here the level-l box is read once and the level's coefficients are written once:
//
//   * a workgroup owns WHOLE dim-1 lines (n0 = 64 RPL NW rows: NW waves, RPL rows per lane -- 4 Float32 or 2 Float64 rows, every
//     global access a 16-byte vector; 2 Float32 rows (8 bytes) for lines of 128 -- and the dim-1 window of the topmost lanes wraps
//     inside the workgroup's own LDS exchange: no halo rows, no helper wave),
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

template <typename T, int F>
struct Fwd3DArgs {
    const T *src; int64_t c1, c2;          // level-l box, strides 1, c1, c2
    T *y; int64_t y1, y2;                  // full array, strides 1, y1, y2
    T *ll;                                 // approximation corner: dense (h0, h1, h2), or nullptr = into y
    int n0, n1, n2;
    int TJ;                                // owned dim-2 columns per segment (multiple of 8)
    int nseg, ntile;
    TapsF<T, F> tp;
};

// element-wise forms (this file is built with -fno-slp-vectorize: scalar v_mul / v_add take a tap straight from its SGPR, the packed
// Float32 forms wanted the taps duplicated into aligned SGPR pairs and their operands in aligned VGPR pairs)
template <typename T, int N> struct Vx { typedef T type __attribute__((ext_vector_type(N))); };
template <typename T, typename V, int N>
__device__ __forceinline__ V smul(T h, const V x)
{
    V o;
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = h * x[i];
    return o;
}
template <typename T, typename V, int N>
__device__ __forceinline__ V smad(const V acc, T h, const V x)
{
    V o;
#pragma unroll
    for (int i = 0; i < N; ++i) o[i] = acc[i] + h * x[i];
    return o;
}

template <int N, typename V>
__device__ __forceinline__ void wait_vm1(V &a)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}

// 16 (8) bytes per lane from (64-bit scalar base) + (32-bit unsigned per-lane byte offset)
template <bool NT, typename V, typename T>
__device__ __forceinline__ void gload_s(V &dst, const T *sbase, uint32_t voff)
{
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "global_load_dwordx4 / x2");
    if constexpr (sizeof(V) == 16) {
        if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    } else {
        if constexpr (NT) asm volatile("global_load_dwordx2 %0, %1, %2 nt" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
        else asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
    }
}

// POL: 0 plain, 1 non-temporal, 2 write-through (store_pol, wl_dev.h)
template <int POL, typename V, typename T>
__device__ __forceinline__ void gstore_s(T *sbase, uint32_t voff, const V v)
{
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "global_store_dwordx4 / x2");
    if constexpr (sizeof(V) == 16) {
        if constexpr (POL == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
        else if constexpr (POL == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    } else {
        if constexpr (POL == 1) asm volatile("global_store_dwordx2 %0, %1, %2 nt\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
        else asm volatile("global_store_dwordx2 %0, %1, %2\n\ts_nop 1" :: "v"(voff), "v"(v), "s"(sbase) : "memory");
    }
}

// The same load tied to the four running sums ("+v"): everything that feeds them -- every product of the plane that just left `dst` --
// is complete before the load is issued, so the register allocator can give the new load the old plane's registers (without the tie
// hipcc sank the products below the following loads and the landing ring took twice its registers: spills, and a compiler-placed
// vmcnt(0) per scratch reload)
template <bool NT, typename V, typename T>
__device__ __forceinline__ void gload_s_tied(V &dst, const T *sbase, uint32_t voff, V &t0, V &t1, V &t2, V &t3)
{
    static_assert(sizeof(V) == 16 || sizeof(V) == 8, "global_load_dwordx4 / x2");
    if constexpr (sizeof(V) == 16) {
        if constexpr (NT) asm volatile("global_load_dwordx4 %0, %5, %6 nt" : "=v"(dst), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(voff), "s"(sbase) : "memory");
        else asm volatile("global_load_dwordx4 %0, %5, %6" : "=v"(dst), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(voff), "s"(sbase) : "memory");
    } else {
        if constexpr (NT) asm volatile("global_load_dwordx2 %0, %5, %6 nt" : "=v"(dst), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(voff), "s"(sbase) : "memory");
        else asm volatile("global_load_dwordx2 %0, %5, %6" : "=v"(dst), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(voff), "s"(sbase) : "memory");
    }
}

template <typename T, int RPL, int F, int NW>
__global__ void __launch_bounds__(64 * NW, 2) k_fwd3d_one(Fwd3DArgs<T, F> a)
{
    typedef typename Vx<T, 2>::type T2;                        // {scaling, detail} of one row in the exchange
    typedef typename Vx<T, RPL>::type V;                       // the lane's RPL rows
    typedef typename Vx<T, 4>::type X4;                        // two exchange rows: one 16- / 32-byte LDS access
    constexpr int SH = (F - 2) / 2, KR = F + 2, RS = (F <= 8) ? 8 : 10, U = RS / 2, D = KR;      // (10 taps: a 10-slot ring, groups of 5 steps)
    constexpr int NQ = RPL / 2, NE = 10 + 2 * (NQ - 1);        // scaling (and detail) rows a lane produces; its dim-1 window in rows
    static_assert(F >= 2 && F <= 10 && (F % 2) == 0, "column ring of 8 (10) slots");
    static_assert(F <= 8 || (sizeof(T) == 4 && RPL == 2), "10 taps: Float32 on 8-byte lanes only (ring of 10 x 4 planes)");
    static_assert(RPL == 2 || RPL == 4, "two or four rows per lane");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    constexpr bool multi = NW > 1;
    const int lp = (int)threadIdx.x;
    // XCD b & 7 owns a contiguous range of tiles (all their segments): tiles that share raw planes share an L2
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8);
    const uint32_t logical = first + (b >> 3);
    const int tile = (int)(logical / (uint32_t)a.nseg);
    const int seg = (int)(logical % (uint32_t)a.nseg);
    // dim 2 / dim 3 extents that are not multiples of the segment / of 4: the last segment (tile) is moved back to end at the edge and
    // recomputes a few columns (one plane pair) of its neighbour -- the same values to the same addresses
    const int p0 = (4 * tile + 4 <= a.n2) ? 4 * tile : a.n2 - 4;

    // the line: n0 <= 64 RPL NW rows, a multiple of 2 RPL; lanes past its end (lines that do not fill the last wave: 200, 240, 320 ...)
    // load row 0, publish and store nothing
    const int n0 = a.n0, h0 = n0 >> 1;
    const int n1 = a.n1, n2 = a.n2, h1 = n1 >> 1, h2 = n2 >> 1;
    constexpr int rows1 = 64 * RPL * NW + 16;                   // exchange rows per plane (T2 each): the line + a copy of its first 8 rows
    const bool active = RPL * lp < n0;
    T2 *const x1 = reinterpret_cast<T2 *>(smem_raw);            // [2][4][rows1]

    const int ko = NQ * lp;
    int kod = ko + 4;  if (kod >= h0) kod -= h0;
    const bool odd = (lp & 1) != 0;
    const int j0 = (seg * a.TJ + a.TJ <= a.n1) ? seg * a.TJ : a.n1 - a.TJ;
    const int S = a.TJ >> 1;                                    // steps (multiple of U)
    const int kbase = j0 >> 1;

    // raw planes of the tile: element offsets from the box origin (wave-uniform, 32-bit: the launcher requires a box of < 2^32
    // elements); column + plane make a 64-bit scalar base, the lane adds its row offset (global_load ... v_offset, s[base:base+1])
    uint32_t poff[KR];                                          // (elements)
#pragma unroll
    for (int m = 0; m < KR; ++m) {
        int p = p0 + m;
        if (p >= n2) p -= n2;
        poff[m] = (uint32_t)((int64_t)p * a.c2);
    }
    const uint32_t rowb = active ? (uint32_t)sizeof(V) * (uint32_t)lp : 0u;
    const uint32_t vo_s = (uint32_t)sizeof(T) * (uint32_t)(odd ? ko - NQ : ko), vo_d = (uint32_t)sizeof(T) * (uint32_t)(h0 + (odd ? kod - NQ : kod));
    auto colptr = [&](const int c) __attribute__((always_inline)) {
        int jc = j0 + c;
        if (jc >= n1) jc -= n1;
        return a.src + (int64_t)jc * a.c1;
    };

    // output planes: z = 0, 1 scaling planes p0 / 2 + z;  z = 2, 3 detail planes h2 + (p0 / 2 + z - 2 + SH) mod h2
    T *yb[4];
    T *lb[4];
    int64_t ldl[4];
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        int pz = (p0 >> 1) + (z & 1);
        if (z >= 2) { pz += SH; if (pz >= h2) pz -= h2; pz += h2; }
        yb[z] = a.y + (int64_t)pz * a.y2;
        const bool to_ll = (a.ll != nullptr) && z < 2;
        lb[z] = to_ll ? (a.ll + (int64_t)pz * h0 * h1) : yb[z];
        ldl[z] = to_ll ? (int64_t)h0 : a.y1;
    }

    V L[D];
    V ring[RS][4];
    {
        const T *const cp = colptr(0);
#pragma unroll
        for (int m = 0; m < KR; ++m) gload_s<WL_P_3D1_LD != 0>(L[m], cp + poff[m], rowb);
    }

    // One column through the dim-3 pass: plane m of the landing ring is folded into the tile's four sums the moment it has landed
    // (D - 1 younger loads are behind it), and its register goes back out for the same plane of the next column.
    auto column = [&](const int c, const int slot, const bool prefetch) __attribute__((always_inline)) {
        V s0, s1, d0, d1;
#pragma unroll
        for (int i = 0; i < RPL; ++i) { s0[i] = (T)0; s1[i] = (T)0; d0[i] = (T)0; d1[i] = (T)0; }
        const T *const nxt = colptr(c + 1);
#pragma unroll
        for (int m = 0; m < KR; ++m) {
            if (prefetch) wait_vm1<D - 1>(L[m]);
            else wait_vm1<0>(L[m]);                              // (last column of the segment: nothing left to overlap)
            const V x = L[m];
            if (m == 0) { s0 = smul<T, V, RPL>(a.tp.h[0], x); d0 = smul<T, V, RPL>(gq(F - 1), x); }
            else if (m < F) { s0 = smad<T, V, RPL>(s0, a.tp.h[m], x); d0 = smad<T, V, RPL>(d0, gq(F - 1 - m), x); }
            if (m == 2) { s1 = smul<T, V, RPL>(a.tp.h[0], x); d1 = smul<T, V, RPL>(gq(F - 1), x); }
            else if (m > 2) { s1 = smad<T, V, RPL>(s1, a.tp.h[m - 2], x); d1 = smad<T, V, RPL>(d1, gq(F + 1 - m), x); }
            if (prefetch) gload_s_tied<WL_P_3D1_LD != 0>(L[m], nxt + poff[m], rowb, s0, s1, d0, d1);
            __builtin_amdgcn_sched_barrier(0);
        }
        ring[slot][0] = s0; ring[slot][1] = s1; ring[slot][2] = d0; ring[slot][3] = d1;
    };

#pragma unroll
    for (int c = 0; c < F - 2; ++c) column(c, c, true);

    auto step = [&](const int t, const int u, const bool pfb) __attribute__((always_inline)) {
        column(2 * t + F - 2, (2 * u + F - 2) % RS, true);
        column(2 * t + F - 1, (2 * u + F - 1) % RS, pfb);
        // ---- dim-2 pass on the column ring, per output plane; {A, B}[r] = scaling / detail (column k / kd) of row r ----
        T2 *const wbuf = x1 + (u & 1) * 4 * rows1;                 // (t and u have the same parity: groups of U = 4 steps)
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            V sa = smul<T, V, RPL>(a.tp.h[0], ring[(2 * u) % RS][z]);
            V da = smul<T, V, RPL>(gq(F - 1), ring[(2 * u) % RS][z]);
#pragma unroll
            for (int m = 1; m < F; ++m) {
                const V xm = ring[(2 * u + m) % RS][z];
                sa = smad<T, V, RPL>(sa, a.tp.h[m], xm);
                da = smad<T, V, RPL>(da, gq(F - 1 - m), xm);
            }
            T2 *const w1 = wbuf + z * rows1;
#pragma unroll
            for (int r = 0; r < RPL; r += 2) {
                const X4 v = X4{sa[r], da[r], sa[r + 1], da[r + 1]};
                if (active) *reinterpret_cast<X4 *>(w1 + RPL * lp + r) = v;
                if (RPL * lp < 8) *reinterpret_cast<X4 *>(w1 + n0 + RPL * lp + r) = v;   // the line's first rows again behind its end: the top windows wrap
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        wg_lds_sync(multi);
        __builtin_amdgcn_sched_barrier(0);
        const int k = kbase + t;
        int kd = k + SH;
        if (kd >= h1) kd -= h1;
        // ---- dim-1 pass: window rows RPL L' .. RPL L' + NE - 1 as {A, B} pairs ----
#pragma unroll
        for (int z = 0; z < 4; ++z) {
            const T2 *const w1 = wbuf + z * rows1;
            T2 E[NE];
#pragma unroll
            for (int c = 0; c < NE / 2; ++c) {
                const X4 v = *reinterpret_cast<const X4 *>(w1 + RPL * lp + 2 * c);
                E[2 * c] = T2{v.x, v.y};
                E[2 * c + 1] = T2{v.z, v.w};
            }
            T2 P[NQ], Q[NQ];                               // P[q] = {ss, sd} of row ko + q;  Q[q] = {ds, dd} of row kod + q
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                T2 s = smul<T, T2, 2>(a.tp.h[0], E[2 * q]);
#pragma unroll
                for (int m = 1; m < F; ++m) s = smad<T, T2, 2>(s, a.tp.h[m], E[2 * q + m]);
                T2 d = smul<T, T2, 2>(gq(F - 1), E[2 * q + 10 - F]);
#pragma unroll
                for (int m = F - 2; m >= 0; --m) d = smad<T, T2, 2>(d, gq(m), E[2 * q + 9 - m]);
                P[q] = s;
                Q[q] = d;
            }
            // even lane: column k (dim-2 scaling): rows ko .. ko + RPL - 1 and h0 + kod ..;  odd lane: column h1 + kd (dim-2 detail).
            // Both lanes of a pair issue the same two store instructions: vo_s / vo_d are this lane's two row offsets for the whole
            // march, the column bases are scalars.
            V vs, vd;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const T rP = from_partner(odd ? P[q].x : P[q].y), rQ = from_partner(odd ? Q[q].x : Q[q].y);
                vs[q] = odd ? rP : P[q].x;  vs[NQ + q] = odd ? P[q].y : rP;
                vd[q] = odd ? rQ : Q[q].x;  vd[NQ + q] = odd ? Q[q].y : rQ;
            }
            T *const ck = yb[z] + (int64_t)k * a.y1, *const ckd = yb[z] + (int64_t)(h1 + kd) * a.y1, *const cl = lb[z] + (int64_t)k * ldl[z];
            if (active && !odd) {
                if (z < 2) gstore_s<WL_P_3D1_LL>(cl, vo_s, vs);
                else gstore_s<WL_P_3D1_ST>(cl, vo_s, vs);
                gstore_s<WL_P_3D1_ST>(ck, vo_d, vd);
            } else if (active) {
                gstore_s<WL_P_3D1_ST>(ckd, vo_s, vs);
                gstore_s<WL_P_3D1_ST>(ckd, vo_d, vd);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    int t0 = 0;
    for (; t0 < S - U; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u, true);
    }
    // the last column of the segment has no successor to request
#pragma unroll
    for (int u = 0; u < U; ++u) step(t0 + u, u, u < U - 1);
}

// rows per lane for a line of n0 (0: not a shape of this kernel): 16-byte lanes for lines beyond 128 Float32 rows, 8-byte Float32 lanes
// up to 128; the line has to be a multiple of 2 RPL rows (lane pairs store RPL consecutive rows of each half)
template <typename T>
static int fwd3d_rpl(int64_t n0)
{
    if (n0 < 32 || n0 > 1024) return 0;
    if (sizeof(T) == 4 && n0 > 128 && (n0 % 8) == 0) return 4;
    return (n0 % 4) == 0 ? 2 : 0;                               // (Float32 lines that are multiples of 4 only: 8-byte lanes, up to 8 waves)
}
// waves per workgroup: 1, 2, 4 (8: Float64 lines beyond 512) -- the smallest of them that holds the line
static int fwd3d_waves(int64_t n0, int rpl)
{
    int w = 1;
    while ((int64_t)64 * rpl * w < n0) w <<= 1;
    return w;
}

template <typename T>
bool fwd3d_one_ok(int F, const T *cur, int64_t c1, int64_t c2, const T *y, int64_t y1, int64_t y2, const T *ll, const int64_t n[3], bool any_tier)
{
    constexpr int VEC = 16 / (int)sizeof(T);
    if (opt("WL_3D_ONE", 1) == 0) return false;
    if (F < 2 || F > 10 || (F & 1)) return false;
    if (F == 10 && (sizeof(T) != 4 || (n[0] % 4) != 0 || opt("WL_3D_ONE_F10", 1) == 0)) return false;      // (10 taps: Float32 on 8-byte lanes)
    const int64_t n0 = n[0], n1 = n[1], n2 = n[2];
    if (fwd3d_rpl<T>(n0) == 0) return false;
    if (n1 < 16 || (n1 % 2) != 0 || n1 > (1 << 20) || n2 < 16 || (n2 % 2) != 0 || n2 > (1 << 20)) return false;
    if ((c1 % VEC) != 0 || (c2 % VEC) != 0 || (y1 % VEC) != 0 || (y2 % VEC) != 0 || c1 < n0 || y1 < n0) return false;
    if (((uintptr_t)cur & 15) != 0 || ((uintptr_t)y & 15) != 0 || (ll && ((uintptr_t)ll & 15) != 0)) return false;
    if ((uint64_t)c2 * (uint64_t)n2 >= ((uint64_t)1 << 32)) return false;              // (32-bit plane offsets inside the box)
    if (cur == y) return false;                                  // (the level reads its input while its output is being written)
    // (boxes below 2^21 elements: the three single-axis launches are as fast -- measured with half- / quarter-wave lines of 64 / 32 rows:
    //  128^3 full depth 52.8 against 51.8 us, Float64 63.7 against 56.5 -- so the line lengths stop at 128)
    // any_tier: not a shape of the axis / plane kernels -- the alternative is the three any-extent passes, and the gate drops to 2^19 + 1
    // (100^3 level 22.1 -> 14.0 us, 120^3 28.3 -> 14.9; 80^3 stays with the LDS blocks of wl_level3.hip)
    if (n0 * n1 * n2 < (any_tier ? opt("WL_3D_ONE_MIN_ANY", ((long long)1 << 19) + 1) : opt("WL_3D_ONE_MIN", (long long)1 << 21))) return false;
    return true;
}
template bool fwd3d_one_ok<float>(int, const float *, int64_t, int64_t, const float *, int64_t, int64_t, const float *, const int64_t[3], bool);
template bool fwd3d_one_ok<double>(int, const double *, int64_t, int64_t, const double *, int64_t, int64_t, const double *, const int64_t[3], bool);

// (hipFuncSetAttribute(MaxDynamicSharedMemorySize) is sticky per (function, device): once)
template <typename T, int RPL, int F, int NW>
static hipError_t launch_fwd3d_inst(hipStream_t st, unsigned nwg, const Fwd3DArgs<T, F> &a)
{
    const size_t shmem = (size_t)2 * 4 * (64 * RPL * NW + 16) * 2 * sizeof(T);
    static thread_local int attr_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool done = false;
    for (int i = 0; i < 8; ++i) done = done || attr_dev[i] == dev;
    if (!done && shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fwd3d_one<T, RPL, F, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        for (int i = 0; i < 8; ++i) if (attr_dev[i] < 0) { attr_dev[i] = dev; break; }
    }
    hipLaunchKernelGGL((k_fwd3d_one<T, RPL, F, NW>), dim3(nwg), dim3(64 * NW), shmem, st, a);
    return hipGetLastError();
}

template <typename T, int F>
static hipError_t launch_fwd3d_f(hipStream_t st, const Taps<T> &taps, const T *cur, int64_t c1, int64_t c2, T *y, int64_t y1,
                                 int64_t y2, T *ll, const int64_t n[3], int cu_count)
{
    Fwd3DArgs<T, F> a;
    a.src = cur; a.c1 = c1; a.c2 = c2; a.y = y; a.y1 = y1; a.y2 = y2; a.ll = ll;
    a.n0 = (int)n[0]; a.n1 = (int)n[1]; a.n2 = (int)n[2];
    constexpr int RS = (F <= 8) ? 8 : 10;
    const int rpl = (F == 10) ? 2 : fwd3d_rpl<T>(n[0]);
    if (rpl == 0) return hipErrorInvalidValue;
    const int W = fwd3d_waves(a.n0, rpl);
    a.ntile = (a.n2 + 3) / 4;
    // segment length: the largest multiple of 8 columns <= the requested one that leaves >= 8 waves per CU; a length that divides n1 is
    // preferred over a longer one that does not (the last segment of a non-dividing length recomputes columns of its neighbour)
    int TJ = (int)opt("WL_3D_ONE_TJ", 64);
    TJ = (TJ / RS) * RS;
    if (TJ < RS) TJ = RS;
    while (TJ > RS && (TJ > a.n1 || (int64_t)a.ntile * ((a.n1 + TJ - 1) / TJ) * W < (int64_t)cu_count * opt("WL_3D_ONE_WAVES", 8))) TJ -= RS;
    if (TJ > a.n1) return hipErrorInvalidValue;
    for (int t = TJ; t >= RS && t >= TJ - 2 * RS; t -= RS)
        if ((a.n1 % t) == 0) { TJ = t; break; }
    a.TJ = TJ;
    a.nseg = (a.n1 + TJ - 1) / TJ;
    a.tp = shrink<T, F>(taps);
    const unsigned nwg = (unsigned)(a.ntile * a.nseg);
    if constexpr (F == 10) {
        if constexpr (sizeof(T) == 4) {
            if (W == 1) return launch_fwd3d_inst<T, 2, F, 1>(st, nwg, a);
            if (W == 2) return launch_fwd3d_inst<T, 2, F, 2>(st, nwg, a);
            if (W == 4) return launch_fwd3d_inst<T, 2, F, 4>(st, nwg, a);
            return launch_fwd3d_inst<T, 2, F, 8>(st, nwg, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else if constexpr (sizeof(T) == 4) {
        if (rpl == 2 && W == 1) return launch_fwd3d_inst<T, 2, F, 1>(st, nwg, a);
        if (rpl == 2 && W == 2) return launch_fwd3d_inst<T, 2, F, 2>(st, nwg, a);
        if (rpl == 2 && W == 4) return launch_fwd3d_inst<T, 2, F, 4>(st, nwg, a);
        if (rpl == 2) return launch_fwd3d_inst<T, 2, F, 8>(st, nwg, a);
        if (W == 1) return launch_fwd3d_inst<T, 4, F, 1>(st, nwg, a);
        if (W == 2) return launch_fwd3d_inst<T, 4, F, 2>(st, nwg, a);
        return launch_fwd3d_inst<T, 4, F, 4>(st, nwg, a);
    } else {
        if (W == 1) return launch_fwd3d_inst<T, 2, F, 1>(st, nwg, a);
        if (W == 2) return launch_fwd3d_inst<T, 2, F, 2>(st, nwg, a);
        if (W == 4) return launch_fwd3d_inst<T, 2, F, 4>(st, nwg, a);
        return launch_fwd3d_inst<T, 2, F, 8>(st, nwg, a);
    }
}

template <typename T>
hipError_t fwd3d_one_launch(hipStream_t st, const Taps<T> &taps, const T *cur, int64_t c1, int64_t c2, T *y, int64_t y1, int64_t y2,
                            T *ll, const int64_t n[3], int cu_count)
{
    switch (taps.F) {
    case 2: return launch_fwd3d_f<T, 2>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 4: return launch_fwd3d_f<T, 4>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 6: return launch_fwd3d_f<T, 6>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 8: return launch_fwd3d_f<T, 8>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    case 10: return launch_fwd3d_f<T, 10>(st, taps, cur, c1, c2, y, y1, y2, ll, n, cu_count);
    default: return hipErrorInvalidValue;
    }
}
template hipError_t fwd3d_one_launch<float>(hipStream_t, const Taps<float> &, const float *, int64_t, int64_t, float *, int64_t, int64_t, float *,
                                            const int64_t[3], int);
template hipError_t fwd3d_one_launch<double>(hipStream_t, const Taps<double> &, const double *, int64_t, int64_t, double *, int64_t, int64_t, double *,
                                             const int64_t[3], int);

}  // namespace wl
