// wl_lift_tile.hip -- one 2-D lifting level of a cache-resident square block (128 ... 2048 rows) per launch, forward and inverse.
//
// The marching level kernels of wl_lift.hip (k_lift2d_fwd / k_lift2d_inv) need >= 16 column steps of a load -> cascade -> DPP ->
// store chain whatever the block size: 12-14 us per level from 1024^2 down to 128^2, four such levels per 8192^2 transform.
// Here a 256-thread workgroup owns a 64 x 64 piece of the block: the piece and the dependency cone of the scheme around it
// (HS = 4 samples on every side for cdf9/7 and db2, none for haar) are staged to LDS with every load in flight at once (periodic
// wrap resolved while staging), then each pass gives a thread a short open line segment -- 16 or 8 owned (s, d) pairs plus the
// cone -- which it runs through split -> steps -> normalize (or normalize -> steps -> merge) in registers as straight-line code;
// cone pairs are recomputed by the neighbouring segment instead of exchanged, so a pass costs two barriers and no step does.
// Rounding follows the reference: a pair whose operands do not wrap around the ends of the WHOLE line takes the in-bounds form
// x += (c1*a + c2*b), the others x += c1*a; x += c2*b (transforms_lifting.jl:437-483); which one depends on the global pair
// index only (a per-lane select, both forms share their products).
#include "wl_fast.h"
#include "wl_lift_shapes.h"
#include "wl_dev.h"

namespace wl {

namespace {

template <typename T>
struct LiftTileArgs {
    const T *src; int64_t lds;      // fw: block n x n                  inv: coefficient array
    T *y; int64_t ldy;              // fw: coefficient array            inv: result block n x n
    T *ll; int64_t ldl;             // fw: approximation destination or nullptr (-> y);  inv: approximation source or nullptr (-> src)
    int n;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <int ID>
struct TileGeom {
    static constexpr int HPE = (LiftReach<ID>::HP + 1) & ~1;      // cone in pairs, rounded to even: 16-byte aligned region rows
    static constexpr int HS = 2 * HPE;                            // ... in samples
    static constexpr int OWN = 64, OWNP = 32;                     // owned samples / pairs per dimension
    static constexpr int REG = OWN + 2 * HS, REGP = OWNP + 2 * HPE;
    static constexpr int LD = REG + 4;                            // (a multiple of 4, and 16-byte reads at a stride of LD hit distinct banks)
};

// all steps on an open segment of NP pairs whose first pair has the global (periodic) index kg0 of a line with `half` pairs;
// pairs closer than the scheme's reach to either end of the segment come out wrong and are never used
template <typename T, int ID, int NP, bool FAST>
__device__ __forceinline__ void seg_line_steps_v(T (&s)[NP], T (&d)[NP], const T (&c)[LIFT_FAST_STEPS][WL_MAX_NCOEF], const int kg0, const int half)
{
    typedef Shape<ID> SH;
#pragma unroll
    for (int k = 0; k < SH::NS; ++k) {
        const int upd = SH::S[k].upd, nc = SH::S[k].nc, sh = SH::S[k].sh;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int j0 = j - sh;
            if (j0 < 0 || j0 + nc - 1 > NP - 1) continue;
            int jg = kg0 + j;
            if (jg >= half) jg -= half;
            const bool inb = (jg - sh >= 0) && (jg - sh + nc - 1 <= half - 1);
            const T x = upd ? d[j] : s[j];
            const T o0 = upd ? s[j0] : d[j0];
            const T o1 = (nc > 1) ? (upd ? s[(j0 + 1 < NP) ? j0 + 1 : j0] : d[(j0 + 1 < NP) ? j0 + 1 : j0]) : (T)0;
            const T o2 = (nc > 2) ? (upd ? s[(j0 + 2 < NP) ? j0 + 2 : j0] : d[(j0 + 2 < NP) ? j0 + 2 : j0]) : (T)0;
            const T m0 = c[k][0] * o0;
            T acc = m0;
            T xb = x + m0;
            if (nc > 1) { const T m1 = c[k][1] * o1; acc = acc + m1; xb = xb + m1; }
            if (nc > 2) { const T m2 = c[k][2] * o2; acc = acc + m2; xb = xb + m2; }
            const T xin = x + acc;
            const T r = FAST ? xin : (inb ? xin : xb);
            if (upd) d[j] = r; else s[j] = r;
        }
    }
}
// kg0 is the same for every lane of a wave (the callers map a wave to one segment position): the position tests are scalar work,
// and a wave whose segment stays clear of the ends of the line skips the boundary form altogether
template <typename T, int ID, int NP>
__device__ __forceinline__ void seg_line_steps(T (&s)[NP], T (&d)[NP], const T (&c)[LIFT_FAST_STEPS][WL_MAX_NCOEF], const int kg0_, const int half)
{
    const int kg0 = __builtin_amdgcn_readfirstlane(kg0_);
    if (kg0 >= 4 && kg0 + NP + 4 <= half) seg_line_steps_v<T, ID, NP, true>(s, d, c, kg0, half);
    else seg_line_steps_v<T, ID, NP, false>(s, d, c, kg0, half);
}

__device__ __forceinline__ int wrap_into(int v, const int m)
{
    while (v < 0) v += m;
    while (v >= m) v -= m;
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward: dim-2 pass (lines along the columns) on every region row, then dim-1 pass on the owned columns, quadrants stored
// straight from registers
template <typename T, int ID>
__global__ void __launch_bounds__(256) k_lift2d_tile_fwd(LiftTileArgs<T> a)
{
    typedef TileGeom<ID> G;
    constexpr int VEC = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(VEC)));
    constexpr int HPE = G::HPE, HS = G::HS, REG = G::REG, LD = G::LD;
    __shared__ __attribute__((aligned(16))) T P[LD * REG];
    const int tid = threadIdx.x;
    const int n = a.n, h = n >> 1;
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    // ---- stage: P[r + c * LD] = x[(64 bx - HS + r) mod n, (64 by - HS + c) mod n], 16-byte chunks along the rows ----
    {
        constexpr int CPC = REG / VEC, NCH = CPC * REG, PER = (NCH + 255) / 256;
        V v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 256 * u;
            if (id < NCH) {
                const int col = id / CPC, cc = id - col * CPC;
                const int gi = wrap_into(64 * bx - HS + VEC * cc, n), gj = wrap_into(64 * by - HS + col, n);
                v[u] = *reinterpret_cast<const V *>(a.src + gi + (int64_t)gj * a.lds);
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 256 * u;
            if (id < NCH) {
                const int col = id / CPC, cc = id - col * CPC;
                *reinterpret_cast<V *>(P + VEC * cc + col * LD) = v[u];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- dim 2: task = (region row r, half of the owned column pairs); 16 owned pairs + the cone on either side ----
    {
        constexpr int SEG = 16, NP = SEG + 2 * HPE;
        // wave w: segment w & 1 of region rows 64 (w >> 1) ... (one segment position per wave)
        const int seg = (tid >> 6) & 1, r = ((tid >> 7) << 6) + (tid & 63);
        const bool active = r < REG;
        T s[NP], d[NP];
        if (active) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                s[q] = P[r + (2 * (SEG * seg + q)) * LD];
                d[q] = P[r + (2 * (SEG * seg + q) + 1) * LD];
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * by - HPE + SEG * seg, h), h);
        }
        lds_barrier();                                  // every segment has read its cone before anybody overwrites it
        if (active) {
#pragma unroll
            for (int q = HPE; q < HPE + SEG; ++q) {
                P[r + (2 * (SEG * seg + q)) * LD] = s[q] * a.norm1;
                P[r + (2 * (SEG * seg + q) + 1) * LD] = d[q] * a.norm2;
            }
        }
        lds_barrier();
    }
    // ---- dim 1: task = (owned column, quarter of the owned row pairs); rows are contiguous in LDS ----
    {
        constexpr int SEG = 8, NP = SEG + 2 * HPE;
        const int jc = tid & 63, seg = tid >> 6;
        const int c = HS + jc;                          // region column: even = scaling column of pair c/2, odd = detail column
        T v[2 * NP];
#pragma unroll
        for (int e = 0; e < 2 * NP / VEC; ++e) {
            const V t = *reinterpret_cast<const V *>(P + 2 * SEG * seg + VEC * e + c * LD);
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[VEC * e + i] = t[i];
        }
        T s[NP], d[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) { s[q] = v[2 * q]; d[q] = v[2 * q + 1]; }
        seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * bx - HPE + SEG * seg, h), h);
        T so[SEG], dO[SEG];
#pragma unroll
        for (int q = 0; q < SEG; ++q) { so[q] = s[HPE + q] * a.norm1; dO[q] = d[HPE + q] * a.norm2; }
        const int gk = 32 * bx + SEG * seg;             // first owned row pair of this task
        const int gkc = 32 * by + (jc >> 1);            // column pair
        const bool dcol = (jc & 1) != 0;
        T *const lo = (!dcol && a.ll) ? (a.ll + gk + (int64_t)gkc * a.ldl) : (a.y + gk + (int64_t)((dcol ? h : 0) + gkc) * a.ldy);
        T *const hi = a.y + h + gk + (int64_t)((dcol ? h : 0) + gkc) * a.ldy;
#pragma unroll
        for (int e = 0; e < SEG / VEC; ++e) {
            V t, u;
#pragma unroll
            for (int i = 0; i < VEC; ++i) { t[i] = so[VEC * e + i]; u[i] = dO[VEC * e + i]; }
            *reinterpret_cast<V *>(lo + VEC * e) = t;
            *reinterpret_cast<V *>(hi + VEC * e) = u;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward, TWO levels per launch (round 5): the tile above with the level-1 approximation kept in LDS.  The second level needs the
// 32 x 32 level-1 approximations of the tile plus its own cone (HS samples on every side), so level 1 runs on a region widened
// accordingly -- P1 = 32 + 2 HS "approximation-owned" pairs per dimension, 2 P1 + 2 HS staged samples (88 for cdf9/7) -- and stores
// the details of the 32 x 32 truly owned pairs only; its approximations go to a second LDS array on which level 2 is the same
// two passes at half the size.  Cones are recomputed, tiles are independent; the level-1 approximation never reaches memory.
// 512 threads: every pass maps one segment position to a wave (seg_line_steps takes the wave-uniform position tests).
template <int ID>
struct TileGeom2 {
    static constexpr int HPE = TileGeom<ID>::HPE, HS = TileGeom<ID>::HS;
    static constexpr int P1 = 32 + 2 * HS;              // level-1 pairs per dimension whose approximation level 2 needs
    static constexpr int REG1 = 2 * P1 + 2 * HS;        // staged samples per dimension
    static constexpr int LD1 = REG1 + 4;
    static constexpr int REG2 = P1;                     // level-2 region: 32 owned samples + HS on every side
    static constexpr int LD2 = REG2 + 4;
    static constexpr int ELEMS = LD1 * REG1 + LD2 * REG2 + 16;
};

template <typename T, int ID>
__global__ void __launch_bounds__(512) k_lift2d_tile2_fwd(LiftTileArgs<T> a)
{
    typedef TileGeom2<ID> G;
    constexpr int VEC = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(VEC)));
    constexpr int HPE = G::HPE, HS = G::HS, P1 = G::P1, REG1 = G::REG1, LD1 = G::LD1, REG2 = G::REG2, LD2 = G::LD2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);             // level-1 region, column-major, leading dimension LD1
    T *Q = P + LD1 * REG1;                              // level-1 approximations of the P1 x P1 pairs = the level-2 region
    const int tid = threadIdx.x;
    const int n = a.n, h = n >> 1, h2 = n >> 2;
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    // ---- stage: P[r + c * LD1] = x[(64 bx - 3 HS + r) mod n, (64 by - 3 HS + c) mod n] ----
    {
        constexpr int CPC = REG1 / VEC, NCH = CPC * REG1, PER = (NCH + 511) / 512;
        V v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 512 * u;
            if (id < NCH) {
                const int col = id / CPC, cc = id - col * CPC;
                const int gi = wrap_into(64 * bx - 3 * HS + VEC * cc, n), gj = wrap_into(64 * by - 3 * HS + col, n);
                v[u] = *reinterpret_cast<const V *>(a.src + gi + (int64_t)gj * a.lds);
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 512 * u;
            if (id < NCH) {
                const int col = id / CPC, cc = id - col * CPC;
                *reinterpret_cast<V *>(P + VEC * cc + col * LD1) = v[u];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ================= level 1, dim 2: task = (region row r, half of the P1 column pairs) =================
    {
        constexpr int SEG = P1 / 2, NP = SEG + 2 * HPE;
        const int seg = (tid >> 6) & 1, r = ((tid >> 7) << 6) + (tid & 63);          // waves 0..3 (rows 0..127), one segment per wave
        const bool active = (tid < 256) && r < REG1;
        T s[NP], d[NP];
        if (active) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                s[q] = P[r + (2 * (SEG * seg + q)) * LD1];
                d[q] = P[r + (2 * (SEG * seg + q) + 1) * LD1];
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * by - HS - HPE + SEG * seg, h), h);
        }
        lds_barrier();
        if (active) {
#pragma unroll
            for (int q = HPE; q < HPE + SEG; ++q) {
                P[r + (2 * (SEG * seg + q)) * LD1] = s[q] * a.norm1;
                P[r + (2 * (SEG * seg + q) + 1) * LD1] = d[q] * a.norm2;
            }
        }
        lds_barrier();
    }
    // ================= level 1, dim 1: task = (one of the 2 P1 columns, quarter of the P1 row pairs) =================
    {
        constexpr int SEG = P1 / 4, NP = SEG + 2 * HPE;
        static_assert((P1 % 4) == 0 && ((2 * SEG) % VEC) == 0 && ((2 * NP) % VEC) == 0, "segments must stay 16-byte aligned");
        const int seg = tid >> 7, jc = tid & 127;                                    // waves 2 seg, 2 seg + 1: columns 0..127 of that segment
        if (jc < 2 * P1) {
            const int c = HS + jc;                      // region column: even = scaling column of pair jc / 2, odd = detail column
            T v[2 * NP];
#pragma unroll
            for (int e = 0; e < 2 * NP / VEC; ++e) {
                const V t = *reinterpret_cast<const V *>(P + 2 * SEG * seg + VEC * e + c * LD1);
#pragma unroll
                for (int i = 0; i < VEC; ++i) v[VEC * e + i] = t[i];
            }
            T s[NP], d[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) { s[q] = v[2 * q]; d[q] = v[2 * q + 1]; }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * bx - HS - HPE + SEG * seg, h), h);
            const bool dcol = (jc & 1) != 0;
            const int pc = (jc >> 1) - HS;              // column pair relative to the tile's owned pairs (owned: 0 .. 31)
            const bool cown = pc >= 0 && pc < 32;
            const int gkc = 32 * by + pc;               // global column pair (valid where cown)
#pragma unroll
            for (int q = 0; q < SEG; ++q) {
                const T so = s[HPE + q] * a.norm1, dO = d[HPE + q] * a.norm2;
                const int pr = SEG * seg + q - HS;      // row pair relative to the owned ones
                if (!dcol) Q[(SEG * seg + q) + (jc >> 1) * LD2] = so;                // approximation: level 2's input (all P1 x P1)
                if (cown && pr >= 0 && pr < 32) {
                    const int gk = 32 * bx + pr;
                    const int64_t col = (int64_t)((dcol ? h : 0) + gkc) * a.ldy;
                    if (dcol) a.y[gk + col] = so;                                    // s rows of a detail column
                    a.y[h + gk + col] = dO;                                          // d rows of either column
                }
            }
        }
    }
    lds_barrier();
    // ================= level 2 on Q (REG2 = 32 + 2 HS samples per dimension), as the one-level kernel at half the size =================
    {
        constexpr int SEG = 8, NP = SEG + 2 * HPE;
        // dim 2: task = (region row r < REG2, half of the 16 owned column pairs): waves 0, 1
        const int seg = (tid >> 6) & 1, r = tid & 63;
        const bool active = (tid < 128) && r < REG2;
        T s[NP], d[NP];
        if (active) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                s[q] = Q[r + (2 * (SEG * seg + q)) * LD2];
                d[q] = Q[r + (2 * (SEG * seg + q) + 1) * LD2];
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(16 * by - HPE + SEG * seg, h2), h2);
        }
        lds_barrier();
        if (active) {
#pragma unroll
            for (int q = HPE; q < HPE + SEG; ++q) {
                Q[r + (2 * (SEG * seg + q)) * LD2] = s[q] * a.norm1;
                Q[r + (2 * (SEG * seg + q) + 1) * LD2] = d[q] * a.norm2;
            }
        }
        lds_barrier();
    }
    {
        // dim 1: task = (owned column jc < 32, half of the 16 owned row pairs): waves 0, 1 (segment = wave)
        constexpr int SEG = 8, NP = SEG + 2 * HPE;
        const int seg = tid >> 6, jc = tid & 63;
        if (tid < 128 && jc < 32) {
            const int c = HS + jc;
            T s[NP], d[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) { s[q] = Q[2 * (SEG * seg + q) + c * LD2]; d[q] = Q[2 * (SEG * seg + q) + 1 + c * LD2]; }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(16 * bx - HPE + SEG * seg, h2), h2);
            const int gk = 16 * bx + SEG * seg;          // first owned level-2 row pair of this task
            const int gkc = 16 * by + (jc >> 1);
            const bool dcol = (jc & 1) != 0;
            T *const lo = (!dcol && a.ll) ? (a.ll + gk + (int64_t)gkc * a.ldl) : (a.y + gk + (int64_t)((dcol ? h2 : 0) + gkc) * a.ldy);
            T *const hi = a.y + h2 + gk + (int64_t)((dcol ? h2 : 0) + gkc) * a.ldy;
#pragma unroll
            for (int q = 0; q < SEG; ++q) {
                lo[q] = s[HPE + q] * a.norm1;
                hi[q] = d[HPE + q] * a.norm2;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// inverse: the four quadrant pieces staged de-interleaved (scaling rows | detail rows, scaling columns | detail columns), dim-1
// pass (normalize -> steps -> merge along the rows) on every region column, then dim-2 pass on the owned rows
template <typename T, int ID>
__global__ void __launch_bounds__(256) k_lift2d_tile_inv(LiftTileArgs<T> a)
{
    typedef TileGeom<ID> G;
    constexpr int VEC = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(VEC)));
    typedef T V2 __attribute__((ext_vector_type(2)));
    constexpr int HPE = G::HPE, HS = G::HS, REG = G::REG, REGP = G::REGP, LD = G::LD;
    __shared__ __attribute__((aligned(16))) T P[LD * REG];
    const int tid = threadIdx.x;
    const int n = a.n, h = n >> 1;
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    // ---- stage: column slot cs < REGP: scaling column pair cs, cs >= REGP: detail column pair cs - REGP; rows [0, REGP): scaling
    //      row pairs, [REGP, REG): detail row pairs.  Chunks of two rows (the cone is an even number of pairs). ----
    {
        constexpr int CPH = REGP / 2, NCH = 2 * CPH * REG, PER = (NCH + 255) / 256;
        V2 v[PER];
        const T *const lls = a.ll ? a.ll : a.src;
        const int64_t ldl = a.ll ? a.ldl : a.lds;
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 256 * u;
            if (id < NCH) {
                const int cs = id / (2 * CPH), rem = id - cs * (2 * CPH);
                const int rdet = (rem >= CPH) ? 1 : 0, cc = rem - rdet * CPH;
                const int cdet = (cs >= REGP) ? 1 : 0;
                const int gk = wrap_into(32 * bx - HPE + 2 * cc, h), gc = wrap_into(32 * by - HPE + (cs - cdet * REGP), h);
                const T *p = (!rdet && !cdet) ? (lls + gk + (int64_t)gc * ldl) : (a.src + (rdet ? h : 0) + gk + (int64_t)((cdet ? h : 0) + gc) * a.lds);
                v[u] = *reinterpret_cast<const V2 *>(p);
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 256 * u;
            if (id < NCH) {
                const int cs = id / (2 * CPH), rem = id - cs * (2 * CPH);
                *reinterpret_cast<V2 *>(P + 2 * rem + cs * LD) = v[u];       // (rem counts two-row chunks through scaling then detail rows)
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- dim 1: task = (column slot, half of the owned row pairs) ----
    {
        constexpr int SEG = 16, NP = SEG + 2 * HPE;
        const int seg = (tid >> 6) & 1, cs = ((tid >> 7) << 6) + (tid & 63);      // (one segment position per wave)
        const bool active = cs < REG;
        T s[NP], d[NP];
        if (active) {
#pragma unroll
            for (int e = 0; e < NP / VEC; ++e) {
                const V t = *reinterpret_cast<const V *>(P + SEG * seg + VEC * e + cs * LD);
                const V u = *reinterpret_cast<const V *>(P + REGP + SEG * seg + VEC * e + cs * LD);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { s[VEC * e + i] = a.norm1 * t[i]; d[VEC * e + i] = a.norm2 * u[i]; }
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * bx - HPE + SEG * seg, h), h);
        }
        lds_barrier();
        if (active) {
            // merged rows of the owned pairs, interleaved: region row 2 (SEG seg + q), + 1
#pragma unroll
            for (int q = HPE; q < HPE + SEG; q += VEC / 2) {
                V t;
#pragma unroll
                for (int i = 0; i < VEC / 2; ++i) { t[2 * i] = s[q + i]; t[2 * i + 1] = d[q + i]; }
                *reinterpret_cast<V *>(P + 2 * (SEG * seg + q) + cs * LD) = t;
            }
        }
        lds_barrier();
    }
    // ---- dim 2: task = (owned row, quarter of the owned column pairs) ----
    {
        constexpr int SEG = 8, NP = SEG + 2 * HPE;
        const int ir = tid & 63, seg = tid >> 6;
        const int r = HS + ir;
        T s[NP], d[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            s[q] = a.norm1 * P[r + (SEG * seg + q) * LD];
            d[q] = a.norm2 * P[r + (REGP + SEG * seg + q) * LD];
        }
        seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * by - HPE + SEG * seg, h), h);
        T *const o = a.y + 64 * bx + ir + (int64_t)(2 * (32 * by + SEG * seg)) * a.ldy;
#pragma unroll
        for (int q = 0; q < SEG; ++q) {
            o[(int64_t)(2 * q) * a.ldy] = s[HPE + q];
            o[(int64_t)(2 * q + 1) * a.ldy] = d[HPE + q];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// inverse, TWO levels per launch (round 5): the reconstruction of the coarser level is produced in LDS exactly where the finer
// level's staging expects its approximation quadrant.  A tile owns 64 x 64 samples of the FINE output (n x n).  Fine level: REGP =
// 32 + 2 HPE pairs per dimension as above; their scaling-row / scaling-column block (REGP x REGP samples of the coarse output) is
// what the coarse level must deliver: OWc = REGP / 2 = 16 + HPE / 2 ... pairs per dimension plus its own cone, REGPc = OWc + 2 HPE staged
// pairs.  Coarse dim-1 / dim-2 passes, then the fine ones of the one-level kernel.  a.src: coefficient array (both levels' details),
// a.ll: approximation source of the COARSE level (or nullptr: the corner of a.src), a.y: the n x n result.
template <int ID>
struct TileGeomInv2 {
    static constexpr int HPE = TileGeom<ID>::HPE;
    static constexpr int REGP = TileGeom<ID>::REGP, REG = TileGeom<ID>::REG, LD = TileGeom<ID>::LD;
    static constexpr int OWC = REGP / 2;                // coarse output pairs per dimension the fine level needs
    static constexpr int REGPC = OWC + 2 * HPE;         // ... plus the coarse cone: staged coarse pairs
    static constexpr int REGC = 2 * REGPC, LDC = REGC + 4;
    static constexpr int ELEMS = LD * REG + LDC * REGC + 16;
};

template <typename T, int ID>
__global__ void __launch_bounds__(256) k_lift2d_tile2_inv(LiftTileArgs<T> a)
{
    typedef TileGeomInv2<ID> G;
    constexpr int VEC = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(VEC)));
    typedef T V2 __attribute__((ext_vector_type(2)));
    constexpr int HPE = G::HPE, HS = 2 * HPE, REG = G::REG, REGP = G::REGP, LD = G::LD;
    constexpr int OWC = G::OWC, REGPC = G::REGPC, REGC = G::REGC, LDC = G::LDC;
    static_assert((HPE % 2) == 0 && (OWC % 2) == 0, "the coarse region must start at a whole pair and split into two segments");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);             // fine staging (layout of k_lift2d_tile_inv)
    T *C = P + LD * REG;                                // coarse staging, same layout with REGPC
    const int tid = threadIdx.x;
    const int n = a.n, h = n >> 1, hc = n >> 2;
    const int bx = (int)blockIdx.x, by = (int)blockIdx.y;
    const int c0x = 16 * bx - HPE / 2 - HPE, c0y = 16 * by - HPE / 2 - HPE;      // global coarse pair of coarse region pair 0
    // ---- stage the coarse pieces (scalar: the region starts at an odd row where HPE / 2 is odd) ----
    {
        const T *const lls = a.ll ? a.ll : a.src;
        const int64_t ldl = a.ll ? a.ldl : a.lds;
        for (int id = tid; id < REGC * REGC; id += 256) {
            const int cs = id / REGC, rem = id - cs * REGC;
            const int rdet = (rem >= REGPC) ? 1 : 0, cdet = (cs >= REGPC) ? 1 : 0;
            const int gk = wrap_into(c0x + (rem - rdet * REGPC), hc), gc = wrap_into(c0y + (cs - cdet * REGPC), hc);
            const T *p = (!rdet && !cdet) ? (lls + gk + (int64_t)gc * ldl) : (a.src + (rdet ? hc : 0) + gk + (int64_t)((cdet ? hc : 0) + gc) * a.lds);
            C[rem + cs * LDC] = *p;
        }
    }
    // ---- stage the fine detail pieces (the scaling x scaling block comes from the coarse level below) ----
    {
        constexpr int CPH = REGP / 2, NCH = 2 * CPH * REG, PER = (NCH + 255) / 256;
        V2 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 256 * u;
            if (id < NCH) {
                const int cs = id / (2 * CPH), rem = id - cs * (2 * CPH);
                const int rdet = (rem >= CPH) ? 1 : 0, cc = rem - rdet * CPH;
                const int cdet = (cs >= REGP) ? 1 : 0;
                if (rdet || cdet) {
                    const int gk = wrap_into(32 * bx - HPE + 2 * cc, h), gc = wrap_into(32 * by - HPE + (cs - cdet * REGP), h);
                    v[u] = *reinterpret_cast<const V2 *>(a.src + (rdet ? h : 0) + gk + (int64_t)((cdet ? h : 0) + gc) * a.lds);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int id = tid + 256 * u;
            if (id < NCH) {
                const int cs = id / (2 * CPH), rem = id - cs * (2 * CPH);
                const int rdet = (rem >= CPH) ? 1 : 0, cdet = (cs >= REGP) ? 1 : 0;
                if (rdet || cdet) *reinterpret_cast<V2 *>(P + 2 * rem + cs * LD) = v[u];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ================= coarse level, dim 1: task = (column slot cs < REGC, half of the OWC output row pairs); waves 0, 1 =================
    {
        constexpr int SEG = OWC / 2, NP = SEG + 2 * HPE;
        const int seg = tid >> 6, cs = tid & 63;
        const bool active = (tid < 128) && cs < REGC;
        T s[NP], d[NP];
        if (active) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                s[q] = a.norm1 * C[SEG * seg + q + cs * LDC];
                d[q] = a.norm2 * C[REGPC + SEG * seg + q + cs * LDC];
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(c0x + SEG * seg, hc), hc);
        }
        lds_barrier();
        if (active) {
            // merged rows of the owned pairs, interleaved: coarse output row 2 (SEG seg + q - HPE), + 1   (rows 0 .. REGP - 1)
#pragma unroll
            for (int q = HPE; q < HPE + SEG; ++q) {
                C[2 * (SEG * seg + q - HPE) + cs * LDC] = s[q];
                C[2 * (SEG * seg + q - HPE) + 1 + cs * LDC] = d[q];
            }
        }
        lds_barrier();
    }
    // ================= coarse level, dim 2: task = (output row r < REGP, half of the OWC output column pairs); waves 0, 1 =================
    {
        constexpr int SEG = OWC / 2, NP = SEG + 2 * HPE;
        const int seg = tid >> 6, r = tid & 63;
        if (tid < 128 && r < REGP) {
            T s[NP], d[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                s[q] = a.norm1 * C[r + (SEG * seg + q) * LDC];
                d[q] = a.norm2 * C[r + (REGPC + SEG * seg + q) * LDC];
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(c0y + SEG * seg, hc), hc);
            // coarse output (row r, columns 2 (SEG seg + q - HPE), + 1) = the fine level's scaling x scaling block
#pragma unroll
            for (int q = HPE; q < HPE + SEG; ++q) {
                P[r + (2 * (SEG * seg + q - HPE)) * LD] = s[q];
                P[r + (2 * (SEG * seg + q - HPE) + 1) * LD] = d[q];
            }
        }
    }
    lds_barrier();
    // ================= fine level: the two passes of k_lift2d_tile_inv =================
    {
        constexpr int SEG = 16, NP = SEG + 2 * HPE;
        const int seg = (tid >> 6) & 1, cs = ((tid >> 7) << 6) + (tid & 63);      // (one segment position per wave)
        const bool active = cs < REG;
        T s[NP], d[NP];
        if (active) {
#pragma unroll
            for (int e = 0; e < NP / VEC; ++e) {
                const V t = *reinterpret_cast<const V *>(P + SEG * seg + VEC * e + cs * LD);
                const V u = *reinterpret_cast<const V *>(P + REGP + SEG * seg + VEC * e + cs * LD);
#pragma unroll
                for (int i = 0; i < VEC; ++i) { s[VEC * e + i] = a.norm1 * t[i]; d[VEC * e + i] = a.norm2 * u[i]; }
            }
            seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * bx - HPE + SEG * seg, h), h);
        }
        lds_barrier();
        if (active) {
#pragma unroll
            for (int q = HPE; q < HPE + SEG; q += VEC / 2) {
                V t;
#pragma unroll
                for (int i = 0; i < VEC / 2; ++i) { t[2 * i] = s[q + i]; t[2 * i + 1] = d[q + i]; }
                *reinterpret_cast<V *>(P + 2 * (SEG * seg + q) + cs * LD) = t;
            }
        }
        lds_barrier();
    }
    {
        constexpr int SEG = 8, NP = SEG + 2 * HPE;
        const int ir = tid & 63, seg = tid >> 6;
        const int r = HS + ir;
        T s[NP], d[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            s[q] = a.norm1 * P[r + (SEG * seg + q) * LD];
            d[q] = a.norm2 * P[r + (REGP + SEG * seg + q) * LD];
        }
        seg_line_steps<T, ID, NP>(s, d, a.c, wrap_into(32 * by - HPE + SEG * seg, h), h);
        T *const o = a.y + 64 * bx + ir + (int64_t)(2 * (32 * by + SEG * seg)) * a.ldy;
#pragma unroll
        for (int q = 0; q < SEG; ++q) {
            o[(int64_t)(2 * q) * a.ldy] = s[HPE + q];
            o[(int64_t)(2 * q + 1) * a.ldy] = d[HPE + q];
        }
    }
}

template <typename T, int ID, int FW>
hipError_t launch_tile_id(hipStream_t st, const LiftTileArgs<T> &a)
{
    const unsigned g = (unsigned)(a.n / 64);
    if (FW) hipLaunchKernelGGL((k_lift2d_tile_fwd<T, ID>), dim3(g, g), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((k_lift2d_tile_inv<T, ID>), dim3(g, g), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace

bool lift2d_tile_ok(int id, int64_t n) { return id >= 0 && id <= 5 && n >= 128 && n <= 16384 && (n % 64) == 0; }
// two forward levels per launch: the tile grid is that of the first level; the second level (n / 2) must still be a multiple of 32
bool lift2d_tile2_ok(int id, int64_t n) { return (id == 0 || id == 2 || id == 4) && n >= 128 && n <= 16384 && (n % 64) == 0; }

template <typename T, int ID>
static hipError_t launch_tile2_fwd_id(hipStream_t st, const LiftTileArgs<T> &a)
{
    constexpr size_t shmem = (size_t)TileGeom2<ID>::ELEMS * sizeof(T);
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lift2d_tile2_fwd<T, ID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    const unsigned g = (unsigned)(a.n / 64);
    hipLaunchKernelGGL((k_lift2d_tile2_fwd<T, ID>), dim3(g, g), dim3(512), shmem, st, a);
    return hipGetLastError();
}

bool lift2d_tile2_inv_ok(int id, int64_t n) { return (id == 1 || id == 3 || id == 5) && n >= 256 && n <= 16384 && (n % 64) == 0; }

template <typename T, int ID>
static hipError_t launch_tile2_inv_id(hipStream_t st, const LiftTileArgs<T> &a)
{
    constexpr size_t shmem = (size_t)TileGeomInv2<ID>::ELEMS * sizeof(T);
    static thread_local int done_dev = -1;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (done_dev != dev) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lift2d_tile2_inv<T, ID>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        done_dev = dev;
    }
    const unsigned g = (unsigned)(a.n / 64);
    hipLaunchKernelGGL((k_lift2d_tile2_inv<T, ID>), dim3(g, g), dim3(256), shmem, st, a);
    return hipGetLastError();
}

// two inverse levels: x = coefficient array (leading dimension ldx), ll = approximation source of the COARSER level (or nullptr: x's
// corner), out = the n x n reconstruction of the finer level
template <typename T>
hipError_t lift2d_tile2_inv_launch(int id, hipStream_t st, const LiftScheme<T> &sc, const T *x, int64_t ldx, T *out, int64_t ldo, const T *ll, int64_t ldl,
                                   int64_t n)
{
    LiftTileArgs<T> a;
    a.src = x; a.lds = ldx; a.y = out; a.ldy = ldo; a.ll = const_cast<T *>(ll); a.ldl = ldl; a.n = (int)n;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    switch (id) {
    case 1: return launch_tile2_inv_id<T, 1>(st, a);
    case 3: return launch_tile2_inv_id<T, 3>(st, a);
    case 5: return launch_tile2_inv_id<T, 5>(st, a);
    default: return hipErrorInvalidValue;
    }
}
template hipError_t lift2d_tile2_inv_launch<float>(int, hipStream_t, const LiftScheme<float> &, const float *, int64_t, float *, int64_t, const float *, int64_t,
                                                   int64_t);
template hipError_t lift2d_tile2_inv_launch<double>(int, hipStream_t, const LiftScheme<double> &, const double *, int64_t, double *, int64_t, const double *,
                                                    int64_t, int64_t);

// ll: destination of the level-2 approximation (dense, leading dimension ldl) or nullptr (-> the top-left corner of y)
template <typename T>
hipError_t lift2d_tile2_fwd_launch(int id, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy, T *ll, int64_t ldl,
                                   int64_t n)
{
    LiftTileArgs<T> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldl = ldl; a.n = (int)n;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    switch (id) {
    case 0: return launch_tile2_fwd_id<T, 0>(st, a);
    case 2: return launch_tile2_fwd_id<T, 2>(st, a);
    case 4: return launch_tile2_fwd_id<T, 4>(st, a);
    default: return hipErrorInvalidValue;
    }
}
template hipError_t lift2d_tile2_fwd_launch<float>(int, hipStream_t, const LiftScheme<float> &, const float *, int64_t, float *, int64_t, float *, int64_t, int64_t);
template hipError_t lift2d_tile2_fwd_launch<double>(int, hipStream_t, const LiftScheme<double> &, const double *, int64_t, double *, int64_t, double *, int64_t,
                                                    int64_t);

template <typename T>
hipError_t lift2d_tile_launch(int id, int fw, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy, T *ll,
                              int64_t ldl, int64_t n)
{
    LiftTileArgs<T> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldl = ldl; a.n = (int)n;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    switch (id) {
    case 0: return fw ? launch_tile_id<T, 0, 1>(st, a) : hipErrorInvalidValue;
    case 2: return fw ? launch_tile_id<T, 2, 1>(st, a) : hipErrorInvalidValue;
    case 4: return fw ? launch_tile_id<T, 4, 1>(st, a) : hipErrorInvalidValue;
    case 1: return fw ? hipErrorInvalidValue : launch_tile_id<T, 1, 0>(st, a);
    case 3: return fw ? hipErrorInvalidValue : launch_tile_id<T, 3, 0>(st, a);
    case 5: return fw ? hipErrorInvalidValue : launch_tile_id<T, 5, 0>(st, a);
    default: return hipErrorInvalidValue;
    }
}

template hipError_t lift2d_tile_launch<float>(int, int, hipStream_t, const LiftScheme<float> &, const float *, int64_t, float *, int64_t, float *,
                                              int64_t, int64_t);
template hipError_t lift2d_tile_launch<double>(int, int, hipStream_t, const LiftScheme<double> &, const double *, int64_t, double *, int64_t,
                                               double *, int64_t, int64_t);

}  // namespace wl
