// wl_pair2d.hip -- TWO fused forward 2-D filter-bank levels per launch, Float32, even F <= 10 (transforms_filter.jl:161-186, two
// trips of the level loop): the level-l approximation never goes to HBM.
//
//   k_fwd2d_pair<F, W, LVL1>
//
// A workgroup owns a strip of 256 W rows (W = 4 or 2) and marches along the columns of a chunk exactly like k_fwd2d_lds
// (wl_fwd2d.hip): W main waves (4 rows per lane, 16-slot column ring in VGPRs filled by hand-placed global_load_dwordx4, dim-2
// pass in registers, dim-1 pass on 12-row windows read back from a two-slot LDS exchange) and one helper wave that supplies
// the 24 halo rows next to the strip (one row per lane, scalar ring: a quarter of a main wave's dim-2 instructions).  Level l+1 is NOT interleaved into those waves (the round-2 attempt: one more dependent LDS
// round trip on every step's critical path, 143 VGPRs, 135-140 us against 105 + 34 for two launches).  Instead:
//
//   * the main waves and the helper park the level-l approximation column they produce at step t (rows 0 .. 128 W + 7 of the
//     strip: 128 W owned + 8 halo rows from the helper) in slot t % 16 of an LDS column ring -- one ds_write_b64 per lane;
//   * a dedicated LEVEL-(l+1) WAVE (wave W + 1) owns all 128 W approximation rows (2 W per lane).  At even steps it runs the
//     dim-2 pass on ring columns t-F .. t-1 and publishes {scaling, detail} pairs in a second exchange buffer; at odd steps it
//     reads its (2 W + 8)-row window back, runs the dim-1 pass and stores the four level-(l+1) quadrants with 16-byte stores
//     (W = 4: s2 rows 4j .. 4j+3, d2 rows 4j+4 .. 4j+7 of lane j -- no partner exchange).  Its arithmetic per step equals a
//     main wave's (a quarter of the samples, all on one wave), so it never is the slowest wave at the step barrier;
//   * the helper also runs the level-(l+1) dim-2 pass for the 8 halo rows it produced.
//
// All waves meet at ONE s_barrier per step (the level-1 exchange's); the level-(l+1) wave needs no other synchronisation: ring
// slot c is written after barrier c and read after barriers > c, rewritten after barrier c + 16; the second exchange buffer
// is written after an even barrier and read after the following odd one.
//
// A chunk of TJ input columns owns TJ/2 level-l and TJ/4 level-(l+1) output columns.  The last level-(l+1) column needs
// level-l columns up to TJ/2 + F - 3, so the march runs F steps past the chunk: F - 2 that produce approximation columns only
// (no detail arithmetic, no stores) and 2 in which only the level-(l+1) wave works.
// Arithmetic: closed forms of wl_internal.h in the reference's order, no FMA -- bit-identical to two single-level launches.
#include "wl_fast.h"
#include "wl_dev.h"

WL_STAMP_DECL(pair)
WL_STAMP_DECL(ftile)
#define WL_STAMP(k) WL_STAMP_AT(pair, logical, k)
#define WL_TB_STAMP(wg, k) WL_STAMP_AT(ftile, wg, k)
#include "wl_tile_dev.h"

namespace wl {

template <int F>
struct Pair2DArgs {
    const float *src; int64_t lds;
    float *y; int64_t ldy;
    float *ll; int64_t ldll;          // approximation after both levels: next stage's input buffer, or y itself
    int64_t ms, ns;                   // level-l block
    int TJ;                           // owned input columns per chunk (multiple of 32)
    int nstrips, nchunks;
    int rev;
    int prio;                         // 1: the single-wave roles (helper, level-(l+1) wave) run at a raised priority;  2, 3: see rr_div
    int rr_div;                       // prio >= 2: issue priority rotates over the co-resident workgroups, rank = blockIdx / rr_div
    // BT instances (the planes of a translation-invariant denoise batch over blockIdx.y): plane strides, the virtual shift of
    // level 1 (plane p reads copy (spin0 + p) % src_mod with its columns rotated by (spin0 + p) / src_mod, as k_fwd2d_lds) and the
    // hard threshold applied to every final coefficient as it is stored (th < 0: none)
    int64_t bs_src, bs_y, bs_ll;
    int src_mod; int64_t spin0;
    int th; double t_unit, sigma_host; const double *mad_dev;
    // FUSE instances (levels l .. l+3 in ONE launch, see k_fwd2d_pair below): workgroups [0, npair) are the pair, the rest tiles
    unsigned *prog;                   // npair progress words + a completion counter at prog[npair] (all zero between launches)
    int npair;
    int tprio;                        // issue priority of the tile workgroups (0 .. 3)
    int chunk0;                       // first chunk of this launch (WL_PAIR_BANDS experiment: the column chunks in K launches)
    float *ll4; int64_t ldll4;        // approximation after levels l+2, l+3 (next stage's input buffer, or y itself)
    TapsF<float, F> tp;
};

// ---- the tile role of the FUSE instances ------------------------------------------------------------------------------------
// Workgroup `ti` of the tiles: levels l+2, l+3 of one 64 x 64 piece (+ one-sided halo) of the level-(l+1) approximation, exactly
// k_fwd2d_tileB's work (tileB_body, wl_tile_dev.h) -- but inside the launch that produces that approximation.  The pair's
// level-(l+1) waves publish it with write-through stores and count the columns they have completed in prog[pair workgroup]
// (release: every storing wave drains vmcnt, one lane stores the word at agent scope); a tile polls the words of the <= 8 pair
// workgroups it reads from, then loads with `sc1` (coherent) loads.  No dispatch-order assumption is needed for CORRECTNESS beyond
// "a workgroup with a lower index has been placed before one with a higher index is" (tiles have the highest indices and pair
// workgroups never wait), which keeps the spin deadlock-free; the tile ORDER below only decides how much of the tile work hides
// under the pair: the CU arbitrates issue by age, so the pair workgroups of dispatch round r (the r-th workgroup each CU received)
// finish in that order, ~30 % of the launch apart (r06 stamps) -- tiles are numbered so that those fed by round 0 come first.
template <int F, int W>
__device__ __forceinline__ void fused_tile_role(const Pair2DArgs<F> &a, float *smem, const uint32_t ti)
{
    typedef TileLds<F, 2> L;
    constexpr int R0 = L::R0, C0 = L::C0;
    constexpr int SR = 64 * W;                         // level-(l+1) rows per strip
    const int tid = (int)threadIdx.x;
    switch (a.tprio) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
    }
    const int M2 = (int)(a.ms >> 2), N2 = (int)(a.ns >> 2), gx = M2 >> 6, gy = N2 >> 6, ntiles = gx * gy;
    const int TJq = a.TJ >> 2;                         // level-(l+1) columns per chunk
    int bi, bj;
    if ((gy & 31) == 0) {
        // column block bj lies in XCD (bj / (gy/8))'s span of the pair's chunks, in the part its dispatch round (bj % (gy/8)) / (gy/32) owns
        const int nb = ntiles >> 2, kq = gy >> 5;
        int band = (int)ti / nb;
        const int idx = (int)ti - band * nb, xq = idx & 7, rest = idx >> 3;
        if (a.rev) band = 3 - band;
        bi = rest % gx;
        bj = (gy >> 3) * xq + kq * band + rest / gx;
    } else {
        bi = (int)ti % gx;
        bj = (int)ti / gx;
    }
    // ---- wait for the pair workgroups that produce rows [64 bi, 64 bi + R0) x columns [64 bj, 64 bj + C0) (periodic) ----
    const int r0 = 64 * bi, c0 = 64 * bj;
    const int s_lo = r0 / SR, nsd = (r0 + R0 - 1) / SR - s_lo + 1;
    const int c_lo = c0 / TJq, ncd = (c0 + C0 - 1) / TJq - c_lo + 1;
    if (tid < 64) {
        const int si = tid % nsd, ci = tid / nsd;
        const bool active = ci < ncd;
        const int ch = c_lo + ci;
        int need = c0 + C0 - ch * TJq;
        if (need > TJq) need = TJq;
        const int word = (ch % a.nchunks) * a.nstrips + (s_lo + si) % a.nstrips;
        unsigned spins = 0;
        for (;;) {
            const unsigned v = active ? __hip_atomic_load(a.prog + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned)need;
            if (__all((int)(v >= (unsigned)need))) break;
            __builtin_amdgcn_s_sleep(8);
            if (++spins > (1u << 22)) {                // (seconds: never on a healthy launch; leaves a mark instead of hanging the device)
                if (tid == 0) __hip_atomic_store(a.prog + a.npair + 1, 0xdeadu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    TileArgs<float, F> ta;
    ta.src = a.ll; ta.lds = a.ldll; ta.y = a.y; ta.ldy = a.ldy; ta.ll = a.ll4 ? a.ll4 : a.y; ta.ldll = a.ll4 ? a.ldll4 : a.ldy;
    ta.M = M2; ta.N = N2;                              // (ta.tp stays unset: the taps are read from the kernel argument)
    float *Ts = smem, *X1s = smem + L::ldx(L::R0) * (L::C1 + 32);
    tileB_body<F, 2>(ta, a.tp, Ts, X1s, bi, bj, tid, 256, (int)ti);
    // ---- the last tile to finish zeroes the hand-over words for the next launch ----
    __shared__ int last_tile;
    if (tid == 0) last_tile = (__hip_atomic_fetch_add(a.prog + a.npair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ntiles - 1));
    __syncthreads();
    if (last_tile) {
        for (int i = tid; i <= a.npair; i += 256) __hip_atomic_store(a.prog + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// FUSE = 1 (W = 2, BT = 0): FOUR levels in one launch -- workgroups >= a.npair run fused_tile_role above.
template <int F, int W, int LVL1, int BT = 0, int FUSE = 0>
__global__ void __launch_bounds__(64 * (W + 2), 3) k_fwd2d_pair(Pair2DArgs<F> a)
{
    typedef float T;
    typedef float T2 __attribute__((ext_vector_type(2)));
    typedef float T4 __attribute__((ext_vector_type(4)));
    constexpr int SH = (F - 2) / 2;
    constexpr int R = 16, U = 8, PFD = (R - F) / 2;
    constexpr int NT1 = 64 * (W + 1);                 // lanes of the level-l exchange (main waves + helper)
    constexpr int ROWS1 = 4 * NT1 + 16;
    constexpr int NPL = 64 * W;                       // owned lanes
    constexpr int RL = 128 * W + 16;                  // approximation rows per ring slot: 128 W owned + 8 halo (+ 8 pad)
    constexpr int NSLOT = 16;
    constexpr int RW = 2 * W;                         // approximation rows per lane of the level-(l+1) wave
    constexpr int HS = W;                             // its s2 (and d2) rows per lane
    // one block of LDS, carved: x1 (level-l exchange, two slots) | ll1 (approximation column ring) | x2 (level-(l+1) exchange); the tile
    // role of the FUSE instances lays its own two arrays over the same block
    constexpr int SM_PAIR = 4 * ROWS1 + NSLOT * RL + 2 * RL;
    constexpr int SM_TILE = FUSE ? (TileLds<F, 2>::ldx(TileLds<F, 2>::R0) * (TileLds<F, 2>::C1 + 32) + TileLds<F, 2>::ldx(TileLds<F, 2>::R1) * TileLds<F, 2>::C1 + 16) : 0;
    __shared__ __attribute__((aligned(16))) T smem[SM_PAIR > SM_TILE ? SM_PAIR : SM_TILE];
    if constexpr (FUSE != 0) {
        if (blockIdx.x >= (uint32_t)a.npair) {
            fused_tile_role<F, W>(a, smem, blockIdx.x - (uint32_t)a.npair);
            return;
        }
    }
    T2 *const x1 = reinterpret_cast<T2 *>(smem);
    T *const ll1 = smem + 4 * ROWS1;
    T2 *const x2 = reinterpret_cast<T2 *>(smem + 4 * ROWS1 + NSLOT * RL);

    // g[m] = (-1)^m h[m] exactly: only the scaling taps occupy SGPRs, a detail term multiplies by the negated tap (a source modifier)
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    const uint32_t b = blockIdx.x, nwg = FUSE ? (uint32_t)a.npair : gridDim.x;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1u : 0u);
    uint32_t logical = first + (b >> 3);
    if (a.rev) logical = first + (cnt - 1 - (logical - first));
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips) + a.chunk0;

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1, nxj2 = ns >> 2;
    const int msi = (int)ms, hmi = msi >> 1, hm2i = msi >> 2;
    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S_own = (int)((jend - j0) >> 1);        // steps whose level-l outputs this chunk owns (multiple of 16)
    const int S = S_own + F;                          // the level-(l+1) column of ring columns c .. c+F-1 is taken up at step c + F
    T *const yb = a.y + (BT ? (int64_t)blockIdx.y * a.bs_y : 0);
    T *const llb = a.ll ? (a.ll + (BT ? (int64_t)blockIdx.y * a.bs_ll : 0)) : yb;
    const int64_t ldl = a.ll ? a.ldll : a.ldy;
    int64_t splane = BT ? (int64_t)blockIdx.y : 0, crot = 0;
    if (BT && a.src_mod > 0) {
        const int64_t spin = a.spin0 + (int64_t)blockIdx.y;
        splane = spin % a.src_mod;
        crot = spin / a.src_mod;
    }
    const T *const srcp = a.src + (BT ? splane * a.bs_src : 0);
    // threshold! at the stores (BT): a Float32 cut, Float64 only for the survivors of soft / semisoft / Stein (ThCut, wl_dev.h)
    ThCut cut = th_make_cut(-1, 0.0);
    if (BT && a.th >= 0) cut = th_make_cut(a.th, ((a.sigma_host >= 0) ? a.sigma_host : (*a.mad_dev / 0.6745)) * a.t_unit);
    const bool th_ll = BT && (a.ll == nullptr);          // last level of the transform: its approximation is final too

    // The helper and the level-(l+1) wave issue ahead of the main waves: every wave of the workgroup meets at the step barrier,
    // and a single-wave role that loses the VALU to the two main waves of four co-resident workgroups is what the others wait
    // for (measured r04, 8192^2: db4 115.1-115.3 -> 112.2-112.4 us, sym5 125.3 -> 123.9; main waves first: +1 %; rotating the
    // roles over the wave slots by workgroup: neutral or -4 %, the default placement is already the balanced one).
    if (a.prio == 1 && wv >= W) __builtin_amdgcn_s_setprio(2);
    // prio >= 2: the CU arbitrates issue by priority, then AGE -- the first workgroup a CU received wins every conflict and finishes
    // its chunk ~30 % earlier than the three dispatched after it, which then run on a CU that is a quarter empty (r06, per-workgroup
    // stamps).  Rotating the priority over the workgroups (rank = dispatch round of the workgroup) every 8 steps equalises them.
    const int rank = (a.prio >= 2 && a.rr_div > 0) ? (int)(b / (uint32_t)a.rr_div) : 0;
    auto rot_prio = [&](const int i) __attribute__((always_inline)) {
        switch ((rank + i) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    const int psh = (a.prio == 3) ? 4 : 3;
    if (wv == W + 1) {
        // =============================== the level-(l+1) wave ===============================
        const int j = (int)(threadIdx.x & 63);
        const int s2row = strip * (64 * W) + HS * j;                         // first s2 row of this lane
        int d2row = s2row + 4;  if (d2row >= hm2i) d2row -= hm2i;            // first d2 row (details are stored shifted by 4)
        const T *const l1 = ll1 + RW * j;
        T2 *const xw = x2 + RW * j;
        const int64_t kbase2 = j0 >> 2;
        for (int t = 0; t < F; ++t) wg_lds_sync(true);
        WL_STAMP(4);
        for (int t = F; t < S; t += 2) {
            if (a.prio >= 2 && (t & 7) == 0) rot_prio(t >> psh);
            wg_lds_sync(true);                                               // barrier t (even): ring columns <= t-1 are visible
            __builtin_amdgcn_sched_barrier(0);
            {
                T sa[RW], da[RW];
#pragma unroll
                for (int m = 0; m < F; ++m) {
                    const T *const col = l1 + ((t - F + m) & (NSLOT - 1)) * RL;
                    T xm[RW];
#pragma unroll
                    for (int c = 0; c < RW / 4; ++c) {
                        const T4 v = *reinterpret_cast<const T4 *>(col + 4 * c);
                        xm[4 * c] = v.x; xm[4 * c + 1] = v.y; xm[4 * c + 2] = v.z; xm[4 * c + 3] = v.w;
                    }
#pragma unroll
                    for (int r = 0; r < RW; ++r) {
                        if (m == 0) { sa[r] = a.tp.h[0] * xm[r]; da[r] = gq(F - 1) * xm[r]; }
                        else { sa[r] = sa[r] + a.tp.h[m] * xm[r]; da[r] = da[r] + gq(F - 1 - m) * xm[r]; }
                    }
                }
#pragma unroll
                for (int c = 0; c < RW / 2; ++c)
                    *reinterpret_cast<T4 *>(xw + 2 * c) = T4{sa[2 * c], da[2 * c], sa[2 * c + 1], da[2 * c + 1]};
            }
            wg_lds_sync(true);                                               // barrier t + 1 (odd)
            __builtin_amdgcn_sched_barrier(0);
            {
                T2 E[RW + 8];
#pragma unroll
                for (int c = 0; c < (RW + 8) / 2; ++c) {
                    const T4 v = *reinterpret_cast<const T4 *>(xw + 2 * c);
                    E[2 * c] = T2{v.x, v.y};
                    E[2 * c + 1] = T2{v.z, v.w};
                }
                T2 P[HS], Q[HS];                       // P[q] = {ss2, sd2} of row s2row + q;  Q[q] = {ds2, dd2} of row d2row + q
#pragma unroll
                for (int q = 0; q < HS; ++q) {
                    T2 s = a.tp.h[0] * E[2 * q];
#pragma unroll
                    for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * E[2 * q + m];
                    T2 d = gq(F - 1) * E[2 * q + 10 - F];
#pragma unroll
                    for (int m = F - 2; m >= 0; --m) d = d + gq(m) * E[2 * q + 9 - m];
                    P[q] = s;
                    Q[q] = d;
                }
                const int64_t k2 = kbase2 + ((t - F) >> 1);
                int64_t kd2 = k2 + SH;
                if (kd2 >= nxj2) kd2 -= nxj2;
                T *const ck = yb + k2 * a.ldy, *const ckd = yb + (nxj2 + kd2) * a.ldy, *const cl = llb + k2 * ldl;   // (uniform)
                if constexpr (BT != 0) {
                    bool above = false;
#pragma unroll
                    for (int q = 0; q < HS; ++q) {
                        if (th_ll) P[q].x = th_cut(cut, P[q].x);
                        P[q].y = th_cut(cut, P[q].y); Q[q].x = th_cut(cut, Q[q].x); Q[q].y = th_cut(cut, Q[q].y);
                        above = above || (th_ll && P[q].x != 0.f) || P[q].y != 0.f || Q[q].x != 0.f || Q[q].y != 0.f;
                    }
                    if (th_needs_exact(cut, above || cut.tf < 0.f)) {
#pragma unroll
                        for (int q = 0; q < HS; ++q) {
                            if (th_ll) P[q].x = th_exact(cut, P[q].x);
                            P[q].y = th_exact(cut, P[q].y); Q[q].x = th_exact(cut, Q[q].x); Q[q].y = th_exact(cut, Q[q].y);
                        }
                    }
                }
                if constexpr (HS == 4) {
                    store_pol<(FUSE ? 2 : WL_P_PAIR_LL)>(reinterpret_cast<T4 *>(cl + s2row), T4{P[0].x, P[1].x, P[2].x, P[3].x});
                    store_pol<WL_P_PAIR_ST2>(reinterpret_cast<T4 *>(ck + (hm2i + d2row)), T4{Q[0].x, Q[1].x, Q[2].x, Q[3].x});
                    store_pol<WL_P_PAIR_ST2>(reinterpret_cast<T4 *>(ckd + s2row), T4{P[0].y, P[1].y, P[2].y, P[3].y});
                    store_pol<WL_P_PAIR_ST2>(reinterpret_cast<T4 *>(ckd + (hm2i + d2row)), T4{Q[0].y, Q[1].y, Q[2].y, Q[3].y});
                } else {
                    store_pol<(FUSE ? 2 : WL_P_PAIR_LL)>(reinterpret_cast<T2 *>(cl + s2row), T2{P[0].x, P[1].x});
                    store_pol<WL_P_PAIR_ST2>(reinterpret_cast<T2 *>(ck + (hm2i + d2row)), T2{Q[0].x, Q[1].x});
                    store_pol<WL_P_PAIR_ST2>(reinterpret_cast<T2 *>(ckd + s2row), T2{P[0].y, P[1].y});
                    store_pol<WL_P_PAIR_ST2>(reinterpret_cast<T2 *>(ckd + (hm2i + d2row)), T2{Q[0].y, Q[1].y});
                }
                if constexpr (FUSE != 0) {
                    // publish: this wave is the only writer of the chunk's level-(l+1) approximation columns (write-through stores);
                    // once they have left (vmcnt drained), one lane counts them for the tiles -- after the head columns a neighbouring
                    // tile column needs, and at the end of the chunk
                    const int kl = (t - F) >> 1;
                    if (kl == TileLds<F, 2>::C0 - 64 - 1 || kl == (S_own >> 1) - 1) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (j == 0) __hip_atomic_store(a.prog + logical, (unsigned)(kl + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        WL_STAMP(5);
        return;
    }

    if (wv == W) {
        // =============================== the helper wave: ONE halo row per lane ===============================
        // The 24 rows below the strip (rows 4 NPL .. 4 NPL + 23 of it, periodic) feed the last main lanes' windows and the 8 halo
        // approximation rows of the level-(l+1) wave.  With four rows per lane, as in the main waves, six lanes did useful work and
        // the wave still issued a full wave's dim-2 pass every step (an eighth of the workgroup's VALU work); one row per lane
        // and a scalar column ring cost a quarter of that.  Lanes >= 24 load row 0 and compute values nobody reads.
        const int hl = (int)(threadIdx.x & 63);
        const int lp = NPL + hl;                           // lane index in the level-l exchange (lanes NPL .. NPL+3 own approximation rows)
        int hrow = strip * (4 * NPL) + 4 * NPL + hl;
        if (hrow >= msi) hrow -= msi;
        const T *hbase = srcp + ((hl < 24) ? hrow : 0);
        T hring[R];
#pragma unroll
        for (int c = 0; c < R; ++c) hring[c] = 0.f;
#pragma unroll
        for (int c = 0; c < R - 2; ++c) {
            int64_t jc = j0 + c;
            if (jc >= ns) jc -= ns;
            if (BT) { jc -= crot; if (jc < 0) jc += ns; }
            gload4<WL_P_PAIR_LDH != 0>(hring[c], hbase + jc * a.lds);
        }
#pragma unroll
        for (int c = 0; c < R; c += 2) wait_vm<0>(hring[c], hring[c + 1]);
        auto hstep = [&](const int t, const int u, const bool prefetch, const bool full, const bool produce) __attribute__((always_inline)) {
            if (prefetch) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int64_t jc = j0 + 2 * t + (R - 2) + e;
                    if (jc >= ns) jc -= ns;
                    if (jc >= ns) jc -= ns;
                    if (BT) { jc -= crot; if (jc < 0) jc += ns; }
                    gload4<WL_P_PAIR_LDH != 0>(hring[(2 * u + R - 2 + e) % R], hbase + jc * a.lds);
                }
            }
            if (produce) {
                if (prefetch) wait_vm<2 * PFD>(hring[(2 * u + F - 2) % R], hring[(2 * u + F - 1) % R]);      // (loads only: see the main waves)
                else wait_vm<0>(hring[(2 * u + F - 2) % R], hring[(2 * u + F - 1) % R]);
            }
            T2 *const w1 = x1 + (t & 1) * ROWS1;
            if (produce) {
                T sa = a.tp.h[0] * hring[(2 * u) % R];
#pragma unroll
                for (int m = 1; m < F; ++m) sa = sa + a.tp.h[m] * hring[(2 * u + m) % R];
                T da = 0.f;
                if (full) {
                    da = gq(F - 1) * hring[(2 * u) % R];
#pragma unroll
                    for (int m = 1; m < F; ++m) da = da + gq(F - 1 - m) * hring[(2 * u + m) % R];
                }
                if (hl < 24) w1[4 * NPL + hl] = T2{sa, da};
            }
            wg_lds_sync(true);
            __builtin_amdgcn_sched_barrier(0);
            // ---- even steps: level-(l+1) dim-2 pass of the halo rows 128 W + 2i, 128 W + 2i + 1 (lanes i = 0..3) ----
            if (!(u & 1) && t >= F) {
                if (hl < 4) {
                    T2 s2, d2;
#pragma unroll
                    for (int m = 0; m < F; ++m) {
                        const T2 xm = *reinterpret_cast<const T2 *>(ll1 + ((t - F + m) & (NSLOT - 1)) * RL + 2 * lp);
                        if (m == 0) { s2 = a.tp.h[0] * xm; d2 = gq(F - 1) * xm; }
                        else { s2 = s2 + a.tp.h[m] * xm; d2 = d2 + gq(F - 1 - m) * xm; }
                    }
                    *reinterpret_cast<T4 *>(x2 + 2 * lp) = T4{s2.x, d2.x, s2.y, d2.y};
                }
            }
            if (!produce) return;
            // ---- approximation rows ko, ko+1 of lanes i = 0..3 from window rows 4L' .. 4L'+F+1 ----
            if (hl < 4) {
                T A[12];
#pragma unroll
                for (int c = 0; c < (F + 3) / 2; ++c) {
                    const T4 v = *reinterpret_cast<const T4 *>(w1 + 4 * lp + 2 * c);
                    A[2 * c] = v.x;
                    A[2 * c + 1] = v.z;
                }
                T p0 = a.tp.h[0] * A[0], p1 = a.tp.h[0] * A[2];
#pragma unroll
                for (int m = 1; m < F; ++m) { p0 = p0 + a.tp.h[m] * A[m]; p1 = p1 + a.tp.h[m] * A[2 + m]; }
                *reinterpret_cast<T2 *>(ll1 + (t & (NSLOT - 1)) * RL + 2 * lp) = T2{p0, p1};
            }
        };
        int t0 = 0;
        for (; t0 < S_own; t0 += U) {
            if (a.prio >= 2) rot_prio(t0 >> psh);
#pragma unroll
            for (int u = 0; u < U; ++u) hstep(t0 + u, u, true, true, true);
        }
        if constexpr (F == 2) drain_ring(hring);           // (see the main waves)
#pragma unroll
        for (int u = 0; u < F; ++u) hstep(t0 + u, u, u + PFD < F - 2, false, u < F - 2);
        return;
    }

    // =============================== main waves: level l ===============================
    const int lp = threadIdx.x;                       // L': lane index within the workgroup's strip (0 .. NT1-1)
    const int gi = strip * (4 * NPL) + 4 * lp;        // first row of this lane (halo lanes may exceed ms: wrap)
    int row = gi;
    if (row >= msi) row -= msi;
    const int ko = gi >> 1;
    int kod = ko + 4;  if (kod >= hmi) kod -= hmi;    // first d row of this lane
    const bool odd = (lp & 1) != 0;
    const T *base = srcp + row;
    const int64_t kbase = j0 >> 1;

    if (wv == 0) WL_STAMP(0);
#ifdef WL_WGTIME
    if (threadIdx.x == 0 && logical < 8192) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hwid));
        wl_dbg_pair[8 * logical + 7] = ((unsigned long long)blockIdx.x << 32) | ((xcc & 0xf) << 16) | (hwid & 0xffff);
    }
#endif
    T4 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        if (BT) { jc -= crot; if (jc < 0) jc += ns; }
        gload16<WL_P_PAIR_LD != 0>(ring[c], base + jc * a.lds);
    }
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);

    // FULL: a step inside the chunk (details computed and stored).  !FULL: one of the F - 2 steps past the chunk that only produce
    // the approximation column (and, for the helper, the level-(l+1) dim-2 pass of its halo rows).
    auto step = [&](const int t, const int u, const bool prefetch, const bool full, const bool produce) __attribute__((always_inline)) {
        if (prefetch) {                                    // (compile-time)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                if (BT) { jc -= crot; if (jc < 0) jc += ns; }
                gload16<WL_P_PAIR_LD != 0>(ring[(2 * u + R - 2 + e) % R], base + jc * a.lds);
            }
        }
        if (produce) {
            if (prefetch) {
                // The two newest window columns were requested PFD steps ago.  The wait counts LOADS only: "at most 2 PFD vector
                // memory operations outstanding" is implied by "those two loads have landed" only if every younger operation
                // counted is a load (loads return in order among themselves).  Counting the younger STORES as well (3 per step
                // for a main wave) looked equivalent but is not: on some MI355X boxes about one transform in a thousand came
                // back with a few columns computed from stale ring registers (10-tap filters, PFD = 3) -- stores can be
                // acknowledged before an older load returns.  Waiting for the older stores too costs nothing measurable
                // (8192^2 db4 pair: 114.8 vs 115.0 us).
                wait_vm<2 * PFD>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
            } else {
                wait_vm<0>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
            }
        }
        T2 *const w1 = x1 + (t & 1) * ROWS1;
        const bool sonly = !full;                          // (compile-time)
        if (produce) {
            // ---- level l, dim-2 pass on row pairs: {A, B}[r] = scaling / detail (column k / kd) of row r ----
            T2 sa01 = a.tp.h[0] * T2{ring[(2 * u) % R].x, ring[(2 * u) % R].y};
            T2 sa23 = a.tp.h[0] * T2{ring[(2 * u) % R].z, ring[(2 * u) % R].w};
#pragma unroll
            for (int m = 1; m < F; ++m) {
                const T4 xm = ring[(2 * u + m) % R];
                sa01 = sa01 + a.tp.h[m] * T2{xm.x, xm.y};
                sa23 = sa23 + a.tp.h[m] * T2{xm.z, xm.w};
            }
            T2 da01 = T2{0.f, 0.f}, da23 = T2{0.f, 0.f};
            if (!sonly) {
                da01 = gq(F - 1) * T2{ring[(2 * u) % R].x, ring[(2 * u) % R].y};
                da23 = gq(F - 1) * T2{ring[(2 * u) % R].z, ring[(2 * u) % R].w};
#pragma unroll
                for (int m = 1; m < F; ++m) {
                    const T4 xm = ring[(2 * u + m) % R];
                    da01 = da01 + gq(F - 1 - m) * T2{xm.x, xm.y};
                    da23 = da23 + gq(F - 1 - m) * T2{xm.z, xm.w};
                }
            }
            *reinterpret_cast<T4 *>(w1 + 4 * lp) = T4{sa01.x, da01.x, sa01.y, da01.y};
            *reinterpret_cast<T4 *>(w1 + 4 * lp + 2) = T4{sa23.x, da23.x, sa23.y, da23.y};
        }
        wg_lds_sync(true);
        __builtin_amdgcn_sched_barrier(0);
        if (!produce) return;
        T *const slot = ll1 + (t & (NSLOT - 1)) * RL + 2 * lp;
        if (sonly) {
            // ---- approximation only: ss rows ko, ko+1 from window rows 4L' .. 4L'+F+1 ----
            T A[12];
#pragma unroll
            for (int c = 0; c < (F + 3) / 2; ++c) {
                const T4 v = *reinterpret_cast<const T4 *>(w1 + 4 * lp + 2 * c);
                A[2 * c] = v.x;
                A[2 * c + 1] = v.z;
            }
            T p0 = a.tp.h[0] * A[0], p1 = a.tp.h[0] * A[2];
#pragma unroll
            for (int m = 1; m < F; ++m) { p0 = p0 + a.tp.h[m] * A[m]; p1 = p1 + a.tp.h[m] * A[2 + m]; }
            *reinterpret_cast<T2 *>(slot) = T2{p0, p1};
            return;
        }
        // ---- level l, dim-1 pass: window rows 4L' .. 4L'+11 as {A, B} pairs ----
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
            T2 s = a.tp.h[0] * E[2 * q];
#pragma unroll
            for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * E[2 * q + m];
            T2 d = gq(F - 1) * E[2 * q + 10 - F];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + gq(m) * E[2 * q + 9 - m];
            P[q] = s;
            Q[q] = d;
        }
        *reinterpret_cast<T2 *>(slot) = T2{P[0].x, P[1].x};             // approximation column kbase + t -> ring
        if constexpr (BT != 0) {
            bool above = false;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                P[q].y = th_cut(cut, P[q].y); Q[q].x = th_cut(cut, Q[q].x); Q[q].y = th_cut(cut, Q[q].y);
                above = above || P[q].y != 0.f || Q[q].x != 0.f || Q[q].y != 0.f;
            }
            if (th_needs_exact(cut, above || cut.tf < 0.f)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) { P[q].y = th_exact(cut, P[q].y); Q[q].x = th_exact(cut, Q[q].x); Q[q].y = th_exact(cut, Q[q].y); }
            }
        }
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        // even lane: ds rows kod..kod+3 of column k;  odd lane: sd rows ko-2..ko+1 and dd rows kod-2..kod+1 of column kd
        T rA[2], rB[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            rA[q] = from_partner(odd ? Q[q].x : P[q].y);
            rB[q] = from_partner(Q[q].y);
        }
        T *const ck = yb + k * a.ldy, *const ckd = yb + (nxj + kd) * a.ldy;      // (uniform)
        if (!odd) {
            store_pol<WL_P_PAIR_ST1>(reinterpret_cast<T4 *>(ck + (hmi + kod)), T4{Q[0].x, Q[1].x, rA[0], rA[1]});
        } else {
            store_pol<WL_P_PAIR_ST1>(reinterpret_cast<T4 *>(ckd + (ko - 2)), T4{rA[0], rA[1], P[0].y, P[1].y});
            store_pol<WL_P_PAIR_ST1>(reinterpret_cast<T4 *>(ckd + (hmi + kod - 2)), T4{rB[0], rB[1], Q[0].y, Q[1].y});
        }
    };

    if (wv == 0) WL_STAMP(1);
    int t0 = 0;
    for (; t0 < S_own; t0 += U) {
        if (a.prio >= 2) rot_prio(t0 >> psh);
#pragma unroll
        for (int u = 0; u < U; ++u) step(t0 + u, u, true, true, true);
        if (wv == 0 && t0 == S_own / 2 - U) WL_STAMP(6);
    }
    if (wv == 0) WL_STAMP(2);
    // F steps past the chunk: columns S_own .. S_own+F-3 of the approximation feed the last level-(l+1) columns; loads are needed
    // up to step S_own + F - 3 (requested PFD steps ahead).  2 taps: none of these steps waits for a load, yet the last PFD
    // steps' prefetches (columns nobody needs) are still on their way -- they must not land in registers the compiler reuses
    if constexpr (F == 2) drain_ring(ring);
#pragma unroll
    for (int u = 0; u < F; ++u) step(t0 + u, u, u + PFD < F - 2, false, u < F - 2);
    if (wv == 0) WL_STAMP(3);
}

// ------------------------------------------------------------------------------------------
bool fwd2d_pair_ok(int F, int64_t ms, int64_t ns)
{
    if (F < 2 || F > 10 || (F & 1)) return false;
    if (ms >= ((int64_t)1 << 29)) return false;
    // rows: exact strips of 512 (W = 2) or 1024 (W = 4); columns: level-(l+1) columns come in pairs of steps, chunks of 32
    return ms >= 512 && (ms % 512) == 0 && ns >= 64 && (ns % 32) == 0;
}

struct PairBatch { int64_t nbatch, bs_src, bs_y, bs_ll; int src_mod; int64_t spin0; const SrcView *thresh; };
struct PairFuse { float *ll4; int64_t ldll4; unsigned *prog; };       // levels l+2, l+3 in the same launch (FUSE instances)

static int pair_chunk_len(int64_t nstrips, int64_t ns, int64_t nbatch, int cu_count)
{
    int TJ = (int)opt("WL_TJ2", 128);
    if (TJ < 32) TJ = 32;
    TJ &= ~31;
    // one resident round of workgroups (W = 2: four 4-wave workgroups per CU): shorter chunks pay 3 (F - 2) halo columns each,
    // but a chip that is not full is latency-bound (8192^2: 1024 workgroups 116 us, 512 workgroups 129 us; 4096^2: 34.5 vs 39 us)
    auto nwgs = [&](int tj) { return nstrips * ((ns + tj - 1) / tj) * nbatch; };
    while (TJ > 32 && (TJ % 64) == 0 && nwgs(TJ) < (int64_t)cu_count * opt("WL_PAIR_WG_PER_CU", 4)) TJ >>= 1;
    return TJ;
}

template <int F, int W>
static hipError_t launch_pair_fw(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                                 float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count, const PairBatch *pb, const PairFuse *pf = nullptr)
{
    Pair2DArgs<F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.bs_src = pb ? pb->bs_src : 0; a.bs_y = pb ? pb->bs_y : 0; a.bs_ll = pb ? pb->bs_ll : 0;
    a.src_mod = pb ? pb->src_mod : 0; a.spin0 = pb ? pb->spin0 : 0;
    a.th = (pb && pb->thresh) ? pb->thresh->th : -1; a.t_unit = (pb && pb->thresh) ? pb->thresh->t_unit : 0.0;
    a.sigma_host = (pb && pb->thresh) ? pb->thresh->sigma_host : 0.0; a.mad_dev = (pb && pb->thresh) ? pb->thresh->mad_dev : nullptr;
    const int64_t nbatch = pb ? pb->nbatch : 1;
    a.nstrips = (int)(ms / (256 * W));
    a.TJ = pair_chunk_len(a.nstrips, ns, nbatch, cu_count);
    a.nchunks = (int)((ns + a.TJ - 1) / a.TJ);
    a.rev = (!lvl1 && opt("WL_REVERSE", 1)) ? 1 : 0;
    // 3: issue priority rotates over the co-resident workgroups every 16 steps (r06: 8192^2 L = 13 -1.5 .. -2.5 us against 1 = raised
    // single-wave roles, in four interleaved A/B runs on three boxes; 2 = every 8 steps: neutral; 0 = none: +5 us)
    a.prio = (int)opt("WL_PAIR_PRIO", 3);
    if (a.prio < 0 || a.prio > 3) a.prio = 3;
    a.rr_div = cu_count > 0 ? cu_count : 256;
    a.tp = shrink<float, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    a.tprio = (int)opt("WL_FUSE_TPRIO", 0) & 3;
    a.chunk0 = 0;
    a.prog = pf ? pf->prog : nullptr; a.npair = (int)nwg; a.ll4 = pf ? pf->ll4 : nullptr; a.ldll4 = pf ? pf->ldll4 : 0;
    if (pf) {
        if constexpr (W == 2) {
            const unsigned ntiles = (unsigned)((ms >> 8) * (ns >> 8));          // 64 x 64 pieces of the level-(l+1) approximation
            if (lvl1) hipLaunchKernelGGL((k_fwd2d_pair<F, 2, 1, 0, 1>), dim3(nwg + ntiles), dim3(64 * (W + 2)), 0, st, a);
            else hipLaunchKernelGGL((k_fwd2d_pair<F, 2, 0, 0, 1>), dim3(nwg + ntiles), dim3(64 * (W + 2)), 0, st, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else if (pb) {
        if constexpr (W == 2) {          // (the batched instances exist for the default strip shape only)
            if (lvl1) hipLaunchKernelGGL((k_fwd2d_pair<F, 2, 1, 1>), dim3(nwg, (unsigned)nbatch), dim3(64 * (W + 2)), 0, st, a);
            else hipLaunchKernelGGL((k_fwd2d_pair<F, 2, 0, 1>), dim3(nwg, (unsigned)nbatch), dim3(64 * (W + 2)), 0, st, a);
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        // WL_PAIR_BANDS = K > 1 (measurement only, r06): the column chunks in K launches of nchunks / K chunks each, back to back on the
        // call's stream -- what splitting the launch into bands costs before any overlap with the tiles could pay it back
        int K = (int)opt("WL_PAIR_BANDS", 1);
        if (K < 1 || (a.nchunks % K) != 0) K = 1;
        const int per = a.nchunks / K;
        for (int k = 0; k < K; ++k) {
            a.chunk0 = k * per;
            const unsigned g = (unsigned)(a.nstrips * per);
            if (lvl1) hipLaunchKernelGGL((k_fwd2d_pair<F, W, 1>), dim3(g), dim3(64 * (W + 2)), 0, st, a);
            else hipLaunchKernelGGL((k_fwd2d_pair<F, W, 0>), dim3(g), dim3(64 * (W + 2)), 0, st, a);
        }
    }
    return hipGetLastError();
}

template <int F>
static hipError_t launch_pair_f(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                                float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count, const PairBatch *pb)
{
    if (pb) return launch_pair_fw<F, 2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, pb);
    // strips of 512 rows by default: 4-wave workgroups, four per CU, beat 6-wave workgroups of 1024 rows (8192^2 db4: 116 vs 135 us)
    int W = (int)opt("WL_PAIR_W", 2);
    if (W != 2 && W != 4) W = 2;
    if ((ms % (256 * W)) != 0) W = 2;
    if (W == 4) return launch_pair_fw<F, 4>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nullptr);
    return launch_pair_fw<F, 2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nullptr);
}

hipError_t fwd2d_pair_launch(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                             float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count, int64_t nbatch, int64_t bs_src, int64_t bs_y,
                             int64_t bs_ll, int src_mod, int64_t spin0, const SrcView *thresh)
{
    PairBatch pbv = {nbatch, bs_src, bs_y, bs_ll, src_mod, spin0, thresh};
    const PairBatch *pb = (nbatch > 1 || src_mod > 0 || thresh) ? &pbv : nullptr;
    if (pb && nbatch > 65535) return hipErrorInvalidValue;
    switch (taps.F) {
    case 2: return launch_pair_f<2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, pb);
    case 4: return launch_pair_f<4>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, pb);
    case 6: return launch_pair_f<6>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, pb);
    case 8: return launch_pair_f<8>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, pb);
    case 10: return launch_pair_f<10>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, pb);
    default: return hipErrorInvalidValue;
    }
}

// ---- levels l .. l+3 in one launch: the pair (levels l, l+1) and, behind in-launch hand-over flags, the 64 x 64 tiles of levels
//      l+2, l+3 (fused_tile_role).  `prog`: the context's zeroed hand-over block (wl::tl_sync). ----
bool fwd2d_pair_tile_ok(int F, int64_t ms, int64_t ns, int cu_count)
{
    if (!fwd2d_pair_ok(F, ms, ns) || !fwd2d_tile_ok(F, 2, ms >> 2, ns >> 2)) return false;
    if ((ms % 256) != 0 || (ns % 256) != 0) return false;
    const int64_t nstrips = ms / 512;
    const int TJ = pair_chunk_len(nstrips, ns, 1, cu_count);
    if ((ns % TJ) != 0) return false;                                        // whole chunks only: a chunk publishes TJ / 4 columns
    const int64_t npair = nstrips * (ns / TJ), ntiles = (ms >> 8) * (ns >> 8);
    return npair + 2 <= (int64_t)kSyncWords && npair + ntiles < ((int64_t)1 << 31);
}

hipError_t fwd2d_pair_tile_launch(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                                  float *ll2, int64_t ldll2, float *ll4, int64_t ldll4, int64_t ms, int64_t ns, int cu_count, unsigned *prog)
{
    if (!prog || !ll2) return hipErrorInvalidValue;
    PairFuse pf = {ll4, ldll4, prog};
    switch (taps.F) {
    case 2: return launch_pair_fw<2, 2>(st, taps, lvl1, src, lds, y, ldy, ll2, ldll2, ms, ns, cu_count, nullptr, &pf);
    case 4: return launch_pair_fw<4, 2>(st, taps, lvl1, src, lds, y, ldy, ll2, ldll2, ms, ns, cu_count, nullptr, &pf);
    case 6: return launch_pair_fw<6, 2>(st, taps, lvl1, src, lds, y, ldy, ll2, ldll2, ms, ns, cu_count, nullptr, &pf);
    case 8: return launch_pair_fw<8, 2>(st, taps, lvl1, src, lds, y, ldy, ll2, ldll2, ms, ns, cu_count, nullptr, &pf);
    case 10: return launch_pair_fw<10, 2>(st, taps, lvl1, src, lds, y, ldy, ll2, ldll2, ms, ns, cu_count, nullptr, &pf);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
