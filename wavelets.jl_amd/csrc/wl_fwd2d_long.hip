// wl_fwd2d_long.hip -- forward 2-D filter-bank levels, Float32, 12..20 taps (db6..db10, sym6..sym10, coif4 / coif6, beyl) in
// ONE pass over HBM: the LDS-exchange streaming kernel of wl_fwd2d.hip with a wider window.
//
//   k_fwd2d_lds_long<F, R, LVL1>
//
// Round 2 ran these filters as two passes per level (axis pass into a scratch array + line pass: 0.40-0.47 ms for 8192^2).
// Here, as for F <= 10: a main wave owns 256 rows (4 per lane), marches along the columns of a chunk with an R-slot column
// ring in VGPRs (R = 24: a window of up to F = 20 columns + the columns in flight), runs the dim-2 pass in registers,
// publishes {scaling, detail} pairs in a two-slot LDS exchange and reads its WIN-row window back for the dim-1 pass:
//     s rows 2L', 2L'+1            from window rows 4L' .. 4L'+F+1
//     d rows 2L'+DSH, 2L'+DSH+1    from window rows 4L'+2 DSH+2-F .. 4L'+2 DSH+3     (d[k] uses x[2k+2-F .. 2k+1])
// with DSH = the smallest multiple of 4 >= (F-2)/2 (the d rows of a lane pair stay one aligned group of four rows that never
// straddles the periodic wrap) and WIN = max(F + 2, 2 DSH + 4): 20 rows for 12..18 taps, 28 for 20.  Exact tiling only: W main waves + a
// helper wave that supplies the WIN - 4 halo rows above the strip.  The kernel is bound by VALU issue, not by HBM, from about
// 16 taps on (2 F multiply-adds per sample and pass, no FMA: the price of bit-exactness).
// The ring has R = 24 slots, so the unrolled body covers 12 steps; chunks need not be a multiple of that: every step is
// guarded by its (workgroup-uniform) index, the barrier count stays uniform.
#include "wl_fast.h"
#include "wl_dev.h"

// Exchange layout (round 6): the {s, d} pairs of rows (2u, 2u+1) are one 16-byte unit u; a lane writes units 2L', 2L'+1 and reads units
// 2L' .. 2L'+WIN/2-1.  Stored consecutively (byte address 32 L' + 16 c) every ds_read/write_b128 ran at a lane stride of 32 bytes:
// two lanes per bank, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50 (profiles/r06_secondary_pmc_table.md).  Split by parity -- even
// units in one plane, odd units in the other, unit u at slot u >> 1 -- lane L' touches slot L' + (c >> 1) of plane c & 1: lane stride 16
// bytes, conflict-free, for the writes and for every window read.  Measured (8192^2, L = 13, three interleaved rounds): db6 222.5 -> 217.0 us,
// sym8 259.6 -> 262.3, db10 300.5 -> 300.4 -- the conflicts are real but these kernels wait on their dependent multiply-add chains, not on
// LDS; the split layout is used for 12 and 14 taps, where it pays.
#ifndef WL_LONG_SPLIT
#define WL_LONG_SPLIT 1
#endif

namespace wl {

template <int F>
struct LdsLongArgs {
    const float *src; int64_t lds;
    float *y; int64_t ldy;
    float *ll; int64_t ldll;          // approximation: next stage's input buffer, or y itself
    int64_t ms, ns;                   // level-l block
    int TJ;                           // owned input columns per chunk (even)
    int nstrips, nchunks;
    int npl;                          // owned lanes per workgroup (64 W)
    int rev;
    int prio;                         // wave priorities by role (s_setprio): 1 helper first, 2 main waves first; 0 = none (default)
    TapsF<float, F> tp;
};

template <int F>
struct LongGeom {
    static constexpr int SHD = (F - 2) / 2;
    // shift of the stored detail rows: a multiple of 4, so that the four d rows of a lane pair are one aligned 16-byte group that
    // never straddles the periodic wrap of the detail quadrant
    static constexpr int DSH = ((SHD + 3) / 4) * 4;
    static constexpr int WIN = ((F + 2 > 2 * DSH + 4 ? F + 2 : 2 * DSH + 4) + 1) & ~1;
    static constexpr int HL = (WIN - 4 + 3) / 4;                                  // helper lanes that load halo rows
};

template <int F, int R, int LVL1>
__global__ void __launch_bounds__(320, 2) k_fwd2d_lds_long(LdsLongArgs<F> a)
{
    typedef float T;
    typedef float T2 __attribute__((ext_vector_type(2)));
    typedef float T4 __attribute__((ext_vector_type(4)));
    typedef LongGeom<F> G;
    constexpr int SH = (F - 2) / 2;
    constexpr int U = R / 2, PFD = (R - F) / 2;
    constexpr int DSH = G::DSH, WIN = G::WIN, HL = G::HL;
    static_assert(PFD >= 2 && (R % 2) == 0, "ring too small");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int nthreads = blockDim.x;
    const int lp = threadIdx.x;                       // L': lane index within the workgroup's strip
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1u : 0u);
    uint32_t logical = first + (b >> 3);
    if (a.rev) logical = first + (cnt - 1 - (logical - first));
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);

    const int rows1 = 4 * nthreads + WIN + 4;         // exchange rows per slot
    [[maybe_unused]] const int NU = nthreads + 2;      // 16-byte units per parity plane (WL_LONG_SPLIT)
    T2 *const x1 = reinterpret_cast<T2 *>(smem_raw);

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1;
    const int msi = (int)ms, hmi = msi >> 1;
    const int gi = strip * (4 * a.npl) + 4 * lp;      // first row of this lane (halo lanes may exceed ms: wrap)
    int row = gi;
    if (row >= msi) row -= msi;
    const bool loader = lp < a.npl + HL;
    const bool helper = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == (nthreads >> 6) - 1;
    if (a.prio == 1 && helper) __builtin_amdgcn_s_setprio(2);
    else if (a.prio == 2 && !helper) __builtin_amdgcn_s_setprio(2);
    const int ko = gi >> 1;
    int kod = ko + DSH;  if (kod >= hmi) kod -= hmi;  // first d row of this lane
    const bool odd = (lp & 1) != 0;

    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S = (int)((jend - j0) >> 1);            // steps = output columns of this chunk
    // lanes that hold no input rows (the helper's upper lanes) load the first rows of the same column: one extra cache line per
    // column, and the loads below need no per-lane predicate
    const T *base = a.src + (loader ? row : 0);

    T4 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        gload16<WL_P_LONG_LD != 0>(ring[c], base + jc * a.lds);
    }
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);
    T *const yb = a.y;
    T *const llb = a.ll ? a.ll : a.y;
    const int64_t ldl = a.ll ? a.ldll : a.ldy;
    const int64_t kbase = j0 >> 1;

    // g[m] = (-1)^m h[m] exactly (wl_internal.h: make_taps), so only the scaling taps travel in SGPRs: a detail term is the
    // product with the negated tap -- a source modifier of the multiply, not an instruction (2 F SGPRs instead of 4 F: the
    // 20-tap instance spilled 125 SGPRs with both tables)
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    // may_prefetch (compile-time): false for the steps of the remainder that cannot have a successor PFD steps ahead -- no load is
    // emitted there at all (a load whose result nobody consumes would leave the compiler free to reuse its registers at once)
    auto step = [&](const int t, const int u, const bool may_prefetch) __attribute__((always_inline)) {
        // the columns requested here are the newest two of step t + PFD's window
        // The load is skipped INSIDE the asm statement when the chunk needs no more columns: a C++ `if` around an asynchronous
        // load makes its destination a phi, and a register copy at the join would read the register before the data has landed
        // (seen with this kernel: every step from 2 PFD on came back with stale columns).
        if (may_prefetch) {
            const int prefetch = __builtin_amdgcn_readfirstlane((t + PFD < S) ? 1 : 0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                gload16_if<WL_P_LONG_LD != 0>(ring[(2 * u + R - 2 + e) % R], base + jc * a.lds, prefetch);
            }
            // loads only in the count (wl_dev.h); once the prefetch has stopped, fewer than 2 PFD loads are behind: drain.  One
            // asm statement, not an `if` around two waits: see wait_vm_sel
            wait_vm_sel<2 * PFD>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R], prefetch);
        } else {
            wait_vm<0>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
        }
        // ---- dim-2 pass on row pairs: {A, B}[r] = scaling / detail (column k / kd) of row r ----
        T2 sa01 = a.tp.h[0] * T2{ring[(2 * u) % R].x, ring[(2 * u) % R].y};
        T2 da01 = gq(F - 1) * T2{ring[(2 * u) % R].x, ring[(2 * u) % R].y};
        T2 sa23 = a.tp.h[0] * T2{ring[(2 * u) % R].z, ring[(2 * u) % R].w};
        T2 da23 = gq(F - 1) * T2{ring[(2 * u) % R].z, ring[(2 * u) % R].w};
#pragma unroll
        for (int m = 1; m < F; ++m) {
            const T4 xm = ring[(2 * u + m) % R];
            sa01 = sa01 + a.tp.h[m] * T2{xm.x, xm.y};
            da01 = da01 + gq(F - 1 - m) * T2{xm.x, xm.y};
            sa23 = sa23 + a.tp.h[m] * T2{xm.z, xm.w};
            da23 = da23 + gq(F - 1 - m) * T2{xm.z, xm.w};
        }
        T2 *const w1 = x1 + (t & 1) * rows1;
#if WL_LONG_SPLIT
        if constexpr (F <= 14) {
        *reinterpret_cast<T4 *>(w1 + 2 * lp) = T4{sa01.x, da01.x, sa01.y, da01.y};
        *reinterpret_cast<T4 *>(w1 + 2 * (NU + lp)) = T4{sa23.x, da23.x, sa23.y, da23.y};
        } else
#endif
        {
        *reinterpret_cast<T4 *>(w1 + 4 * lp) = T4{sa01.x, da01.x, sa01.y, da01.y};
        *reinterpret_cast<T4 *>(w1 + 4 * lp + 2) = T4{sa23.x, da23.x, sa23.y, da23.y};
        }
        wg_lds_sync(true);
        __builtin_amdgcn_sched_barrier(0);
        if (helper) return;                                // the helper wave owns no output
        // ---- dim-1 pass: window rows 4L' .. 4L'+WIN-1 as {A, B} pairs ----
        T2 E[WIN];
#pragma unroll
        for (int c = 0; c < WIN / 2; ++c) {
            const T4 v = (WL_LONG_SPLIT && F <= 14) ? *reinterpret_cast<const T4 *>(w1 + 2 * ((c & 1) * NU + lp + (c >> 1)))
                                                    : *reinterpret_cast<const T4 *>(w1 + 4 * lp + 2 * c);
            E[2 * c] = T2{v.x, v.y};
            E[2 * c + 1] = T2{v.z, v.w};
        }
        T2 P[2], Q[2];                                 // P[q] = {ss, sd} of row ko + q;  Q[q] = {ds, dd} of row kod + q
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            T2 s = a.tp.h[0] * E[2 * q];
#pragma unroll
            for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * E[2 * q + m];
            T2 d = gq(F - 1) * E[2 * q + 2 * DSH + 2 - F];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + gq(m) * E[2 * q + 2 * DSH + 1 - m];
            P[q] = s;
            Q[q] = d;
        }
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        // even lane: ss rows ko..ko+3 and ds rows kod..kod+3 of column k;  odd lane: sd / dd of column kd
        T rP[2], rQ[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            rP[q] = from_partner(odd ? P[q].x : P[q].y);
            rQ[q] = from_partner(odd ? Q[q].x : Q[q].y);
        }
        T *const ck = yb + k * a.ldy, *const ckd = yb + (nxj + kd) * a.ldy, *const cl = llb + k * ldl;      // (uniform)
        if (!odd) {
            *reinterpret_cast<T4 *>(cl + ko) = T4{P[0].x, P[1].x, rP[0], rP[1]};
            store_pol<WL_P_LONG_ST>(reinterpret_cast<T4 *>(ck + (hmi + kod)), T4{Q[0].x, Q[1].x, rQ[0], rQ[1]});
        } else {
            store_pol<WL_P_LONG_ST>(reinterpret_cast<T4 *>(ckd + (ko - 2)), T4{rP[0], rP[1], P[0].y, P[1].y});
            store_pol<WL_P_LONG_ST>(reinterpret_cast<T4 *>(ckd + (hmi + kod - 2)), T4{rQ[0], rQ[1], Q[0].y, Q[1].y});
        }
    };

    for (int t0 = 0; t0 < S; t0 += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // workgroup-uniform guard: every wave takes the same barriers.  An exit, not a skipped step: the control-flow graph
            // then holds no path "step skipped, later step taken", which tools/isa_check.py would have to (and cannot) rule out
            if (t0 + u >= S) return;
            step(t0 + u, u, true);
        }
    }
}

// ------------------------------------------------------------------------------------------
bool fwd2d_long_ok(int F, int64_t ms, int64_t ns)
{
    if (F < 12 || F > 20 || (F & 1)) return false;
    if (ms >= ((int64_t)1 << 30)) return false;
    // exact tiling only: strips of 256 rows per main wave; columns: pairs, at least one full window
    return ms >= 256 && (ms % 256) == 0 && ns >= 32 && (ns % 2) == 0;
}

template <int F>
static hipError_t launch_long_f(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                                float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    // ring slots: 20 up to 16 taps, 24 above -- fewer VGPRs for the shorter filters (20 slots keep them under
    // 170 VGPRs = 3 waves per SIMD; 16 taps then request two steps ahead instead of four: 175 -> 173 us); the unrolled body covers R / 2 steps, any chunk length (guarded steps)
    constexpr int R = (F <= 16) ? 20 : 24;
    typedef LongGeom<F> G;
    LdsLongArgs<F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    // strips of 1024 rows by default (8192^2 level: db6 146 us with W = 4, 172 with W = 2; db8 165 / 202; two passes: 220 / 218)
    int W = (int)opt("WL_LONG_W", 0);
    if (W != 1 && W != 2 && W != 4) W = (ms >= 2048) ? 4 : 1;   // small levels: more, narrower strips (1024^2: 10.9 us with W = 1, 12.9 with W = 4)
    while (W > 1 && (ms % (256 * W)) != 0) W >>= 1;
    a.npl = 64 * W;
    a.nstrips = (int)(ms / (256 * W));
    // (a chunk's ring fill -- R - 2 columns -- is pure overhead: 256 columns where one workgroup per CU remains; r04: db8 175 -> 169 us)
    int TJ = (int)opt("WL_LONG_TJ", 256);
    if (TJ < 16) TJ = 16;
    TJ &= ~1;
    auto nwgs = [&](int tj) { return (int64_t)a.nstrips * ((ns + tj - 1) / tj); };
    while (TJ > 16 && nwgs(TJ) < (int64_t)cu_count * opt("WL_LONG_WG_PER_CU", 1)) TJ = (TJ / 2) & ~1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.rev = (!lvl1 && opt("WL_REVERSE", 1)) ? 1 : 0;
    // (measured r04, 8192^2 level: helper first db8 175 -> 192 us, db10 200 -> 253 -- it shares a SIMD with a main wave and starves
    //  it; main waves first: neutral.  Unlike the pair kernels, whose helper carries a quarter of a main wave's work.)
    a.prio = (int)opt("WL_LONG_PRIO", 0);
    a.tp = shrink<float, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    const int nthreads = 64 * (W + 1);
    const size_t shmem = (size_t)2 * (4 * nthreads + G::WIN + 4) * 8;
    if (lvl1) hipLaunchKernelGGL((k_fwd2d_lds_long<F, R, 1>), dim3(nwg), dim3(nthreads), shmem, st, a);
    else hipLaunchKernelGGL((k_fwd2d_lds_long<F, R, 0>), dim3(nwg), dim3(nthreads), shmem, st, a);
    return hipGetLastError();
}

hipError_t fwd2d_long_launch(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                             float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count)
{
    switch (taps.F) {
    case 12: return launch_long_f<12>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 14: return launch_long_f<14>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 16: return launch_long_f<16>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 18: return launch_long_f<18>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    case 20: return launch_long_f<20>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
