// wl_fwd2d.hip -- forward 2-D filter-bank levels, Float32, F <= 10 even: the LDS-exchange streaming kernel.
//
//   k_fwd2d_lds<F, LVL1>   one fused 2-D level per launch (the two-level kernel built on it lives in wl_pair2d.hip).
//
// Same marching scheme as k_fwd2d_stream (wl_fwd.hip): a wave owns 256 rows (4 per lane, one 16-byte load per lane
// per column = 1 KiB per wave-instruction) and walks the columns of a chunk with a 16-slot column ring in VGPRs;
// the dim-2 pass (reference "rows", transforms_filter.jl:161-168) never leaves registers.  What changed is the
// dim-1 pass (reference "columns", :169-172): the lanes of a workgroup publish their dim-2 results in LDS as
// {scaling, detail} pairs and every lane reads its 12-row window back with three aligned ds_read_b128 per component
// pair.  Measured on MI355X (tools/probes/valu_probe.hip): a DPP operand costs 5.5-6.6 issue cycles against 2.25 for a
// plain VALU op, and the DPP version needed another ~60 register moves per step to build operand pairs -- together
// about 40 % of the old kernels' issue slots.  LDS reads are not VALU work and the windows come back already paired.
//
// One-sided halo.  Lane L' (rows 4L'..4L'+3 of the workgroup's strip) produces
//     s rows 2L', 2L'+1          from window rows 4L' .. 4L'+F+1
//     d rows 2L'+4, 2L'+5        from window rows 4L'+10-F .. 4L'+11        (d[k] uses x[2k+2-F .. 2k+1])
// so every window lies in [4L', 4L'+12): nothing is needed from below the strip, 8 rows from above it, and the d rows
// of a lane pair (2i, 2i+1) are again four consecutive, 16-byte aligned rows.
//
// Workgroup shapes (launcher): blockDim = 64 * NW waves, NPL = owned lanes (pitch = 4*NPL rows), lanes below
// NPL + 2 load data (2 halo lanes).
//   NW = 1, NPL = 64 - HL - ...   overlapped single-wave strips: no barrier at all (LDS used wave-privately)
//   NW = W + 1, NPL = 64 W        exact tiling: W full waves whose global accesses are all 1 KiB aligned lines, plus a
//                                 helper wave that loads only the 8 halo rows and feeds them into the exchange
// Arithmetic is the closed form of wl_internal.h, bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <int F>
struct Lds2DArgs {
    const float *src; int64_t lds;
    float *y; int64_t ldy;
    float *ll; int64_t ldll;          // approximation: next stage's input buffer, or y itself
    int64_t ms, ns;                   // level-l block
    int TJ;                           // owned input columns per chunk (multiple of 16)
    int nstrips, nchunks;
    int npl;                          // owned lanes per workgroup
    int nload;                        // lanes that load input rows (npl + halo lanes, <= blockDim)
    int rev;
    int helper;                       // 1: the workgroup's last wave only supplies halo rows (exact tiling)
    int prio;                         // 1: the helper wave issues ahead of the main waves (s_setprio; see wl_pair2d.hip)
    int64_t bs_src, bs_y, bs_ll; int nll;    // batch of independent blocks over blockIdx.y (planes of a 3-D level)
    int src_mod; int64_t spin0;       // > 0: plane p reads copy (spin0 + p) % src_mod with its columns rotated by (spin0 + p) / src_mod (SrcView)
    int th; double t_unit, sigma_host; const double *mad_dev;     // TH instances: threshold the details at the store (SrcView)
    TapsF<float, F> tp;
};

template <int F, int LVL1, int TH = 0>
__global__ void __launch_bounds__(320, 3) k_fwd2d_lds(Lds2DArgs<F> a)
{
    typedef float T;
    typedef float T2 __attribute__((ext_vector_type(2)));
    typedef float T4 __attribute__((ext_vector_type(4)));
    constexpr int SH = (F - 2) / 2;
    constexpr int R = 16, U = 8, PFD = (R - F) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    // g[m] = (-1)^m h[m] exactly: only the scaling taps occupy SGPRs, a detail term multiplies by the negated tap (a source modifier)
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    const int nthreads = blockDim.x;
    const bool multi = nthreads > 64;
    const int lp = threadIdx.x;                       // L': lane index within the workgroup's strip
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1u : 0u);
    uint32_t logical = first + (b >> 3);
    if (a.rev) logical = first + (cnt - 1 - (logical - first));
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);

    // LDS: exchange rows [2][4*nthreads + 16] of T2
    const int rows1 = 4 * nthreads + 16;
    T2 *const x1 = reinterpret_cast<T2 *>(smem_raw);

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1, hm = ms >> 1;
    // row indices are 32-bit (the launcher requires ms < 2^30): per-lane offsets stay in one VGPR, column bases in SGPRs
    const int msi = (int)ms, hmi = (int)hm;
    const int gi = strip * (4 * a.npl) + 4 * lp;                             // first row of this lane (may exceed ms: wraps)
    int row = gi;
    if (row >= msi) row -= msi;
    if (row >= msi) row = 0;                                                 // (lanes far beyond the array: never used)
    const bool loader = lp < a.nload;
    // exact tiling: the last wave is the halo helper (wave-uniform, kept in an SGPR)
    const bool helper = a.helper && (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == (nthreads >> 6) - 1);
    if (a.prio && helper) __builtin_amdgcn_s_setprio(2);
    const bool own = (lp < a.npl) && (gi < msi);
    const int ko = gi >> 1;
    int kod = ko + 4;  if (kod >= hmi) kod -= hmi;                           // first d row of this lane
    const bool odd = (lp & 1) != 0;

    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S = (int)((jend - j0) >> 1);                       // steps = output columns of this chunk (multiple of 8)
    // virtual circular shift along dim 2 (translation-invariant denoise): column j of the plane is column j - crot of its source
    int64_t splane = blockIdx.y, crot = 0;
    if (a.src_mod > 0) {
        const int64_t spin = a.spin0 + (int64_t)blockIdx.y;
        splane = spin % a.src_mod;
        crot = spin / a.src_mod;
    }
    // lanes that hold no input rows (the helper's upper lanes, lanes past the strip) load the first rows of the same column instead
    // of being masked: one extra cache line per column, and no `if` -- hence no phi and no register copy -- around the
    // asynchronous loads (wl_dev.h: gload16_if)
    const T *base = a.src + splane * a.bs_src + (loader ? row : 0);

    if (helper) {
        // ---- the halo wave of an exactly tiled strip: the 8 rows below it, ONE row per lane in a scalar column ring (with four rows
        //      per lane two lanes did useful work while the wave issued a full wave's dim-2 pass every step; see wl_pair2d.hip).
        //      Same loads / waits / one barrier per step as the main waves; lanes >= 8 load row 0 and write nothing. ----
        const int hl = (int)(threadIdx.x & 63);
        int hrow = strip * (4 * a.npl) + 4 * a.npl + hl;
        if (hrow >= msi) hrow -= msi;
        if (hrow >= msi) hrow = 0;
        const T *hbase = a.src + splane * a.bs_src + ((hl < 8) ? hrow : 0);
        T hring[R];
#pragma unroll
        for (int c = 0; c < R; ++c) hring[c] = 0.f;
#pragma unroll
        for (int c = 0; c < R - 2; ++c) {
            int64_t jc = j0 + c;
            if (jc >= ns) jc -= ns;
            jc -= crot;
            if (jc < 0) jc += ns;
            gload4(hring[c], hbase + jc * a.lds);
        }
#pragma unroll
        for (int c = 0; c < R; c += 2) wait_vm<0>(hring[c], hring[c + 1]);
        auto hstep = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
            if (prefetch) {
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    int64_t jc = j0 + 2 * t + (R - 2) + e;
                    if (jc >= ns) jc -= ns;
                    if (jc >= ns) jc -= ns;
                    jc -= crot;
                    if (jc < 0) jc += ns;
                    gload4(hring[(2 * u + R - 2 + e) % R], hbase + jc * a.lds);
                }
                wait_vm<2 * PFD>(hring[(2 * u + F - 2) % R], hring[(2 * u + F - 1) % R]);
            } else {
                wait_vm<0>(hring[(2 * u + F - 2) % R], hring[(2 * u + F - 1) % R]);
            }
            T sa = a.tp.h[0] * hring[(2 * u) % R], da = gq(F - 1) * hring[(2 * u) % R];
#pragma unroll
            for (int m = 1; m < F; ++m) {
                sa = sa + a.tp.h[m] * hring[(2 * u + m) % R];
                da = da + gq(F - 1 - m) * hring[(2 * u + m) % R];
            }
            if (hl < 8) (x1 + (t & 1) * rows1)[4 * a.npl + hl] = T2{sa, da};
            wg_lds_sync(multi);
        };
        int t0 = 0;
        for (; t0 < S - U; t0 += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) hstep(t0 + u, u, true);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) hstep(t0 + u, u, u < U - PFD);
        return;
    }

    T4 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        jc -= crot;
        if (jc < 0) jc += ns;
        gload16<WL_P_LDS_LD != 0>(ring[c], base + jc * a.lds);
    }
    // the first steps find all 14 prologue columns complete (one full wait per wave, once)
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);
    // The newest two columns of step t's window (2t+F-2, 2t+F-1) were requested PFD steps earlier; when step t needs them,
    // 2*PFD younger LOADS are behind them (two per step, this step's included) -- see the wait in step().
    T *const yb = a.y + (int64_t)blockIdx.y * a.bs_y;
    const bool to_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    T *const llb = to_ll ? (a.ll + (int64_t)blockIdx.y * a.bs_ll) : yb;
    const int64_t ldl = to_ll ? a.ldll : a.ldy;
    const int64_t kbase = j0 >> 1;                     // multiple of 8
    // TH: threshold!(xt, th, sigma * dnt.t) at the stores -- a Float32 cut per coefficient, Float64 only for what survives the cut
    // of soft / semisoft / Stein, in a wave-uniform branch (ThCut, wl_dev.h).  (Round 3 applied the general threshold_one in
    // Float64 to every coefficient: this kernel, VALU-bound with 10 taps, became slower than the separate pass it was to save.)
    ThCut cut = th_make_cut(-1, 0.0);
    if constexpr (TH != 0) cut = th_make_cut(a.th, ((a.sigma_host >= 0) ? a.sigma_host : (*a.mad_dev / 0.6745)) * a.t_unit);

    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {                                    // (compile-time)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                jc -= crot;
                if (jc < 0) jc += ns;
                gload16<WL_P_LDS_LD != 0>(ring[(2 * u + R - 2 + e) % R], base + jc * a.lds);
            }
        }
        if (prefetch) {
            // loads only: stores may be acknowledged before an older load returns, so they must not be counted as "younger
            // operations still outstanding" (see wl_pair2d.hip; round 2 also counted 4 stores per step here: 104.8-105.0 us
            // against 105.2-106.2 us for level 1 of 8192^2 -- not worth a wrong column once in a thousand transforms)
            wait_vm<2 * PFD>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
        } else {
            wait_vm<0>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);      // (last steps of the chunk: nothing left to overlap)
        }
        // ---- level l, dim-2 pass on row pairs: {A, B}[r] = scaling / detail (column k / kd) of row r ----
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
        *reinterpret_cast<T4 *>(w1 + 4 * lp) = T4{sa01.x, da01.x, sa01.y, da01.y};
        *reinterpret_cast<T4 *>(w1 + 4 * lp + 2) = T4{sa23.x, da23.x, sa23.y, da23.y};
        wg_lds_sync(multi);
        __builtin_amdgcn_sched_barrier(0);
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
        if constexpr (TH != 0) {
            // the three detail components of every row (sd, ds, dd) are final coefficients: threshold!(xt, th, sigma * t)
            bool above = false;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                P[q].y = th_cut(cut, P[q].y); Q[q].x = th_cut(cut, Q[q].x); Q[q].y = th_cut(cut, Q[q].y);
                if (!to_ll) P[q].x = th_cut(cut, P[q].x);                 // (last level: the approximation is final too)
                above = above || P[q].y != 0.f || Q[q].x != 0.f || Q[q].y != 0.f || (!to_ll && P[q].x != 0.f);
            }
            if (th_needs_exact(cut, above || cut.tf < 0.f)) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    P[q].y = th_exact(cut, P[q].y); Q[q].x = th_exact(cut, Q[q].x); Q[q].y = th_exact(cut, Q[q].y);
                    if (!to_ll) P[q].x = th_exact(cut, P[q].x);
                }
            }
        }
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        {
            // even lane: ss rows ko..ko+3 and ds rows kod..kod+3 of column k;  odd lane: sd / dd of column kd
            T rP[2], rQ[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                rP[q] = from_partner(odd ? P[q].x : P[q].y);
                rQ[q] = from_partner(odd ? Q[q].x : Q[q].y);
            }
            if (own) {
                T *const ck = yb + k * a.ldy, *const ckd = yb + (nxj + kd) * a.ldy, *const cl = llb + k * ldl;      // (uniform)
                if (!odd) {
                    *reinterpret_cast<T4 *>(cl + ko) = T4{P[0].x, P[1].x, rP[0], rP[1]};
                    store_pol<WL_P_LDS_ST>(reinterpret_cast<T4 *>(ck + (hmi + kod)), T4{Q[0].x, Q[1].x, rQ[0], rQ[1]});
                } else {
                    store_pol<WL_P_LDS_ST>(reinterpret_cast<T4 *>(ckd + (ko - 2)), T4{rP[0], rP[1], P[0].y, P[1].y});
                    store_pol<WL_P_LDS_ST>(reinterpret_cast<T4 *>(ckd + (hmi + kod - 2)), T4{rQ[0], rQ[1], Q[0].y, Q[1].y});
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

// ------------------------------------------------------------------------------------------
// Workgroup shape for a block of `ms` rows: returns waves per workgroup, owned lanes and loader lanes.
struct Shape2D { int nw, npl, nload, nstrips, helper; };
static Shape2D pick_shape(int64_t ms, int mode, int wmain)
{
    const int HL = 2;
    Shape2D s;
    s.helper = 0;
    if (mode == 0) {
        // exact tiling: W full waves + a helper wave for the halo rows (W = the largest of 4, 2, 1 whose strip divides ms)
        int W = wmain;
        while (W > 1 && (ms % (256 * W)) != 0) W >>= 1;
        if ((ms % (256 * W)) == 0) {
            s.nw = W + 1; s.npl = 64 * W; s.nload = s.npl + HL; s.nstrips = (int)(ms / (256 * W)); s.helper = 1;
            return s;
        }
    }
    // overlapped strips of nw waves; the pitch is kept a multiple of 32 rows (128 B): misaligned strips measured
    // 15 % slower in pure data movement (tools/probes/march_probe.hip)
    const int nw = (mode >= 1 && mode <= 4) ? mode : 1;
    s.nw = nw;
    s.npl = ((64 * nw - HL) / 8) * 8;
    if (4 * (int64_t)s.npl > ms) s.npl = (int)(ms / 4);
    s.nload = s.npl + HL;
    if (s.nload > 64 * nw) s.nload = 64 * nw;
    s.nstrips = (int)((ms + 4 * s.npl - 1) / (4 * s.npl));
    return s;
}

template <int F>
static hipError_t launch_lds_f(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds,
                               float *y, int64_t ldy, float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                               int64_t nbatch, int64_t bs_src, int64_t bs_y, int64_t bs_ll, int nll, int src_mod, int64_t spin0,
                               const SrcView *thresh)
{
    Lds2DArgs<F> a;
    a.th = thresh ? thresh->th : -1; a.t_unit = thresh ? thresh->t_unit : 0.0; a.sigma_host = thresh ? thresh->sigma_host : 0.0;
    a.mad_dev = thresh ? thresh->mad_dev : nullptr;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.bs_src = bs_src; a.bs_y = bs_y; a.bs_ll = bs_ll; a.nll = nll; a.src_mod = src_mod; a.spin0 = spin0;
    const Shape2D sh = pick_shape(ms, (int)opt("WL_LDS_MODE", 0), (int)opt("WL_LDS_W", 4));
    a.npl = sh.npl; a.nload = sh.nload; a.nstrips = sh.nstrips; a.helper = sh.helper;
    a.prio = opt("WL_LDS_PRIO", 1) != 0 ? 1 : 0;
    int TJ = (int)opt("WL_TJ", 128);
    if (TJ < 16 || (TJ % 16) != 0) TJ = 128;              // (a test knob must not be able to divide by zero)
    auto nwaves = [&](int tj) { return (int64_t)a.nstrips * sh.nw * ((ns + tj - 1) / tj) * nbatch; };
    const int wpc = (int)opt("WL_WAVES_PER_CU", 8);
    while (TJ > 32 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * wpc) TJ >>= 1;
    while (TJ > 16 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * opt("WL_WAVES_MIN", 8)) TJ >>= 1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.rev = (!lvl1 && opt("WL_REVERSE", 1)) ? 1 : 0;
    a.tp = shrink<float, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    const int nthreads = 64 * sh.nw;
    const size_t shmem = (size_t)2 * (4 * nthreads + 16) * 8;
    if (thresh && lvl1) hipLaunchKernelGGL((k_fwd2d_lds<F, 1, 1>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);
    else if (thresh) hipLaunchKernelGGL((k_fwd2d_lds<F, 0, 1>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);
    else if (lvl1) hipLaunchKernelGGL((k_fwd2d_lds<F, 1>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);
    else hipLaunchKernelGGL((k_fwd2d_lds<F, 0>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);
    return hipGetLastError();
}

bool fwd2d_lds_ok(int F, int nlev, int64_t ms, int64_t ns)
{
    if (nlev == 2) return fwd2d_pair_ok(F, ms, ns);
    if (F < 2 || F > 10 || (F & 1)) return false;
    // rows: lanes own 4 rows, d rows wrap in groups of 4; columns: chunks of 16
    if (ms >= ((int64_t)1 << 30)) return false;
    return ms >= 64 && (ms % 8) == 0 && ns >= 16 && (ns % 16) == 0;
}

hipError_t fwd2d_lds_launch(hipStream_t st, const Taps<float> &taps, int nlev, bool lvl1, const float *src, int64_t lds,
                            float *y, int64_t ldy, float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                            int64_t nbatch, int64_t bs_src, int64_t bs_y, int64_t bs_ll, int nll, int src_mod, int64_t spin0,
                            const SrcView *thresh)
{
    if (nlev == 2) {
        if (nbatch != 1) return hipErrorInvalidValue;
        return fwd2d_pair_launch(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count);
    }
    switch (taps.F) {
    case 2: return launch_lds_f<2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll, src_mod, spin0, thresh);
    case 4: return launch_lds_f<4>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll, src_mod, spin0, thresh);
    case 6: return launch_lds_f<6>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll, src_mod, spin0, thresh);
    case 8: return launch_lds_f<8>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll, src_mod, spin0, thresh);
    case 10: return launch_lds_f<10>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll, src_mod, spin0, thresh);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
