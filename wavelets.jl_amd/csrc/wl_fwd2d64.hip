// wl_fwd2d64.hip -- forward 2-D filter-bank levels, Float64, F <= 10 even: the LDS-exchange streaming kernel of wl_fwd2d.hip
// with TWO rows per lane (one 16-byte load per lane and column holds two Float64 rows).
//
//   k_fwd2d_lds64<F, LVL1>   one fused 2-D level per launch (dim-2 pass in registers, dim-1 pass on windows read back from
//                            a two-slot LDS exchange), exact tiling only: W main waves of 128 rows + a helper wave for the
//                            8 halo rows above the strip.  Replaces the round-1 overlapped-strip kernel (k_fwd2d_stream,
//                            DPP neighbour exchange) for blocks whose rows tile into strips of 128 / 256 / 512.
//
// Lane L' (rows 2L', 2L'+1 of the strip) produces s row L' from window rows 2L' .. 2L'+F-1 and d row L'+4 from window rows
// 2L'+10-F .. 2L'+9 (d[k] uses x[2k+2-F .. 2k+1]; details are stored shifted by 4), so every window lies in [2L', 2L'+10):
// ten aligned ds_read_b128 of {scaling, detail} pairs.  A lane pair owns two consecutive s rows and two consecutive d rows:
// the partner exchange (DPP quad_perm) makes every store 16 bytes.
// Arithmetic: closed forms of wl_internal.h in the reference's order, no FMA -- bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <int F>
struct Lds2DArgs64 {
    const double *src; int64_t lds;
    double *y; int64_t ldy;
    double *ll; int64_t ldll;         // approximation: next stage's input buffer, or y itself
    int64_t ms, ns;                   // level-l block
    int TJ;                           // owned input columns per chunk (multiple of 16)
    int nstrips, nchunks;
    int npl;                          // owned lanes per workgroup (64 W)
    int rev;
    int64_t bs_src, bs_y, bs_ll; int nll;    // batch of independent blocks over blockIdx.y (planes of a 3-D level)
    TapsF<double, F> tp;
};

template <int F, int LVL1>
__global__ void __launch_bounds__(320, 3) k_fwd2d_lds64(Lds2DArgs64<F> a)
{
    typedef double T;
    typedef D2 T2;
    constexpr int SH = (F - 2) / 2;
    constexpr int R = 16, U = 8, PFD = (R - F) / 2;
    constexpr int HL = 4;                             // halo lanes: 8 rows above the strip
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    // g[m] = (-1)^m h[m] exactly: only the scaling taps occupy SGPRs, a detail term multiplies by the negated tap (a source modifier)
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };
    const int nthreads = blockDim.x;
    const int lp = threadIdx.x;                       // L': lane index within the workgroup's strip
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t first = xcd * q8 + (xcd < r8 ? xcd : r8), cnt = q8 + (xcd < r8 ? 1u : 0u);
    uint32_t logical = first + (b >> 3);
    if (a.rev) logical = first + (cnt - 1 - (logical - first));
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);

    // LDS: exchange rows [2][2*nthreads + 16] of {A, B}
    const int rows1 = 2 * nthreads + 16;
    T2 *const x1 = reinterpret_cast<T2 *>(smem_raw);

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1;
    const int msi = (int)ms, hmi = msi >> 1;
    const int gi = strip * (2 * a.npl) + 2 * lp;      // first row of this lane (halo lanes may exceed ms: wrap)
    int row = gi;
    if (row >= msi) row -= msi;
    const bool loader = lp < a.npl + HL;
    const bool helper = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == (nthreads >> 6) - 1;
    const int ko = gi >> 1;
    int kod = ko + 4;  if (kod >= hmi) kod -= hmi;    // the d row of this lane
    const bool odd = (lp & 1) != 0;

    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S = (int)((jend - j0) >> 1);            // steps = output columns of this chunk (multiple of 8)
    // lanes that hold no input rows (the helper's upper lanes, lanes past the strip) load the first rows of the same column instead
    // of being masked: one extra cache line per column, and no `if` -- hence no phi and no register copy -- around the
    // asynchronous loads (wl_dev.h: gload16_if)
    const T *base = a.src + (int64_t)blockIdx.y * a.bs_src + (loader ? row : 0);

    T2 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T2{0.0, 0.0};
#pragma unroll
    for (int c = 0; c < R - 2; ++c) {
        int64_t jc = j0 + c;
        if (jc >= ns) jc -= ns;
        gload16<WL_P_LDS64_LD != 0>(ring[c], base + jc * a.lds);
    }
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);
    T *const yb = a.y + (int64_t)blockIdx.y * a.bs_y;
    const bool to_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    T *const llb = to_ll ? (a.ll + (int64_t)blockIdx.y * a.bs_ll) : yb;
    const int64_t ldl = to_ll ? a.ldll : a.ldy;
    const int64_t kbase = j0 >> 1;

    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch) {                                    // (compile-time)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                gload16<WL_P_LDS64_LD != 0>(ring[(2 * u + R - 2 + e) % R], base + jc * a.lds);
            }
        }
        // loads only in the count (see wl_dev.h: stores may be acknowledged before an older load returns)
        if (prefetch) wait_vm<2 * PFD>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
        else wait_vm<0>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
        // ---- dim-2 pass on the lane's two rows: {A, B}[r] = scaling / detail (column k / kd) of row r ----
        T2 sa = a.tp.h[0] * ring[(2 * u) % R];
        T2 da = gq(F - 1) * ring[(2 * u) % R];
#pragma unroll
        for (int m = 1; m < F; ++m) {
            const T2 xm = ring[(2 * u + m) % R];
            sa = sa + a.tp.h[m] * xm;
            da = da + gq(F - 1 - m) * xm;
        }
        T2 *const w1 = x1 + (t & 1) * rows1;
        w1[2 * lp] = T2{sa.x, da.x};
        w1[2 * lp + 1] = T2{sa.y, da.y};
        wg_lds_sync(true);
        __builtin_amdgcn_sched_barrier(0);
        if (helper) return;                                // the helper wave owns no output
        // ---- dim-1 pass: window rows 2L' .. 2L'+9 as {A, B} pairs ----
        T2 E[10];
#pragma unroll
        for (int c = 0; c < 10; ++c) E[c] = w1[2 * lp + c];
        T2 P = a.tp.h[0] * E[0];                       // {ss, sd} of row ko
#pragma unroll
        for (int m = 1; m < F; ++m) P = P + a.tp.h[m] * E[m];
        T2 Q = gq(F - 1) * E[10 - F];              // {ds, dd} of row kod
#pragma unroll
        for (int m = F - 2; m >= 0; --m) Q = Q + gq(m) * E[9 - m];
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        // even lane: ss rows ko, ko+1 and ds rows kod, kod+1 of column k;  odd lane: sd / dd of column kd
        const T rP = from_partner(odd ? P.x : P.y);
        const T rQ = from_partner(odd ? Q.x : Q.y);
        T *const ck = yb + k * a.ldy, *const ckd = yb + (nxj + kd) * a.ldy, *const cl = llb + k * ldl;      // (uniform)
        if (!odd) {
            *reinterpret_cast<T2 *>(cl + ko) = T2{P.x, rP};
            store_pol<WL_P_LDS64_ST>(reinterpret_cast<T2 *>(ck + (hmi + kod)), T2{Q.x, rQ});
        } else {
            store_pol<WL_P_LDS64_ST>(reinterpret_cast<T2 *>(ckd + (ko - 1)), T2{rP, P.y});
            store_pol<WL_P_LDS64_ST>(reinterpret_cast<T2 *>(ckd + (hmi + kod - 1)), T2{rQ, Q.y});
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
bool fwd2d_lds64_ok(int F, int64_t ms, int64_t ns)
{
    if (F < 2 || F > 10 || (F & 1)) return false;
    if (ms >= ((int64_t)1 << 30)) return false;
    // exact tiling only: strips of 128 rows per main wave; columns: chunks of 16
    return ms >= 128 && (ms % 128) == 0 && ns >= 16 && (ns % 16) == 0;
}

template <int F>
static hipError_t launch_lds64_f(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds,
                                 double *y, int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                                 int64_t nbatch, int64_t bs_src, int64_t bs_y, int64_t bs_ll, int nll)
{
    Lds2DArgs64<F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.bs_src = bs_src; a.bs_y = bs_y; a.bs_ll = bs_ll; a.nll = nll;
    int W = (int)opt("WL_LDS_W", 4);
    if (W != 1 && W != 2 && W != 4) W = 4;
    while (W > 1 && (ms % (128 * W)) != 0) W >>= 1;
    a.npl = 64 * W;
    a.nstrips = (int)(ms / (128 * W));
    int TJ = (int)opt("WL_TJ", 128);
    if (TJ < 16 || (TJ % 16) != 0) TJ = 128;
    auto nwaves = [&](int tj) { return (int64_t)a.nstrips * (W + 1) * ((ns + tj - 1) / tj) * nbatch; };
    while (TJ > 32 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * opt("WL_WAVES_PER_CU", 8)) TJ >>= 1;
    while (TJ > 16 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * opt("WL_WAVES_MIN", 8)) TJ >>= 1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.rev = (!lvl1 && opt("WL_REVERSE", 1)) ? 1 : 0;
    a.tp = shrink<double, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    const int nthreads = 64 * (W + 1);
    const size_t shmem = (size_t)2 * (2 * nthreads + 16) * 16;
    if (lvl1) hipLaunchKernelGGL((k_fwd2d_lds64<F, 1>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);
    else hipLaunchKernelGGL((k_fwd2d_lds64<F, 0>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);
    return hipGetLastError();
}

hipError_t fwd2d_lds64_launch(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds,
                              double *y, int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                              int64_t nbatch, int64_t bs_src, int64_t bs_y, int64_t bs_ll, int nll)
{
    switch (taps.F) {
    case 2: return launch_lds64_f<2>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 4: return launch_lds64_f<4>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 6: return launch_lds64_f<6>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 8: return launch_lds64_f<8>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 10: return launch_lds64_f<10>(st, taps, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
