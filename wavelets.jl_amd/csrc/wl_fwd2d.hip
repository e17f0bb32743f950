// wl_fwd2d.hip -- forward 2-D filter-bank levels, Float32, F <= 10 even: the LDS-exchange streaming kernel.
//
//   k_fwd2d_lds<F, NLEV>   one (NLEV = 1) or two (NLEV = 2) fused 2-D levels per launch.
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
// of a lane pair (2i, 2i+1) are again four consecutive, 16-byte aligned rows.  Level 2 (NLEV = 2) does the same on the
// level-1 approximation, which stays in an 8-slot register ring (two rows per lane): s2 row L', d2 row L'+4, window =
// approximation rows 2L' .. 2L'+9, i.e. 24 input rows above the strip in total.
//
// Workgroup shapes (launcher): blockDim = 64 * NW waves, NPL = owned lanes (pitch = 4*NPL rows), lanes below
// NPL + HL load data (HL = 2 / 6 halo lanes for NLEV = 1 / 2).
//   NW = 1, NPL = 64 - HL - ...   overlapped single-wave strips: no barrier at all (LDS used wave-privately)
//   NW = W + 1, NPL = 64 W        exact tiling: W full waves whose global accesses are all 1 KiB aligned lines, plus a
//                                 helper wave that loads only the 8 / 24 halo rows and feeds them into the exchange
// Arithmetic is the closed form of wl_internal.h, bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <int F>
struct Lds2DArgs {
    const float *src; int64_t lds;
    float *y; int64_t ldy;
    float *ll; int64_t ldll;          // approximation after NLEV levels: next stage's input buffer, or y itself
    int64_t ms, ns;                   // level-l block
    int TJ;                           // owned input columns per chunk (multiple of 16)
    int nstrips, nchunks;
    int npl;                          // owned lanes per workgroup
    int nload;                        // lanes that load input rows (npl + halo lanes, <= blockDim)
    int rev;
    int helper;                       // 1: the workgroup's last wave only supplies halo rows (exact tiling)
    int64_t bs_src, bs_y, bs_ll; int nll;    // batch of independent blocks over blockIdx.y (planes of a 3-D level)
    TapsF<float, F> tp;
};

__device__ __forceinline__ void wg_lds_sync(bool multi)
{
    if (multi) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// LVL1 only gives the launch that consumes the full-size input its own symbol (rocprofv3 --stats then reports the dominant
// kernel separately from the same code running on the smaller levels).
// Column loads and their waits are written by hand.  hipcc's own wait-count insertion, given the rotating 16-slot ring, puts
// vmcnt(1) / vmcnt(0) in front of two of every eight steps (checked in the ISA: tools/probes/waitcnt_probe.hip has the small
// reproduction): the wave then waits for the loads it issued a few instructions earlier AND for all of its stores, twice per
// iteration -- the four-step prefetch distance never exists.  Here the load is opaque to the compiler and the wait names the
// two ring slots it guards, so every consumer depends on the wait through its data.
typedef float F4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gload16(F4 &dst, const float *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
// "at most N vector-memory operations outstanding": loads and stores retire in issue order on gfx9-family counters, so this
// covers every load that has at least N younger operations behind it.
template <int N>
__device__ __forceinline__ void wait_vm(F4 &a, F4 &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

template <int F, int NLEV, int LVL1>
__global__ void __launch_bounds__(320, 3) k_fwd2d_lds(Lds2DArgs<F> a)
{
    typedef float T;
    typedef float T2 __attribute__((ext_vector_type(2)));
    typedef float T4 __attribute__((ext_vector_type(4)));
    constexpr int SH = (F - 2) / 2;
    constexpr int R = 16, U = 8, PFD = (R - F) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

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

    // LDS: level-1 exchange rows [2][4*nthreads + 16] of T2, then level-2 exchange rows [2][2*nthreads + 16] of T2
    const int rows1 = 4 * nthreads + 16, rows2 = 2 * nthreads + 16;
    T2 *const x1 = reinterpret_cast<T2 *>(smem_raw);
    T2 *const x2 = x1 + 2 * rows1;
    // level-1 approximation ring (NLEV = 2): column k's rows (ko, ko+1) of every lane in slot k % 8 -- private to the lane
    // that wrote it (16 VGPRs parked in LDS; no synchronisation involved)
    T2 *const x3 = x2 + 2 * rows2 + lp;

    const int64_t ms = a.ms, ns = a.ns, nxj = ns >> 1, hm = ms >> 1, nxj2 = ns >> 2, hm2 = ms >> 2;
    // row indices are 32-bit (the launcher requires ms < 2^30): per-lane offsets stay in one VGPR, column bases in SGPRs
    const int msi = (int)ms, hmi = (int)hm, hm2i = (int)hm2;
    const int gi = strip * (4 * a.npl) + 4 * lp;                             // first row of this lane (may exceed ms: wraps)
    int row = gi;
    if (row >= msi) row -= msi;
    if (row >= msi) row = 0;                                                 // (lanes far beyond the array: never used)
    const bool loader = lp < a.nload;
    // exact tiling: the last wave is the halo helper (wave-uniform, kept in an SGPR)
    const bool helper = a.helper && (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == (nthreads >> 6) - 1);
    const bool own = (lp < a.npl) && (gi < msi);
    const int ko = gi >> 1, ko2 = gi >> 2;
    int kod = ko + 4;  if (kod >= hmi) kod -= hmi;                           // first d row of this lane
    int kod2 = ko2 + 4; if (kod2 >= hm2i) kod2 -= hm2i;
    const bool odd = (lp & 1) != 0;

    const int64_t j0 = (int64_t)chunk * a.TJ;
    const int64_t jend = (j0 + a.TJ < ns) ? (j0 + a.TJ) : ns;
    const int S_own = (int)((jend - j0) >> 1);                   // steps whose level-1 outputs this chunk owns (multiple of 8)
    const int S = S_own + (NLEV == 2 ? U : 0);                   // + 8 steps that only feed level 2
    const T *base = a.src + (int64_t)blockIdx.y * a.bs_src + row;

    T4 ring[R];
#pragma unroll
    for (int c = 0; c < R; ++c) ring[c] = T4{0.f, 0.f, 0.f, 0.f};
    if (loader) {
#pragma unroll
        for (int c = 0; c < R - 2; ++c) {
            int64_t jc = j0 + c;
            if (jc >= ns) jc -= ns;
            gload16(ring[c], base + jc * a.lds);
        }
    }
    // the first steps find all 14 prologue columns complete (one full wait per wave, once)
#pragma unroll
    for (int c = 0; c < R; c += 2) wait_vm<0>(ring[c], ring[c + 1]);
    // The newest two columns of step t's window (2t+F-2, 2t+F-1) were requested PFD steps earlier.  Younger operations behind
    // them when step t needs them: 2*PFD loads (two per step, this step's included) plus the stores of the PFD steps in
    // between -- 4 per step for a wave that owns rows (NLEV = 1), at least 3 per step while level-1 details are being written
    // (NLEV = 2); none for helper / out-of-range waves.
    const bool wave_stores = __builtin_amdgcn_ballot_w64(own) != 0;
    T *const yb = a.y + (int64_t)blockIdx.y * a.bs_y;
    const bool to_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    T *const llb = to_ll ? (a.ll + (int64_t)blockIdx.y * a.bs_ll) : yb;
    const int64_t ldl = to_ll ? a.ldll : a.ldy;
    const int64_t kbase = j0 >> 1;                     // multiple of 8
    const int64_t kbase2 = j0 >> 2;

    auto step = [&](const int t, const int u, const bool prefetch) __attribute__((always_inline)) {
        if (prefetch && loader) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                int64_t jc = j0 + 2 * t + (R - 2) + e;
                if (jc >= ns) jc -= ns;
                if (jc >= ns) jc -= ns;
                gload16(ring[(2 * u + R - 2 + e) % R], base + jc * a.lds);
            }
        }
        if (prefetch) {
            constexpr int NL = 2 * PFD, NS = NL + PFD * (NLEV == 1 ? 4 : 3);
            static_assert(NS < 64, "vmcnt is a 6-bit counter");
            if (wave_stores && (NLEV == 1 || t <= S_own)) wait_vm<NS>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
            else wait_vm<NL>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);
        } else {
            wait_vm<0>(ring[(2 * u + F - 2) % R], ring[(2 * u + F - 1) % R]);      // (last steps of the chunk: nothing left to overlap)
        }
        // ---- level l, dim-2 pass on row pairs: {A, B}[r] = scaling / detail (column k / kd) of row r ----
        T2 sa01 = a.tp.h[0] * T2{ring[(2 * u) % R].x, ring[(2 * u) % R].y};
        T2 da01 = a.tp.g[F - 1] * T2{ring[(2 * u) % R].x, ring[(2 * u) % R].y};
        T2 sa23 = a.tp.h[0] * T2{ring[(2 * u) % R].z, ring[(2 * u) % R].w};
        T2 da23 = a.tp.g[F - 1] * T2{ring[(2 * u) % R].z, ring[(2 * u) % R].w};
#pragma unroll
        for (int m = 1; m < F; ++m) {
            const T4 xm = ring[(2 * u + m) % R];
            sa01 = sa01 + a.tp.h[m] * T2{xm.x, xm.y};
            da01 = da01 + a.tp.g[F - 1 - m] * T2{xm.x, xm.y};
            sa23 = sa23 + a.tp.h[m] * T2{xm.z, xm.w};
            da23 = da23 + a.tp.g[F - 1 - m] * T2{xm.z, xm.w};
        }
        T2 *const w1 = x1 + (t & 1) * rows1;
        *reinterpret_cast<T4 *>(w1 + 4 * lp) = T4{sa01.x, da01.x, sa01.y, da01.y};
        *reinterpret_cast<T4 *>(w1 + 4 * lp + 2) = T4{sa23.x, da23.x, sa23.y, da23.y};
        // level l+1 runs behind the exchange, spread over two steps: at even steps (lvl2a) its dim-2 pass on the
        // approximation ring -- this step's column (k) is not there yet, so the window is columns k-8 .. k-1 = slots
        // u .. u+7 -- publishes {A2, B2} for the next barrier; at odd steps (lvl2b) the dim-1 pass consumes them
        const bool lvl2a = (NLEV == 2) && !(u & 1) && (t >= U);
        const bool lvl2b = (NLEV == 2) && (u & 1) && (t >= U + 1);
        T2 *const w2 = x2 + ((t >> 1) & 1) * rows2;
        auto level2_dim2 = [&]() __attribute__((always_inline)) {
            T2 r2[F];
#pragma unroll
            for (int m = 0; m < F; ++m) r2[m] = x3[((u + m) % U) * nthreads];
            T2 sa = a.tp.h[0] * r2[0];
            T2 da = a.tp.g[F - 1] * r2[0];
#pragma unroll
            for (int m = 1; m < F; ++m) {
                sa = sa + a.tp.h[m] * r2[m];
                da = da + a.tp.g[F - 1 - m] * r2[m];
            }
            return T4{sa.x, da.x, sa.y, da.y};
        };
        wg_lds_sync(multi);
        __builtin_amdgcn_sched_barrier(0);
        if (helper) {
            // the helper wave owns no output; for level 2 its first lanes keep the approximation rows above the strip alive
            if (NLEV == 2) {
                T4 ab2 = T4{0.f, 0.f, 0.f, 0.f};
                if (lvl2a) ab2 = level2_dim2();            // (reads slot u before it is overwritten below)
                T2 Eh[10];
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const T4 v = *reinterpret_cast<const T4 *>(w1 + 4 * lp + 2 * c);
                    Eh[2 * c] = T2{v.x, v.y};
                    Eh[2 * c + 1] = T2{v.z, v.w};
                }
                T2 s0 = a.tp.h[0] * Eh[0], s1 = a.tp.h[0] * Eh[2];
#pragma unroll
                for (int m = 1; m < F; ++m) { s0 = s0 + a.tp.h[m] * Eh[m]; s1 = s1 + a.tp.h[m] * Eh[2 + m]; }
                x3[u * nthreads] = T2{s0.x, s1.x};
                if (lvl2a) *reinterpret_cast<T4 *>(w2 + 2 * lp) = ab2;
            }
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
            T2 d = a.tp.g[F - 1] * E[2 * q + 10 - F];
#pragma unroll
            for (int m = F - 2; m >= 0; --m) d = d + a.tp.g[m] * E[2 * q + 9 - m];
            P[q] = s;
            Q[q] = d;
        }
        const int64_t k = kbase + t;
        int64_t kd = k + SH;
        if (kd >= nxj) kd -= nxj;
        if (NLEV == 2) {
            if (lvl2a) *reinterpret_cast<T4 *>(w2 + 2 * lp) = level2_dim2();   // (reads slot u before it is overwritten)
            x3[u * nthreads] = T2{P[0].x, P[1].x};     // approximation column kbase + t
            if (t < S_own) {
                // even lane: ds rows kod..kod+3 of column k;  odd lane: sd rows ko-2..ko+1 and dd rows kod-2..kod+1 of column kd
                T rA[2], rB[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    rA[q] = from_partner(odd ? Q[q].x : P[q].y);
                    rB[q] = from_partner(Q[q].y);
                }
                if (own) {
                    T *const ck = yb + k * a.ldy, *const ckd = yb + (nxj + kd) * a.ldy;      // (uniform)
                    if (!odd) {
                        *reinterpret_cast<T4 *>(ck + (hmi + kod)) = T4{Q[0].x, Q[1].x, rA[0], rA[1]};
                    } else {
                        *reinterpret_cast<T4 *>(ckd + (ko - 2)) = T4{rA[0], rA[1], P[0].y, P[1].y};
                        *reinterpret_cast<T4 *>(ckd + (hmi + kod - 2)) = T4{rB[0], rB[1], Q[0].y, Q[1].y};
                    }
                }
            }
        } else {
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
                    *reinterpret_cast<T4 *>(ck + (hmi + kod)) = T4{Q[0].x, Q[1].x, rQ[0], rQ[1]};
                } else {
                    *reinterpret_cast<T4 *>(ckd + (ko - 2)) = T4{rP[0], rP[1], P[0].y, P[1].y};
                    *reinterpret_cast<T4 *>(ckd + (hmi + kod - 2)) = T4{rQ[0], rQ[1], Q[0].y, Q[1].y};
                }
            }
        }
        // ---- level l+1, dim-1 pass: window = approximation rows 2L' .. 2L'+9 ----
        if (lvl2b) {
            __builtin_amdgcn_sched_barrier(0);         // (keeps the level-2 window out of the level-1 live range: register pressure)
            T2 E2[10];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const T4 v = *reinterpret_cast<const T4 *>(w2 + 2 * lp + 2 * c);
                E2[2 * c] = T2{v.x, v.y};
                E2[2 * c + 1] = T2{v.z, v.w};
            }
            T2 P2 = a.tp.h[0] * E2[0];                 // {ss2, sd2} of row ko2
#pragma unroll
            for (int m = 1; m < F; ++m) P2 = P2 + a.tp.h[m] * E2[m];
            T2 Q2 = a.tp.g[F - 1] * E2[10 - F];        // {ds2, dd2} of row kod2
#pragma unroll
            for (int m = F - 2; m >= 0; --m) Q2 = Q2 + a.tp.g[m] * E2[9 - m];
            {
                const int64_t k2 = kbase2 + ((t - U - 1) >> 1);
                int64_t kd2 = k2 + SH;
                if (kd2 >= nxj2) kd2 -= nxj2;
                const T rP = from_partner(odd ? P2.x : P2.y);
                const T rQ = from_partner(odd ? Q2.x : Q2.y);
                if (own) {
                    // even lane: ss2 rows ko2, ko2+1 / ds2 rows kod2, kod2+1 of column k2;  odd lane: sd2 / dd2 of column kd2
                    T *const ck = yb + k2 * a.ldy, *const ckd = yb + (nxj2 + kd2) * a.ldy, *const cl = llb + k2 * ldl;  // (uniform)
                    if (!odd) {
                        *reinterpret_cast<T2 *>(cl + ko2) = T2{P2.x, rP};
                        *reinterpret_cast<T2 *>(ck + (hm2i + kod2)) = T2{Q2.x, rQ};
                    } else {
                        *reinterpret_cast<T2 *>(ckd + (ko2 - 1)) = T2{rP, P2.y};
                        *reinterpret_cast<T2 *>(ckd + (hm2i + kod2 - 1)) = T2{rQ, Q2.y};
                    }
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
static hipError_t set_lds_attr(const void *fn, size_t bytes)
{
    static thread_local const void *done_fn[32];
    static thread_local int done_dev[32];
    static thread_local int ndone = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    for (int i = 0; i < ndone; ++i)
        if (done_fn[i] == fn && done_dev[i] == dev) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && ndone < 32) { done_fn[ndone] = fn; done_dev[ndone] = dev; ++ndone; }
    return e;
}

// Workgroup shape for a block of `ms` rows: returns waves per workgroup, owned lanes and loader lanes.
struct Shape2D { int nw, npl, nload, nstrips, helper; };
static Shape2D pick_shape(int64_t ms, int nlev, int mode, int wmain)
{
    const int HL = (nlev == 2) ? 6 : 2;
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
static hipError_t launch_lds_f(hipStream_t st, const Taps<float> &taps, int nlev, bool lvl1, const float *src, int64_t lds,
                               float *y, int64_t ldy, float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                               int64_t nbatch, int64_t bs_src, int64_t bs_y, int64_t bs_ll, int nll)
{
    Lds2DArgs<F> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldll = ldll; a.ms = ms; a.ns = ns;
    a.bs_src = bs_src; a.bs_y = bs_y; a.bs_ll = bs_ll; a.nll = nll;
    const Shape2D sh = pick_shape(ms, nlev, (int)opt("WL_LDS_MODE", 0), (int)opt("WL_LDS_W", 4));
    a.npl = sh.npl; a.nload = sh.nload; a.nstrips = sh.nstrips; a.helper = sh.helper;
    int TJ = (int)opt(nlev == 2 ? "WL_TJ2" : "WL_TJ", 128);
    auto nwaves = [&](int tj) { return (int64_t)a.nstrips * sh.nw * ((ns + tj - 1) / tj) * nbatch; };
    const int wpc = (int)opt("WL_WAVES_PER_CU", 8);
    while (TJ > 32 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * wpc) TJ >>= 1;
    if (nlev == 1)
        while (TJ > 16 && (TJ % 32) == 0 && nwaves(TJ) < (int64_t)cu_count * opt("WL_WAVES_MIN", 8)) TJ >>= 1;
    a.TJ = TJ;
    a.nchunks = (int)((ns + TJ - 1) / TJ);
    a.rev = (!lvl1 && opt("WL_REVERSE", 1)) ? 1 : 0;
    a.tp = shrink<float, F>(taps);
    const unsigned nwg = (unsigned)(a.nstrips * a.nchunks);
    const int nthreads = 64 * sh.nw;
    const size_t shmem = (size_t)2 * (4 * nthreads + 16) * 8 + (nlev == 2 ? (size_t)2 * (2 * nthreads + 16) * 8 + (size_t)8 * nthreads * 8 : 0);
#define WL_LDS_LAUNCH(NL_, L1_)                                                                                        \
    do {                                                                                                               \
        hipError_t e = set_lds_attr(reinterpret_cast<const void *>(&k_fwd2d_lds<F, NL_, L1_>), 65536);                 \
        if (e != hipSuccess) return e;                                                                                 \
        hipLaunchKernelGGL((k_fwd2d_lds<F, NL_, L1_>), dim3(nwg, (unsigned)nbatch), dim3(nthreads), shmem, st, a);     \
    } while (0)
    if constexpr (F <= 8) {
        if (nlev == 2) {
            if (lvl1) WL_LDS_LAUNCH(2, 1); else WL_LDS_LAUNCH(2, 0);
            return hipGetLastError();
        }
    }
    if (lvl1) WL_LDS_LAUNCH(1, 1); else WL_LDS_LAUNCH(1, 0);
#undef WL_LDS_LAUNCH
    return hipGetLastError();
}

bool fwd2d_lds_ok(int F, int nlev, int64_t ms, int64_t ns)
{
    if (F < 2 || F > 10 || (F & 1)) return false;
    if (nlev == 2 && F > 8) return false;
    // rows: lanes own 4 rows, d rows wrap in groups of 4 (level 1) / 2 (level 2); columns: chunks of 16
    if (ms >= ((int64_t)1 << 30)) return false;
    if (nlev == 1) return ms >= 64 && (ms % 8) == 0 && ns >= 16 && (ns % 16) == 0;
    return ms >= 128 && (ms % 16) == 0 && ns >= 64 && (ns % 32) == 0;
}

hipError_t fwd2d_lds_launch(hipStream_t st, const Taps<float> &taps, int nlev, bool lvl1, const float *src, int64_t lds,
                            float *y, int64_t ldy, float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                            int64_t nbatch, int64_t bs_src, int64_t bs_y, int64_t bs_ll, int nll)
{
    switch (taps.F) {
    case 2: return launch_lds_f<2>(st, taps, nlev, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 4: return launch_lds_f<4>(st, taps, nlev, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 6: return launch_lds_f<6>(st, taps, nlev, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 8: return launch_lds_f<8>(st, taps, nlev, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
    case 10:
        if (nlev == 1) return launch_lds_f<10>(st, taps, nlev, lvl1, src, lds, y, ldy, ll, ldll, ms, ns, cu_count, nbatch, bs_src, bs_y, bs_ll, nll);
        return hipErrorInvalidValue;
    default: return hipErrorInvalidValue;
    }
}

}  // namespace wl
