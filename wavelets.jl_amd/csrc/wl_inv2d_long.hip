// wl_inv2d_long.hip -- ONE fused 2-D INVERSE level per pass over HBM through an LDS exchange: the mirror of k_fwd2d_lds_long
// (wl_fwd2d_long.hip).  Written in round 4 for the 12 ... 20-tap filters (db6 ... db10, sym6 ... sym10, coif4, coif6, beyl), which
// up to round 3 took two line launches + one axis launch per inverse level through an N-element intermediate (2 x the traffic,
// 3 launches).  Also used where it measured ahead of the DPP-halo streaming kernel (k_inv2d_stream, wl_inv.hip):
//   Float32  12 ... 20 taps: levels of >= 256 output rows;  10 taps (sym5, the default wavelet of denoise): blocks / batches of >= 2^22
//            samples (8192^2 level 130 -> 117 us, 64 x 2048^2 batch 532 -> 436 us);  8 taps: not used (112 against 114 us, and the
//            fused pair of k_inv2d_pair is ahead of both)
//   Float64  8 ... 20 taps (8192^2 idwt L = 13: db4 331 -> 316 us, sym5 390 -> 355, db6 711 -> 376, db8 733 -> 437, db10 781 -> 631;
//            14 taps and more run one wave per SIMD with the ring's overflow in AGPRs)
// Batches (planes of a 3-D level, images of wl_dwt_filter_batch, the spins of the TI denoise) ride on blockIdx.y.
//
//   k_inv2d_lds_long<T, F, W, D>
//
// Reference order (transforms_filter.jl:173-186): dim-1 reconstruction of every column, then dim-2 reconstruction of every row.
// A workgroup of W waves (default 1) owns 256 W OUTPUT rows (= 128 W coefficient pairs along dim 1: two pairs per lane, exact
// tiling, no overlap) and marches along the output column pairs p of a chunk, as k_inv2d_stream does:
//   * per step it takes the raw coefficient column p of the left half and column p + SH of the right half (the lane's two
//     scaling and two detail coefficients of each: four 8-byte loads, requested D steps ahead),
//   * publishes them in LDS -- the reach of a 20-tap filter is 9 pairs on either side, beyond any DPP exchange -- together with
//     the SH scaling pairs before the strip and the SH detail pairs after it (periodic wrap), which 2 SH lanes of wave 0 load,
//   * after the step's only barrier reads its windows back (scaling pairs r-SH .. r+1, detail pairs r .. r+1+SH) and
//     reconstructs the two columns along dim 1 (window_inv, wl_dev.h) into two register rings of R >= SH + 1 slots,
//   * combines ring columns p-SH .. p / p .. p+SH into output columns 2p, 2p+1 (window_inv again) and stores 16 bytes per column.
// The LDS buffers alternate with the step parity, so one barrier per step is enough.  The loop body is unrolled over the ring
// (R steps, R a multiple of D so that request slots are compile-time); a step past the chunk is an EXIT (see k_fwd2d_lds_long).
// Loads and waits are the compiler's here.
// What bounds it (measured r04, profiles/r04_inv_long.md): a step is ~500 f32 multiplies / adds per lane (2 F per sample and pass,
// no FMA by the arithmetic contract) + ~100 other VALU instructions = 1370 cycles of VALU pipe time at the 2.3-cycle rate two or
// more waves reach together (one wave alone issues every 4.6 cycles); measured 2500 cycles per wave-step and SIMD with the two
// waves that 150 ... 230 VGPRs allow: the exchange round trip of one wave is covered by one other wave only.  8192^2 db8 = 150 us
// = 3.6 TB/s, well above the two-pass tier (226 us, twice the traffic), below the copy ceiling.  Packed arithmetic is neutral (a
// packed instruction costs two scalar ones on a shared pipe), a rolled loop (1/9 of the code) is slower, longer request
// distances are neutral; four waves per SIMD (10 taps, shorter chunks) gain 4-10 %.
// A first draft used a helper wave for the halo (as k_fwd2d_lds_long): it held a wave slot with the main waves' 130 ... 210 VGPRs,
// i.e. half of the two-waves-per-SIMD occupancy, for two loads per step.
// Arithmetic: the closed form of filtup! in the reference's summation order (wl_internal.h) -- bit-identical to the two-pass tier.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <typename T, int F>
struct InvLongArgs {
    const T *x; int64_t ldx;            // coefficient array
    const T *ll; int64_t ldl;           // deeper reconstruction = approximation quadrant, h0 x h1 (nullptr: it is in x)
    T *dst; int64_t ldd;                // n0 x n1 result
    int64_t n0, n1;
    int TP;                             // output column pairs per chunk
    int nstrips, nchunks;
    // batch of independent blocks over blockIdx.y (planes of a 3-D level, images of a batch): element strides; only the first nll
    // planes take their approximation quadrant from ll
    int64_t bs_x, bs_ll, bs_dst; int nll;
    TapsF<T, F> tp;
};

// window_inv (wl_dev.h) with the detail taps taken from the scaling taps: g[m] = (-1)^m h[m] EXACTLY (make_taps, wl_internal.h), so
// g[2q] = h[2q] and g[2q + 1] = -h[2q + 1]: the negation is a source modifier of the multiply, and only F instead of 2 F tap values
// occupy SGPRs (the 16-tap instance reloaded 78 spilled SGPRs per step through v_readlane_b32: 13 % of its VALU instructions)
template <typename T, int F>
__device__ __forceinline__ void window_inv_h(const T (&sw)[(F - 2) / 2 + 1], const T (&dw)[(F - 2) / 2 + 1], const TapsF<T, F> &tp, T &xe, T &xo)
{
    constexpr int SH = (F - 2) / 2;
    T Se = tp.h[F - 2] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Se = Se + tp.h[F - 2 - 2 * q] * sw[q];
    T De = (-tp.h[1]) * dw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) De = De + (-tp.h[1 + 2 * q]) * dw[q];
    xe = Se + De;
    T So = tp.h[F - 1] * sw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) So = So + tp.h[F - 1 - 2 * q] * sw[q];
    T Do = tp.h[0] * dw[0];
#pragma unroll
    for (int q = 1; q <= SH; ++q) Do = Do + tp.h[2 * q] * dw[q];
    xo = So + Do;
}

#ifndef WL_INVLONG_HTAPS_MIN
#define WL_INVLONG_HTAPS_MIN 12          // (10 taps and fewer: nothing spills, and the separate tap table measured 1-3 % ahead)
#endif
template <typename T, int F>
__device__ __forceinline__ void window_inv_sel(const T (&sw)[(F - 2) / 2 + 1], const T (&dw)[(F - 2) / 2 + 1], const TapsF<T, F> &tp, T &xe, T &xo)
{
    if constexpr (F >= WL_INVLONG_HTAPS_MIN) window_inv_h<T, F>(sw, dw, tp, xe, xo);
    else window_inv<T, F>(sw, dw, tp, xe, xo);
}

// four consecutive samples: 16 bytes per store (one for Float32, two for Float64)
template <typename T>
__device__ __forceinline__ void store4(T *p, const T (&v)[4])
{
    constexpr int C = (int)(16 / sizeof(T));
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < 4 / C; ++c) {
        V t;
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        store_pol<((WL_P_ILONG_ST != 0 && sizeof(T) == 4) ? 1 : 0)>(reinterpret_cast<V *>(p + c * C), t);
    }
}

// PPL: coefficient pairs per lane (2: strips of 256 W rows, 8-byte loads and 16-byte stores; 1: strips of 128 W rows with half
// the ring registers -- four waves per SIMD for the 12 ... 20-tap instances -- at 4-byte loads and 8-byte stores)
template <typename T, int F, int W, int D, int ROLL = 0, int PPL = 2>
__global__ void __launch_bounds__(64 * W, (sizeof(T) == 8 && F >= 14) ? 1 : 2) k_inv2d_lds_long(InvLongArgs<T, F> a)
{
    typedef T T2 __attribute__((ext_vector_type(2)));
    static_assert(PPL == 1 || PPL == 2, "one or two pairs per lane");
    constexpr int SH = (F - 2) / 2, SHP = (SH + 1) & ~1;      // reach in pairs; the same rounded up to even (8-byte aligned LDS rows)
    constexpr int R = ((SH + 1 + D - 1) / D) * D;             // ring depth = unroll: >= SH + 1, a multiple of the request distance D
    constexpr int NP = 64 * PPL * W;                          // coefficient pairs of the strip
    constexpr int LA = NP + SHP + 4;                          // one LDS array: positions [0, NP + SHP) (+ pad)
    __shared__ __attribute__((aligned(16))) T lds[2][4][LA];  // [step parity][Ls, Ld, Rs, Rd]

    const int wv = (W > 1) ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    const uint32_t q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    const uint32_t logical = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    const int strip = (int)(logical % (uint32_t)a.nstrips);
    const int chunk = (int)(logical / (uint32_t)a.nstrips);
    const int64_t h0 = a.n0 >> 1, h1 = a.n1 >> 1;
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < h1) ? (p0 + a.TP) : h1;
    const int S = (int)(pend - p0);
    const int64_t r0 = (int64_t)strip * NP;
    const bool from_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    const int64_t ls_ld = from_ll ? a.ldl : a.ldx;
    const T *const xb = a.x + (int64_t)blockIdx.y * a.bs_x;
    const T *const llb = from_ll ? (a.ll + (int64_t)blockIdx.y * a.bs_ll) : xb;
    // the raw columns of step t (t from -SH: wraps below zero; the requests past the chunk wrap above and are never used)
    auto col_s = [&](const int t) __attribute__((always_inline)) { int64_t j = p0 + t; if (j < 0) j += h1; if (j >= h1) j -= h1; return j; };
    auto col_d = [&](const int t) __attribute__((always_inline)) { int64_t j = p0 + t + SH; if (j >= h1) j -= h1; return j; };

    // ================= lane Lm = 64 wv + lane owns pairs r0 + 2 Lm, r0 + 2 Lm + 1 =================
    const int Lm = 64 * wv + lane;
    const int64_t rr = r0 + PPL * Lm;
    const T *ls_base = llb + rr;                              // left half, scaling rows
    const T *ld_base = xb + h0 + rr;                          // left half, detail rows
    const T *rs_base = xb + h1 * a.ldx + rr;                  // right half, scaling rows
    const T *rd_base = rs_base + h0;                          // right half, detail rows
    const int ws = SHP + PPL * Lm;                            // LDS positions: scaling arrays hold pair r0 - SHP + i at i, detail arrays r0 + i
    const int wd = PPL * Lm;
    // the halo of the exchange (periodic wrap) rides on wave 0: lane h < SH takes scaling pair r0 - SH + h of both halves,
    // lane SH <= h < 2 SH detail pair r0 + NP + (h - SH).  (Up to round-4 draft 1 a helper wave did this: it cost a wave slot with
    // the main waves' register allocation, i.e. half of the two-waves-per-SIMD occupancy.)
    const bool halo_wave = (wv == 0);
    const bool hd = lane >= SH;
    const bool hl = lane < 2 * SH;
    int64_t hr = 0;
    if (hl) { hr = hd ? (r0 + NP + (lane - SH)) : (r0 - SH + lane); if (hr < 0) hr += h0; if (hr >= h0) hr -= h0; }
    const T *hlb = hd ? (xb + h0 + hr) : (llb + hr);
    const int64_t hl_ld = hd ? a.ldx : ls_ld;
    const T *hrb = xb + h1 * a.ldx + (hd ? h0 : 0) + hr;
    const int hw = hd ? (NP + lane - SH) : (SHP - SH + lane);
    const int hal = hd ? 1 : 0;

    T raw[D][4][PPL];                                         // requests in flight: D steps ahead
    auto ldp = [&](const T *q, T (&v)[PPL]) __attribute__((always_inline)) {
        if constexpr (PPL == 2) { const T2 t2 = load_pol<(WL_P_ILONG_LD != 0 && sizeof(T) == 4)>(reinterpret_cast<const T2 *>(q)); v[0] = t2.x; v[1] = t2.y; }
        else v[0] = load_pol<(WL_P_ILONG_LD != 0 && sizeof(T) == 4)>(q);
    };
    auto stp = [&](T *q, const T (&v)[PPL]) __attribute__((always_inline)) {
        if constexpr (PPL == 2) *reinterpret_cast<T2 *>(q) = T2{v[0], v[1]};
        else *q = v[0];
    };
    T hv[D][2];
    auto load_raw = [&](const int t, const int s) __attribute__((always_inline)) {
        const int64_t js = col_s(t), jd = col_d(t);
        ldp(ls_base + js * ls_ld, raw[s][0]);
        ldp(ld_base + js * a.ldx, raw[s][1]);
        ldp(rs_base + jd * a.ldx, raw[s][2]);
        ldp(rd_base + jd * a.ldx, raw[s][3]);
        if (halo_wave) {
            if (hl) { hv[s][0] = hlb[js * hl_ld]; hv[s][1] = hrb[jd * a.ldx]; }
        }
    };

    T iS[R][2 * PPL], iD[R][2 * PPL];                         // dim-1-reconstructed columns: left half / right half
    T *out = a.dst + (int64_t)blockIdx.y * a.bs_dst + 2 * (r0 + PPL * Lm);    // the lane's 2 PPL output rows
    // one column of the exchange -> the lane's 2 PPL dim-1-reconstructed samples
    auto recon = [&](const T *sa, const T *da, T (&o)[2 * PPL]) __attribute__((always_inline)) {
        // scaling pairs r - SH .. r + PPL - 1 at positions ws - SH .. (PPL = 2: ws - SHP is even, 8-byte reads from there), detail pairs r .. r + PPL - 1 + SH
        T sv[SHP + 2], dv[SH + 3];
        if constexpr (PPL == 2) {
#pragma unroll
            for (int i = 0; i < (SHP + 2) / 2; ++i) {
                const T2 v = *reinterpret_cast<const T2 *>(sa + ws - SHP + 2 * i);
                sv[2 * i] = v.x; sv[2 * i + 1] = v.y;
            }
#pragma unroll
            for (int i = 0; i < (SH + 3) / 2; ++i) {
                const T2 v = *reinterpret_cast<const T2 *>(da + wd + 2 * i);
                dv[2 * i] = v.x; dv[2 * i + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int i = 0; i <= SH; ++i) { sv[SHP - SH + i] = sa[ws - SH + i]; dv[i] = da[wd + i]; }
        }
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int i = 0; i <= SH; ++i) { sw[i] = sv[SHP - SH + p + i]; dw[i] = dv[p + i]; }
            window_inv_sel<T, F>(sw, dw, a.tp, o[2 * p], o[2 * p + 1]);
        }
    };
    // t: step (from -SH: the first SH steps only fill the rings), s: request slot, u: ring slot of this step's columns
    auto step = [&](const int t, const int s, const int u, const bool combine) __attribute__((always_inline)) {
        T(*buf)[LA] = lds[t & 1];
        stp(&buf[0][ws], raw[s][0]);
        stp(&buf[1][wd], raw[s][1]);
        stp(&buf[2][ws], raw[s][2]);
        stp(&buf[3][wd], raw[s][3]);
        if (halo_wave) {
            if (hl) { buf[hal][hw] = hv[s][0]; buf[2 + hal][hw] = hv[s][1]; }
        }
        load_raw(t + D, s);
        lds_barrier();
        // (never taken: a block boundary between the request / publish half and the reconstruction half of a step -- without it the
        //  compiler schedules the unrolled steps as one region, 256 VGPRs + scratch for F >= 16; with it 125 ... 209, no scratch)
        if (__builtin_expect(a.nchunks < 0, 0)) return;
        recon(buf[0], buf[1], iS[u]);
        recon(buf[2], buf[3], iD[u]);
        if (!combine) return;
        T xe[2 * PPL], xo[2 * PPL];
#pragma unroll
        for (int q = 0; q < 2 * PPL; ++q) {
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int i = 0; i <= SH; ++i) { sw[i] = iS[(u + i + R - SH) % R][q]; dw[i] = iD[(u + i + R - SH) % R][q]; }
            window_inv_sel<T, F>(sw, dw, a.tp, xe[q], xo[q]);
        }
        const int64_t p = p0 + t;
        if constexpr (PPL == 2) {
            store4<T>(out + (2 * p) * a.ldd, xe);
            store4<T>(out + (2 * p + 1) * a.ldd, xo);
        } else {
            *reinterpret_cast<T2 *>(out + (2 * p) * a.ldd) = T2{xe[0], xe[1]};
            *reinterpret_cast<T2 *>(out + (2 * p + 1) * a.ldd) = T2{xo[0], xo[1]};
        }
    };

    if constexpr (ROLL != 0) {
        // rolled form (D = 1, R = SH + 1): ONE step's code in a real loop, the rings shifted by register moves instead of being
        // addressed by an unrolled slot index -- 1/R of the instruction footprint for 2 R x 4 moves per step
        static_assert(D == 1, "the rolled form requests one step ahead");
        load_raw(-SH, 0);
#pragma unroll 1
        for (int t = -SH; t < S; ++t) {
#pragma unroll
            for (int i = 0; i + 1 < R; ++i) {
#pragma unroll
                for (int q = 0; q < 2 * PPL; ++q) { iS[i][q] = iS[i + 1][q]; iD[i][q] = iD[i + 1][q]; }
            }
            step(t, 0, R - 1, t >= 0);
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < D; ++s) load_raw(s - SH, s);
    // prologue: steps -SH .. -1 -> ring slots R - SH .. R - 1, request slots c % D (step t uses request slot (t + SH) % D; R % D == 0)
#pragma unroll
    for (int c = 0; c < SH; ++c) step(c - SH, c % D, c - SH + R, false);
    for (int t0 = 0; t0 < S; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
            if (t0 + u >= S) return;                          // (workgroup-uniform: every wave takes the same barriers)
            step(t0 + u, (u + SH) % D, u, true);
        }
    }
}

// ------------------------------------------------------------------------------------------
bool inv2d_long_ok(int F, int64_t n0, int64_t n1, int esize)
{
    if (F < 8 || F > 20 || (F & 1)) return false;
    if (esize == 4 && F < (int)opt("WL_INVLONG_FMIN", 10)) return false;
    if (esize == 8 && (F < (int)opt("WL_INVLONG_FMIN64", 8) || F > (int)opt("WL_INVLONG_FMAX64", 20))) return false;
    if (n0 >= ((int64_t)1 << 30) || n1 >= ((int64_t)1 << 30)) return false;
    // exact tiling: strips of 256 output rows per main wave; the reach (SH pairs) must not wrap twice
    const int64_t h0 = n0 >> 1, h1 = n1 >> 1;
    return n0 >= 256 && (n0 % 256) == 0 && (n1 % 2) == 0 && h1 >= (F - 2) / 2 + 5 && h0 >= (F - 2) / 2;   // (+ 4: requests up to 4 steps ahead wrap once)
}

template <typename T, int F, int W, int D, int ROLL = 0, int PPL = 2>
static hipError_t launch_inv_long_fwd(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl,
                                      T *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count, const InvLongBatch &bt)
{
    InvLongArgs<T, F> a;
    a.x = x; a.ldx = ldx; a.ll = ll; a.ldl = ldl; a.dst = dst; a.ldd = ldd; a.n0 = n0; a.n1 = n1;
    a.bs_x = bt.bs_x; a.bs_ll = bt.bs_ll; a.bs_dst = bt.bs_dst; a.nll = bt.nll;
    const int64_t h1 = n1 >> 1;
    a.nstrips = (int)(n0 / (128 * PPL * W));
    // a chunk pays SH prologue steps: long chunks where the array is large enough to fill the chip (8 waves per CU: two per SIMD
    // at 130 ... 210 VGPRs) with them, shorter ones below (measured r04: 8192^2 TP 64, 4096^2 TP 16, 2048^2 TP 8)
    int TP = (int)opt("WL_INVLONG_TP", 64);
    if (TP < 8) TP = 64;
    // (10 taps, Float32, 125 VGPRs = four waves per SIMD possible: 8192^2 level 119.5 us with 64-pair chunks, 115.3 with 32, 137-151
    //  with 128 (one wave per SIMD) -- but a 16-waves-per-CU target halves the chunks of the smaller levels too and the whole
    //  transform loses: 187 against 182 us.  Kept at 8.)
    const int64_t want = (int64_t)cu_count * opt("WL_INVLONG_WAVES_PER_CU", 8);
    while (TP > 8 && (int64_t)a.nstrips * ((h1 + TP - 1) / TP) * W * bt.nplanes < want) TP >>= 1;
    a.TP = TP;
    a.nchunks = (int)((h1 + TP - 1) / TP);
    a.tp = shrink<T, F>(taps);
    hipLaunchKernelGGL((k_inv2d_lds_long<T, F, W, D, ROLL, PPL>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)bt.nplanes), dim3(64 * W), 0, st, a);
    return hipGetLastError();
}

template <typename T, int F, int W>
static hipError_t launch_inv_long_fw(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl,
                                     T *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count, const InvLongBatch &bt)
{
    // request distance in steps: a step is ~0.3 us of arithmetic, a loaded HBM round trip 1 - 2 us
    int D = (int)opt("WL_INVLONG_D", F <= 10 ? 2 : 3);       // (measured r04: db8 164 us with 3 against 167 with 2; sym5 batches 436 with 2 against 463 with 3)
    if (sizeof(T) == 8 && D > 3) D = 3;
    if constexpr (sizeof(T) == 4 && W == 1) {                   // (experiment knob: the rolled form, Float32, one wave per workgroup)
        if (opt("WL_INVLONG_ROLL", 0) != 0) return launch_inv_long_fwd<T, F, W, 1, 1>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
        // one pair per lane (strips of 128 rows, half the ring registers): twice the strips and four waves per SIMD.  Measured r04
        // (db6 / db8 / db10, one level, us): 8192^2 129 / 150 / 166 -> 131 / 153 / 178 (more LDS instructions per output, 8-byte
        // stores), 4096^2 38 / 48 / 57 -> 40 / 46 / 55, 2048^2 16 / 21 / 26 -> 14 / 18 / 22, 1024^2 14 / 18 / 23 -> 10 / 13 / 16,
        // 512^2 13 / 17 / 22 -> 9 / 12 / 15: the small levels are short of workgroups, not of bandwidth
        int ppl = (int)opt("WL_INVLONG_PPL", 0);
        if (ppl != 1 && ppl != 2) ppl = (F >= 12 && (n0 <= 2048 || (F >= 16 && n0 <= 4096))) ? 1 : 2;
        if (ppl == 1) return launch_inv_long_fwd<T, F, W, 3, 0, 1>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    }                       // (Float64 with 4 columns in flight spills at 10 / 12 taps)
    if (D <= 1) return launch_inv_long_fwd<T, F, W, 1>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    if (D == 2) return launch_inv_long_fwd<T, F, W, 2>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    if (D == 3) return launch_inv_long_fwd<T, F, W, 3>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    if constexpr (sizeof(T) == 4) return launch_inv_long_fwd<T, F, W, 4>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    else return launch_inv_long_fwd<T, F, W, 3>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
}

template <typename T, int F>
static hipError_t launch_inv_long_f(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl,
                                    T *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count, const InvLongBatch &bt)
{
    int W = (int)opt("WL_INVLONG_W", 0);
    if (W != 1 && W != 2 && W != 4) W = 1;                    // (one wave per workgroup: the step's barrier is free; W = 2, 4 measured equal or slower)
    while (W > 1 && (n0 % (256 * W)) != 0) W >>= 1;
    if (W == 4) return launch_inv_long_fw<T, F, 4>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    if (W == 2) return launch_inv_long_fw<T, F, 2>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    return launch_inv_long_fw<T, F, 1>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
}

template <typename T>
hipError_t inv2d_long_launch(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl,
                             T *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count, const InvLongBatch &bt)
{
    switch (taps.F) {
    case 8: return launch_inv_long_f<T, 8>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    case 10: return launch_inv_long_f<T, 10>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    case 12: return launch_inv_long_f<T, 12>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    case 14: return launch_inv_long_f<T, 14>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    case 16: return launch_inv_long_f<T, 16>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    case 18: return launch_inv_long_f<T, 18>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    case 20: return launch_inv_long_f<T, 20>(st, taps, x, ldx, ll, ldl, dst, ldd, n0, n1, cu_count, bt);
    default: return hipErrorInvalidValue;
    }
}

template hipError_t inv2d_long_launch<float>(hipStream_t, const Taps<float> &, const float *, int64_t, const float *, int64_t, float *, int64_t,
                                             int64_t, int64_t, int, const InvLongBatch &);
template hipError_t inv2d_long_launch<double>(hipStream_t, const Taps<double> &, const double *, int64_t, const double *, int64_t, double *, int64_t,
                                              int64_t, int64_t, int, const InvLongBatch &);

}  // namespace wl
