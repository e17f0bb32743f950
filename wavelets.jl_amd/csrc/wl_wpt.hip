// wl_wpt.hip -- fully split depths of the 1-D wavelet PACKET transform (transforms_filter.jl:301-359: at depth d the vector is
// 2^d segments of length n / 2^d and every segment gets one [s ; d] level), Float32 / Float64, even F <= 10.
//
// The reference -- and this library up to round 3 -- runs one pass over the whole vector per depth.  In a packet tree BOTH
// halves of a segment recurse, so nothing leaves the chip between depths except the leaves:
//
//   k_wpt_fwd_multi   NL <= 3 consecutive depths per pass over HBM.  A workgroup owns a tile of TS samples of one segment,
//                     stages it with a halo of H0 = (F-2)(2^NL - 1) samples on both sides (periodic wrap of the SEGMENT
//                     resolved while staging) and runs the depths LDS -> LDS: at fused level t the tile is 2^t bands, each
//                     covering the owned range widened by H_t = (F-2)(2^(NL-t) - 1), so pair i of a band reads the fixed
//                     window [2i, 2i + 2F - 2) of its parent band (the indexing of k_fwd1d_multi, wl_fwd.hip).  Only the
//                     2^NL leaf bands of the last level go to memory, TS / 2^NL contiguous samples each.
//   k_wpt_fwd_tail    every remaining depth of the segments that fit a workgroup (<= TS samples, powers of two): a chunk of
//                     TS samples = whole segments is staged once, each depth is one LDS -> LDS pass with exact periodic
//                     indexing (bit mask), down to segments of two samples; one launch instead of one per depth.
//   k_wpt_inv_tail    the mirror for iwpt: the deepest depths first, [s ; d] halves of a segment -> the segment.
//
// Arithmetic: the closed forms of wl_internal.h in the reference's summation order, no FMA -- bit-identical to the per-depth
// kernels (tests/test_gpu_parity.py::test_wpt_bitexact pins both against the oracle).
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

template <typename T, int F>
struct WptMultiArgs {
    const T *src; T *dst;
    int64_t nj;                     // segment length at the first fused depth (a multiple of TS)
    int NL, TS;
    int bs[4];                      // band stride in LDS at fused level t (elements, multiple of 16 bytes)
    int buf_elems;                  // elements per LDS buffer
    // partially split trees (round 5): node bits of the whole tree on the device (one byte per node, node (2^D - 1) + p = segment p
    // of depth D; nullptr: every segment of every fused depth splits) and the depth of the first fused level.  A valid tree has no
    // set node below an unset one, so "bit clear" means "this region is (part of) a leaf: pass it through".
    const uint8_t *mask; int depth;
    TapsF<T, F> tp;
};

template <typename T, int F>
__global__ void __launch_bounds__(256) k_wpt_fwd_multi(WptMultiArgs<T, F> a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int VEC = 16 / sizeof(T);
    constexpr int PPT = VEC;                                              // pairs per thread: 16 bytes of each output band
    constexpr int NWIN = ((2 * PPT + 2 * F - 4) + VEC - 1) / VEC * VEC;
    const int tid = threadIdx.x;
    const int NL = a.NL, TS = a.TS;
    int H[4];
    H[NL] = 0;
    for (int t = NL; t >= 1; --t) H[t - 1] = 2 * H[t] + (F - 2);
    const int64_t own0 = (int64_t)blockIdx.x * TS;                        // position in the vector
    const int64_t root = own0 / a.nj;                                     // segment of the first fused depth
    const int64_t r0 = own0 - root * a.nj;                                // ... and the tile's offset inside it
    const T *seg = a.src + root * a.nj;
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + a.buf_elems;
    auto gq = [&](const int m) __attribute__((always_inline)) { return (m & 1) ? -a.tp.h[m] : a.tp.h[m]; };

    // ---- stage: A[j] = seg[(r0 - H0 + j) mod nj], j = 0 .. TS + 2 H0 - 1; 16-byte chunks at aligned positions ----
    {
        const int lenA = TS + 2 * H[0];
        const int64_t start = r0 - H[0];
        const int r = (int)(((start % VEC) + VEC) % VEC);                 // 0, or 2 for Float32 (H0 is even)
        const int64_t astart = start - r;
        const int nch = (lenA + r + VEC - 1) / VEC;
        constexpr int UL = 6;
        for (int c0 = tid; c0 < nch; c0 += UL * 256) {
            T v[UL][VEC];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int c = c0 + u * 256;
                if (c < nch) {
                    int64_t g = astart + (int64_t)c * VEC;
                    if (g < 0) g += a.nj;
                    if (g >= a.nj) g -= a.nj;
                    vload<T, VEC>(seg + g, v[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int c = c0 + u * 256;
                if (c < nch) {
                    const int j0 = c * VEC - r;
                    if (r == 0 && j0 + VEC <= lenA) {
                        vstore16<T, VEC>(bufA + j0, v[u]);
                    } else {
#pragma unroll
                        for (int e = 0; e < VEC; e += 2) {                // (r and lenA are even: pairs never straddle the ends)
                            const int j = j0 + e;
                            if (j >= 0 && j + 2 <= lenA) vstore<T, 2>(bufA + j, reinterpret_cast<const T(&)[2]>(v[u][e]));
                        }
                    }
                }
            }
        }
    }
    lds_barrier_vm();

    T *Ain = bufA, *Aout = bufB;
    for (int t = 1; t <= NL; ++t) {
        const int ownt = TS >> t;
        const int Lout = ownt + 2 * H[t];
        const int gpb = (Lout + PPT - 1) / PPT;                           // thread groups per parent band
        const int nbands = 1 << (t - 1);
        const int total = gpb * nbands;
        const int bsi = a.bs[t - 1], bso = a.bs[t];
        const bool lastlev = (t == NL);
        // leaves: band b of the root segment lives at root * nj + b * (nj >> NL), this tile's piece at (r0 >> NL)
        const int64_t leaf = a.nj >> NL;
        T *out0 = a.dst + root * a.nj + (r0 >> NL);
        const int64_t segb = root << (t - 1);                             // first segment of depth a.depth + t - 1 under this root
        for (int g = tid; g < total; g += 256) {
            const int b = g / gpb;
            const int i = (g - b * gpb) * PPT;
            if (a.mask) {
                // live: every ancestor of parent band b inside this launch splits; bit: the band itself splits
                bool live = true;
                for (int u = 1; u < t; ++u)
                    live = live && a.mask[(((int64_t)1 << (a.depth + u - 1)) - 1) + (root << (u - 1)) + (b >> (t - u))] != 0;
                if (!live) continue;
                if (a.mask[(((int64_t)1 << (a.depth + t - 1)) - 1) + segb + b] == 0) {
                    // a leaf of the tree: its owned samples (parent-band local index 2 (i + q) + F - 2 <-> owned pair i + q - H[t]) go out as they are
                    typedef typename VecOf<T, 2>::type P2;
                    T *lo = a.dst + root * a.nj + (int64_t)b * (a.nj >> (t - 1)) + (r0 >> (t - 1));
#pragma unroll
                    for (int q = 0; q < PPT; ++q) {
                        const int io = i + q - H[t];
                        if (io >= 0 && io < ownt)
                            *reinterpret_cast<P2 *>(lo + 2 * io) = *reinterpret_cast<const P2 *>(Ain + b * bsi + 2 * (i + q) + F - 2);
                    }
                    continue;
                }
            }
            T xv[NWIN];
            vload16<T, NWIN>(Ain + b * bsi + 2 * i, xv);                  // window of pairs i .. i+PPT-1 of parent band b
            T so[PPT], dO[PPT];
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                T sv = a.tp.h[0] * xv[2 * q + F - 2];
#pragma unroll
                for (int m = 1; m < F; ++m) sv = sv + a.tp.h[m] * xv[2 * q + F - 2 + m];
                T dv = gq(F - 1) * xv[2 * q];
#pragma unroll
                for (int m = F - 2; m >= 0; --m) dv = dv + gq(m) * xv[2 * q + F - 1 - m];
                so[q] = sv;
                dO[q] = dv;
            }
            if (!lastlev) {
                vstore16<T, PPT>(Aout + (2 * b) * bso + i, so);           // (the tail of the last group lands in the band's padding)
                vstore16<T, PPT>(Aout + (2 * b + 1) * bso + i, dO);
            } else if (i + PPT <= ownt) {                                 // H[NL] = 0: local pair i is owned pair i
                vstore16<T, PPT>(out0 + (int64_t)(2 * b) * leaf + i, so);
                vstore16<T, PPT>(out0 + (int64_t)(2 * b + 1) * leaf + i, dO);
            }
        }
        lds_barrier();
        T *tmp = Ain; Ain = Aout; Aout = tmp;
    }
}

// ---------------------------------------------------------------------------------------------------
// iwpt, NL <= 3 depths per pass over HBM: the mirror of k_wpt_fwd_multi.  A workgroup owns TS samples of a segment of the
// SHALLOWEST fused depth and stages, from each of the 2^NL leaf bands, its TS / 2^NL samples plus G[NL] on both sides (periodic
// wrap of the band resolved while staging).  Level t -> t-1: parent pair q of a band (local index 2q, 2q+1) is
// window_inv(S[q .. q+SH], D[q+SH .. q+2SH]) of its two children -- G[t] = G[t-1] / 2 + SH (rounded up to even, so that every
// level's range starts at an even sample); halos are recomputed, workgroups are independent; only level 0 goes to memory.
template <typename T, int F>
struct WptInvMultiArgs {
    const T *src; T *dst;
    int64_t nj;                     // segment length at the shallowest fused depth (a multiple of TS)
    int NL, TS;
    int G[4];                       // halo (samples per side) of a band at fused level t
    int bs[4];                      // band stride in LDS at fused level t
    int buf_elems;
    const uint8_t *mask; int depth; // as WptMultiArgs; depth = the SHALLOWEST fused depth
    TapsF<T, F> tp;
};

template <typename T, int F>
__global__ void __launch_bounds__(256) k_wpt_inv_multi(WptInvMultiArgs<T, F> a)
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int SH = (F - 2) / 2;
    constexpr int PPT = 4;                                                // parent pairs per thread and iteration
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int NL = a.NL, TS = a.TS;
    const int64_t own0 = (int64_t)blockIdx.x * TS;
    const int64_t root = own0 / a.nj;
    const int64_t r0 = own0 - root * a.nj;
    T *bufA = reinterpret_cast<T *>(smem_raw);
    T *bufB = bufA + a.buf_elems;
    // ---- stage the leaves: band b, local j <-> element (r0 >> NL) - G[NL] + j of the band (mod its length), pairs of samples ----
    {
        const int64_t blen = a.nj >> NL;
        const int Lt = (TS >> NL) + 2 * a.G[NL];
        const int hp = Lt >> 1;                                           // pairs per band (Lt is even)
        const int total = hp << NL;
        const T *seg = a.src + root * a.nj;
        const int64_t start = (r0 >> NL) - a.G[NL];
        for (int it = tid; it < total; it += 256) {
            const int b = it / hp, jp = it - b * hp;
            int64_t g = start + 2 * jp;
            if (g < 0) g += blen;
            if (g >= blen) g -= blen;
            const T2 v = *reinterpret_cast<const T2 *>(seg + (int64_t)b * blen + g);
            *reinterpret_cast<T2 *>(bufA + b * a.bs[NL] + 2 * jp) = v;
        }
    }
    lds_barrier_vm();
    T *Cin = bufA, *Pout = bufB;
    for (int t = NL; t >= 1; --t) {
        const int Lp = (TS >> (t - 1)) + 2 * a.G[t - 1];                  // parent band length (even)
        const int hp = Lp >> 1;                                           // parent pairs per band
        const int gpb = (hp + PPT - 1) / PPT;
        const int nb = 1 << (t - 1);
        const int total = gpb * nb;
        const int bsc = a.bs[t], bsp = a.bs[t - 1];
        const bool last = (t == 1);
        // parent pair q of a band is pair q + G[t] - G[t-1] / 2 of its children: SH, or SH + 1 where G[t] was rounded up to even
        const int eps = a.G[t] - a.G[t - 1] / 2 - SH;
        T *out0 = a.dst + root * a.nj + r0;
        for (int g = tid; g < total; g += 256) {
            const int b = g / gpb;
            const int q0 = (g - b * gpb) * PPT;
            if (a.mask && a.mask[(((int64_t)1 << (a.depth + t - 1)) - 1) + (root << (t - 1)) + b] == 0) {
                // parent band b is (part of) a leaf: its samples are taken from the source as they are -- local pair q <-> element
                // (r0 >> (t-1)) - G[t-1] + 2 q of the band (mod its length).  (If an ancestor is a leaf too nobody reads this band.)
                const int64_t plen = a.nj >> (t - 1);
                const T *pb = a.src + root * a.nj + (int64_t)b * plen;
                const int64_t start = (r0 >> (t - 1)) - a.G[t - 1];
#pragma unroll
                for (int p = 0; p < PPT; ++p) {
                    if (q0 + p < hp) {
                        int64_t gi = start + 2 * (q0 + p);
                        if (gi < 0) gi += plen;
                        if (gi >= plen) gi -= plen;
                        const T2 v = *reinterpret_cast<const T2 *>(pb + gi);
                        if (!last) *reinterpret_cast<T2 *>(Pout + b * bsp + 2 * (q0 + p)) = v;
                        else *reinterpret_cast<T2 *>(out0 + 2 * (q0 + p)) = v;
                    }
                }
                continue;
            }
            const T *S = Cin + (2 * b) * bsc + q0 + eps;
            const T *D = Cin + (2 * b + 1) * bsc + q0 + SH + eps;
            T sv[PPT + SH], dv[PPT + SH];
#pragma unroll
            for (int i = 0; i < PPT + SH; ++i) { sv[i] = S[i]; dv[i] = D[i]; }
            T xo[2 * PPT];
#pragma unroll
            for (int p = 0; p < PPT; ++p) {
                T sw[SH + 1], dw[SH + 1];
#pragma unroll
                for (int i = 0; i <= SH; ++i) { sw[i] = sv[p + i]; dw[i] = dv[p + i]; }
                window_inv<T, F>(sw, dw, a.tp, xo[2 * p], xo[2 * p + 1]);
            }
            if (!last) {
#pragma unroll
                for (int p = 0; p < PPT; ++p)
                    *reinterpret_cast<T2 *>(Pout + b * bsp + 2 * (q0 + p)) = T2{xo[2 * p], xo[2 * p + 1]};     // (past hp: the band's padding)
            } else if (q0 + PPT <= hp) {
                vstore16<T, 2 * PPT>(out0 + 2 * q0, xo);
            } else {
#pragma unroll
                for (int p = 0; p < PPT; ++p)
                    if (q0 + p < hp) *reinterpret_cast<T2 *>(out0 + 2 * (q0 + p)) = T2{xo[2 * p], xo[2 * p + 1]};
            }
        }
        lds_barrier();
        T *tmp = Cin; Cin = Pout; Pout = tmp;
    }
}

// ---------------------------------------------------------------------------------------------------
template <typename T, int F>
struct WptTailArgs {
    const T *src; T *dst;
    int lgts;                       // chunk = 2^lgts samples per workgroup
    int lgm;                        // segment length 2^lgm at the SHALLOWEST depth handled here (<= chunk)
    int ndepth;                     // depths handled (segments shrink / grow by 2 per depth), 1 <= ndepth <= lgm
    const uint8_t *mask; int depth; // as WptMultiArgs; depth = the shallowest depth handled (segments of 2^lgm)
    TapsF<T, F> tp;
};

template <typename T>
__device__ __forceinline__ void wpt_chunk_in(const T *src, T *A, int total, int tid, int nthr)
{
    constexpr int VEC = 16 / sizeof(T);
    for (int c = tid; c < total / VEC; c += 4 * nthr) {
        T v[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c + u * nthr < total / VEC) vload<T, VEC>(src + (int64_t)(c + u * nthr) * VEC, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c + u * nthr < total / VEC) vstore16<T, VEC>(A + (c + u * nthr) * VEC, v[u]);
    }
}
template <typename T>
__device__ __forceinline__ void wpt_chunk_out(T *dst, const T *A, int total, int tid, int nthr)
{
    constexpr int VEC = 16 / sizeof(T);
    for (int c = tid; c < total / VEC; c += nthr) {
        T v[VEC];
        vload16<T, VEC>(A + c * VEC, v);
        vstore<T, VEC>(dst + (int64_t)c * VEC, v);
    }
}

template <typename T, int F>
__global__ void __launch_bounds__(512) k_wpt_fwd_tail(WptTailArgs<T, F> a)
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int NW = (F == 2) ? 2 : 2 * F - 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TS = 1 << a.lgts;
    T *A = reinterpret_cast<T *>(smem_raw);
    T *B = A + TS;
    wpt_chunk_in<T>(a.src + (int64_t)blockIdx.x * TS, A, TS, tid, nthr);
    lds_barrier_vm();
    for (int dep = 0; dep < a.ndepth; ++dep) {
        const int lgm = a.lgm - dep, m = 1 << lgm, hm = m >> 1;
        for (int p = tid; p < (TS >> 1); p += nthr) {
            const int sg = p >> (lgm - 1), k = p & (hm - 1);
            const T *sb = A + (sg << lgm);
            if (a.mask && a.mask[(((int64_t)1 << (a.depth + dep)) - 1) + ((int64_t)blockIdx.x << (a.lgts - lgm)) + sg] == 0) {
                T *ob = B + (sg << lgm);               // (part of) a leaf: passed through
                ob[k] = sb[k];
                ob[hm + k] = sb[hm + k];
                continue;
            }
            T xv[NW];
#pragma unroll
            for (int e = 0; e < NW / 2; ++e) {
                const T2 v = *reinterpret_cast<const T2 *>(sb + ((2 * k - (F - 2) + 2 * e) & (m - 1)));
                xv[2 * e] = v.x; xv[2 * e + 1] = v.y;
            }
            // (s, d) of pair k from xv[e] = x[(2k - (F-2) + e) mod m]   (F = 2: the pair itself)
            T s = a.tp.h[0] * xv[(F == 2) ? 0 : F - 2];
#pragma unroll
            for (int q = 1; q < F; ++q) s = s + a.tp.h[q] * xv[((F == 2) ? 0 : F - 2) + q];
            T d = a.tp.g[F - 1] * xv[0];
#pragma unroll
            for (int q = F - 2; q >= 0; --q) d = d + a.tp.g[q] * xv[F - 1 - q];
            T *ob = B + (sg << lgm);
            ob[k] = s;
            ob[hm + k] = d;
        }
        lds_barrier();
        T *t = A; A = B; B = t;
    }
    wpt_chunk_out<T>(a.dst + (int64_t)blockIdx.x * TS, A, TS, tid, nthr);
}

// iwpt: depths from the deepest (segments of 2^(lgm - ndepth + 1)) up to segments of 2^lgm
template <typename T, int F>
__global__ void __launch_bounds__(512) k_wpt_inv_tail(WptTailArgs<T, F> a)
{
    typedef typename VecOf<T, 2>::type T2;
    constexpr int SH = (F - 2) / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int TS = 1 << a.lgts;
    T *A = reinterpret_cast<T *>(smem_raw);
    T *B = A + TS;
    wpt_chunk_in<T>(a.src + (int64_t)blockIdx.x * TS, A, TS, tid, nthr);
    lds_barrier_vm();
    for (int dep = a.ndepth - 1; dep >= 0; --dep) {
        const int lgm = a.lgm - dep, m = 1 << lgm, hm = m >> 1;
        for (int p = tid; p < (TS >> 1); p += nthr) {
            const int sg = p >> (lgm - 1), pp = p & (hm - 1);
            const T *sb = A + (sg << lgm);
            if (a.mask && a.mask[(((int64_t)1 << (a.depth + dep)) - 1) + ((int64_t)blockIdx.x << (a.lgts - lgm)) + sg] == 0) {
                *reinterpret_cast<T2 *>(B + (sg << lgm) + 2 * pp) = *reinterpret_cast<const T2 *>(sb + 2 * pp);
                continue;
            }
            T sw[SH + 1], dw[SH + 1];
#pragma unroll
            for (int q = 0; q <= SH; ++q) {
                sw[q] = sb[(pp - SH + q) & (hm - 1)];
                dw[q] = sb[hm + ((pp + q) & (hm - 1))];
            }
            T xe, xo;
            window_inv<T, F>(sw, dw, a.tp, xe, xo);
            *reinterpret_cast<T2 *>(B + (sg << lgm) + 2 * pp) = T2{xe, xo};
        }
        lds_barrier();
        T *t = A; A = B; B = t;
    }
    wpt_chunk_out<T>(a.dst + (int64_t)blockIdx.x * TS, A, TS, tid, nthr);
}

// ------------------------------------------------------------------------------------------
static inline bool is_pow2(int64_t v) { return v >= 1 && (v & (v - 1)) == 0; }
static inline int ilog2(int64_t v) { int l = 0; while (((int64_t)1 << (l + 1)) <= v) ++l; return l; }

template <typename T>
int wpt_tile_samples() { return (int)opt("WL_WPT_TS", (long long)(16384 / sizeof(T))); }

// NL fused depths starting at segment length nj: tiles must partition a segment, the leaves' pieces must stay 16-byte aligned
template <typename T>
bool wpt_fwd_multi_ok(int F, int64_t n, int64_t nj, int NL)
{
    constexpr int VEC = 16 / sizeof(T);
    if (F < 2 || F > 10 || (F & 1) || NL < 1 || NL > 3) return false;
    const int TS = wpt_tile_samples<T>();
    if (!is_pow2(TS) || TS < 1024 / (int)sizeof(T) * 4) return false;
    if (nj < TS || (nj % TS) != 0 || (n % nj) != 0) return false;
    if (((TS >> NL) % VEC) != 0 || ((nj >> NL) % VEC) != 0) return false;
    if ((F - 2) * ((1 << NL) - 1) >= nj) return false;                    // the halo wraps at most once
    return n / TS < ((int64_t)1 << 31);
}

template <typename T, int F>
static hipError_t launch_wpt_multi_f(hipStream_t st, const Taps<T> &taps, const T *src, T *dst, int64_t n, int64_t nj, int NL, const uint8_t *mask)
{
    constexpr int VEC = 16 / sizeof(T);
    WptMultiArgs<T, F> a;
    a.src = src; a.dst = dst; a.nj = nj; a.NL = NL; a.TS = wpt_tile_samples<T>();
    a.mask = mask; a.depth = ilog2(n / nj);
    int H[4];
    H[NL] = 0;
    for (int t = NL; t >= 1; --t) H[t - 1] = 2 * H[t] + (F - 2);
    int maxlen = 0;
    for (int t = 0; t <= 3; ++t) a.bs[t] = 0;
    for (int t = 0; t <= NL; ++t) {
        // a band of level t: (TS >> t) + 2 H[t] samples + the over-read of the last thread group of the level below it
        const int len = (a.TS >> t) + 2 * H[t] + 2 * VEC + 2 * F + VEC;
        a.bs[t] = (len + VEC - 1) / VEC * VEC;
        const int tot = a.bs[t] << t;
        if (tot > maxlen) maxlen = tot;
    }
    a.buf_elems = (maxlen + 15) & ~15;
    a.tp = shrink<T, F>(taps);
    const size_t shmem = 2 * (size_t)a.buf_elems * sizeof(T);
    hipLaunchKernelGGL((k_wpt_fwd_multi<T, F>), dim3((unsigned)(n / a.TS)), dim3(256), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t wpt_fwd_multi_launch(hipStream_t st, const Taps<T> &taps, const T *src, T *dst, int64_t n, int64_t nj, int NL, const uint8_t *mask)
{
    switch (taps.F) {
    case 2: return launch_wpt_multi_f<T, 2>(st, taps, src, dst, n, nj, NL, mask);
    case 4: return launch_wpt_multi_f<T, 4>(st, taps, src, dst, n, nj, NL, mask);
    case 6: return launch_wpt_multi_f<T, 6>(st, taps, src, dst, n, nj, NL, mask);
    case 8: return launch_wpt_multi_f<T, 8>(st, taps, src, dst, n, nj, NL, mask);
    case 10: return launch_wpt_multi_f<T, 10>(st, taps, src, dst, n, nj, NL, mask);
    default: return hipErrorInvalidValue;
    }
}

template <typename T>
static void wpt_inv_halos(int F, int NL, int (&G)[4])
{
    const int SH = (F - 2) / 2;
    G[0] = 0;
    for (int t = 1; t <= 3; ++t) G[t] = (t <= NL) ? ((G[t - 1] / 2 + SH + 1) & ~1) : 0;
}

// NL fused depths whose SHALLOWEST segment length is nj
template <typename T>
bool wpt_inv_multi_ok(int F, int64_t n, int64_t nj, int NL)
{
    if (!wpt_fwd_multi_ok<T>(F, n, nj, NL)) return false;
    int G[4];
    wpt_inv_halos<T>(F, NL, G);
    return (int64_t)G[NL] < (nj >> NL) && ((nj >> NL) % 2) == 0;
}

template <typename T, int F>
static hipError_t launch_wpt_inv_multi_f(hipStream_t st, const Taps<T> &taps, const T *src, T *dst, int64_t n, int64_t nj, int NL, const uint8_t *mask)
{
    constexpr int VEC = 16 / sizeof(T);
    WptInvMultiArgs<T, F> a;
    a.src = src; a.dst = dst; a.nj = nj; a.NL = NL; a.TS = wpt_tile_samples<T>();
    a.mask = mask; a.depth = ilog2(n / nj);
    wpt_inv_halos<T>(F, NL, a.G);
    int maxlen = 0;
    for (int t = 0; t <= 3; ++t) a.bs[t] = 0;
    for (int t = 0; t <= NL; ++t) {
        // a band of level t + what the last thread group of the level above it reads / writes past the end
        const int len = (a.TS >> t) + 2 * a.G[t] + 2 * 4 + F + VEC;
        a.bs[t] = (len + VEC - 1) / VEC * VEC;
        const int tot = a.bs[t] << t;
        if (tot > maxlen) maxlen = tot;
    }
    a.buf_elems = (maxlen + 15) & ~15;
    a.tp = shrink<T, F>(taps);
    const size_t shmem = 2 * (size_t)a.buf_elems * sizeof(T);
    hipLaunchKernelGGL((k_wpt_inv_multi<T, F>), dim3((unsigned)(n / a.TS)), dim3(256), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t wpt_inv_multi_launch(hipStream_t st, const Taps<T> &taps, const T *src, T *dst, int64_t n, int64_t nj, int NL, const uint8_t *mask)
{
    switch (taps.F) {
    case 2: return launch_wpt_inv_multi_f<T, 2>(st, taps, src, dst, n, nj, NL, mask);
    case 4: return launch_wpt_inv_multi_f<T, 4>(st, taps, src, dst, n, nj, NL, mask);
    case 6: return launch_wpt_inv_multi_f<T, 6>(st, taps, src, dst, n, nj, NL, mask);
    case 8: return launch_wpt_inv_multi_f<T, 8>(st, taps, src, dst, n, nj, NL, mask);
    case 10: return launch_wpt_inv_multi_f<T, 10>(st, taps, src, dst, n, nj, NL, mask);
    default: return hipErrorInvalidValue;
    }
}

// tail: segments of nj = 2^k <= chunk samples, ndepth depths (the last one splits / merges segments of nj >> (ndepth - 1) >= 2)
template <typename T>
bool wpt_tail_ok(int F, int64_t n, int64_t nj, int ndepth)
{
    constexpr int VEC = 16 / sizeof(T);
    if (F < 2 || F > 10 || (F & 1) || ndepth < 1) return false;
    const int TS = wpt_tile_samples<T>();
    if (!is_pow2(nj) || nj < 2 || (nj >> (ndepth - 1)) < 2) return false;
    const int64_t chunk = (n < TS) ? n : TS;
    if (!is_pow2(chunk) || chunk < VEC || nj > chunk || (n % chunk) != 0) return false;
    return n / chunk < ((int64_t)1 << 31);
}

template <typename T, int F>
static hipError_t launch_wpt_tail_f(hipStream_t st, const Taps<T> &taps, int fw, const T *src, T *dst, int64_t n, int64_t nj, int ndepth, const uint8_t *mask)
{
    WptTailArgs<T, F> a;
    const int TS = wpt_tile_samples<T>();
    const int64_t chunk = (n < TS) ? n : TS;
    a.src = src; a.dst = dst; a.lgts = ilog2(chunk); a.lgm = ilog2(nj); a.ndepth = ndepth;
    a.mask = mask; a.depth = ilog2(n / nj);
    a.tp = shrink<T, F>(taps);
    const size_t shmem = 2 * (size_t)chunk * sizeof(T);
    const int threads = chunk >= 2048 ? 512 : (chunk >= 512 ? 256 : 64);
    if (fw) hipLaunchKernelGGL((k_wpt_fwd_tail<T, F>), dim3((unsigned)(n / chunk)), dim3(threads), shmem, st, a);
    else hipLaunchKernelGGL((k_wpt_inv_tail<T, F>), dim3((unsigned)(n / chunk)), dim3(threads), shmem, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t wpt_tail_launch(hipStream_t st, const Taps<T> &taps, int fw, const T *src, T *dst, int64_t n, int64_t nj, int ndepth, const uint8_t *mask)
{
    switch (taps.F) {
    case 2: return launch_wpt_tail_f<T, 2>(st, taps, fw, src, dst, n, nj, ndepth, mask);
    case 4: return launch_wpt_tail_f<T, 4>(st, taps, fw, src, dst, n, nj, ndepth, mask);
    case 6: return launch_wpt_tail_f<T, 6>(st, taps, fw, src, dst, n, nj, ndepth, mask);
    case 8: return launch_wpt_tail_f<T, 8>(st, taps, fw, src, dst, n, nj, ndepth, mask);
    case 10: return launch_wpt_tail_f<T, 10>(st, taps, fw, src, dst, n, nj, ndepth, mask);
    default: return hipErrorInvalidValue;
    }
}

template int wpt_tile_samples<float>();
template int wpt_tile_samples<double>();
template bool wpt_fwd_multi_ok<float>(int, int64_t, int64_t, int);
template bool wpt_fwd_multi_ok<double>(int, int64_t, int64_t, int);
template hipError_t wpt_fwd_multi_launch<float>(hipStream_t, const Taps<float> &, const float *, float *, int64_t, int64_t, int, const uint8_t *);
template hipError_t wpt_fwd_multi_launch<double>(hipStream_t, const Taps<double> &, const double *, double *, int64_t, int64_t, int, const uint8_t *);
template bool wpt_inv_multi_ok<float>(int, int64_t, int64_t, int);
template bool wpt_inv_multi_ok<double>(int, int64_t, int64_t, int);
template hipError_t wpt_inv_multi_launch<float>(hipStream_t, const Taps<float> &, const float *, float *, int64_t, int64_t, int, const uint8_t *);
template hipError_t wpt_inv_multi_launch<double>(hipStream_t, const Taps<double> &, const double *, double *, int64_t, int64_t, int, const uint8_t *);
template bool wpt_tail_ok<float>(int, int64_t, int64_t, int);
template bool wpt_tail_ok<double>(int, int64_t, int64_t, int);
template hipError_t wpt_tail_launch<float>(hipStream_t, const Taps<float> &, int, const float *, float *, int64_t, int64_t, int, const uint8_t *);
template hipError_t wpt_tail_launch<double>(hipStream_t, const Taps<double> &, int, const double *, double *, int64_t, int64_t, int, const uint8_t *);

}  // namespace wl
