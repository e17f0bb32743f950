// wl_lift.hip -- lifting DWT/IDWT fast paths for lines (1-D vectors, batched columns).
//
//   k_lift1d_stream  one lifting level of a line, forward or inverse, ALL steps + split/merge +
//                    normalisation fused (the reference makes >= 6 passes over the level:
//                    split!, one pass per step, normalize!; transforms_lifting.jl:57-64).
//                    Each lane holds 4 (s, d) pairs = 8 consecutive samples (two 16-byte loads);
//                    a step's operands outside the lane come from the neighbouring lane by DPP
//                    AFTER that lane's own update, so nothing is recomputed; the two edge lanes of
//                    a wave absorb the dependency cone (reach <= 4 pairs), lanes 1..62 are stored.
//                    One read + one write of the level: 8 B/sample (f32).
//   k_tail_lift      every remaining level of a line <= 16 Ki f32 / 8 Ki f64 samples inside one
//                    workgroup with the line in LDS (any scheme, true periodic indexing).
//   k_lift_axis_stream   one lifting level along a strided axis (dim 2 / dim 3 of square and cubic arrays) as a
//                    register cascade; k_lift_short_lines: the dim-1 pass for lines of 2..512 samples.
//
// Rounding follows the reference exactly: an element whose operands do not wrap is updated as
// x += (c1*a + c2*b [+ c3*c]) (lift_inbounds!, transforms_lifting.jl:455-483), a wrapped one as
// x += c1*a; x += c2*b; ... (lift_perboundary!, :437-451); no FMA contraction.
#include "wl_fast.h"
#include "wl_lift_shapes.h"
#include "wl_dev.h"


namespace wl {

template <int ID>
static bool shape_matches(int nsteps, const int *upd, const int *nc, const int *sh)
{
    if (nsteps != Shape<ID>::NS) return false;
    for (int i = 0; i < nsteps; ++i)
        if (upd[i] != Shape<ID>::S[i].upd || nc[i] != Shape<ID>::S[i].nc || sh[i] != Shape<ID>::S[i].sh) return false;
    return true;
}

__device__ __forceinline__ int l_dpp_next(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int l_dpp_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int l_dpp_partner(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float l_partner(float v) { return __int_as_float(l_dpp_partner(__float_as_int(v))); }
__device__ __forceinline__ double l_partner(double v) { return __hiloint2double(l_dpp_partner(__double2hiint(v)), l_dpp_partner(__double2loint(v))); }
__device__ __forceinline__ float l_next(float v) { return __int_as_float(l_dpp_next(__float_as_int(v))); }
__device__ __forceinline__ float l_prev(float v) { return __int_as_float(l_dpp_prev(__float_as_int(v))); }
__device__ __forceinline__ double l_next(double v) { return __hiloint2double(l_dpp_next(__double2hiint(v)), l_dpp_next(__double2loint(v))); }
__device__ __forceinline__ double l_prev(double v) { return __hiloint2double(l_dpp_prev(__double2hiint(v)), l_dpp_prev(__double2loint(v))); }

template <typename T>
struct Lift1DArgs {
    // forward: src = line of n samples; sdst/ddst = approximation / detail destinations (n/2 each)
    // inverse: ssrc/dsrc = approximation / detail sources (n/2 each); dst = line of n samples
    const T *a; int64_t a_ls;       // fw: src            inv: ssrc
    const T *b; int64_t b_ls;       // fw: unused         inv: dsrc
    T *o0; int64_t o0_ls;           // fw: sdst           inv: dst
    T *o1; int64_t o1_ls;           // fw: ddst           inv: unused
    int64_t n;                      // line length (multiple of 8, >= 512)
    int64_t ntiles;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <typename T>
__device__ __forceinline__ void ld8(const T *p, T (&v)[8])
{
    constexpr int C = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < 8 / C; ++c) {
        V t = *reinterpret_cast<const V *>(p + c * C);
#pragma unroll
        for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void stN(T *p, const T (&v)[N])
{
    constexpr int C = 16 / sizeof(T);
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        *reinterpret_cast<V *>(p + c * C) = t;
    }
}

template <typename T, int N>
__device__ __forceinline__ void ldv_l(const T *p, T (&v)[N])
{
    constexpr int C = ((int)(16 / sizeof(T)) < N) ? (int)(16 / sizeof(T)) : N;
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t = *reinterpret_cast<const V *>(p + c * C);
#pragma unroll
        for (int i = 0; i < C; ++i) v[c * C + i] = t[i];
    }
}
template <typename T, int N>
__device__ __forceinline__ void stv_l(T *p, const T (&v)[N])
{
    constexpr int C = ((int)(16 / sizeof(T)) < N) ? (int)(16 / sizeof(T)) : N;
    typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
    for (int c = 0; c < N / C; ++c) {
        V t;
#pragma unroll
        for (int i = 0; i < C; ++i) t[i] = v[c * C + i];
        *reinterpret_cast<V *>(p + c * C) = t;
    }
}

template <typename T, int ID, int FW>
__global__ void __launch_bounds__(256) k_lift1d_stream(Lift1DArgs<T> a)
{
    constexpr int VP = 62 * 4;
    typedef Shape<ID> SH;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t n = a.n, half = n >> 1;
    const int64_t line = blockIdx.y;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k0 = tile * VP + (int64_t)(lane - 1) * 4;      // first pair of this lane (may wrap)
        int64_t kw = k0;
        if (kw < 0) kw += half;
        if (kw >= half) kw -= half;
        T s[4], d[4];
        if (FW) {
            T v[8];
            ld8<T>(a.a + line * a.a_ls + 2 * kw, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = v[2 * j]; d[j] = v[2 * j + 1]; }      // Util.split!
        } else {
            T sv[4], dv[4];
            constexpr int C = 16 / sizeof(T);
            typedef T V __attribute__((ext_vector_type(C)));
#pragma unroll
            for (int c = 0; c < 4 / C; ++c) {
                V t0 = *reinterpret_cast<const V *>(a.a + line * a.a_ls + kw + c * C);
                V t1 = *reinterpret_cast<const V *>(a.b + line * a.b_ls + kw + c * C);
#pragma unroll
                for (int i = 0; i < C; ++i) { sv[c * C + i] = t0[i]; dv[c * C + i] = t1[i]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] = a.norm1 * sv[j]; d[j] = a.norm2 * dv[j]; }   // normalize! (inverse first)
        }
#pragma unroll
        for (int st = 0; st < SH::NS; ++st) {
            const int upd = SH::S[st].upd, nc = SH::S[st].nc, sh = SH::S[st].sh;
            // operands: the other half at pair offsets (kk - sh), kk = 0..nc-1
            T *tgt = upd ? d : s;
            const T *op = upd ? s : d;
            // neighbour lanes' current operand values (after their own previous updates)
            T opn[4], opp[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { opn[j] = (T)0; opp[j] = (T)0; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // needed iff some jj + kk - sh lands on this element of the neighbour
                bool need_n = false, need_p = false;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int kk = 0; kk < 3; ++kk)
                        if (kk < nc) {
                            int off = jj + kk - sh;
                            if (off - 4 == j) need_n = true;
                            if (off + 4 == j) need_p = true;
                        }
                if (need_n) opn[j] = l_next(op[j]);
                if (need_p) opp[j] = l_prev(op[j]);
            }
            T res[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                T o[3];
#pragma unroll
                for (int kk = 0; kk < 3; ++kk) {
                    o[kk] = (T)0;
                    if (kk < nc) {
                        const int off = jj + kk - sh;
                        o[kk] = (off < 0) ? opp[off + 4] : (off > 3 ? opn[off - 4] : op[off]);
                    }
                }
                // in bounds <=> no operand wraps around the line: 0 <= j - sh and j + nc - 1 - sh <= half - 1
                const int64_t jg = kw + jj - sh;
                const bool inb = (jg >= 0) && (jg + nc - 1 <= half - 1);
                const T x = tgt[jj];
                T acc = a.c[st][0] * o[0];
                if (nc > 1) acc = acc + a.c[st][1] * o[1];
                if (nc > 2) acc = acc + a.c[st][2] * o[2];
                const T xin = x + acc;
                T xb = x + a.c[st][0] * o[0];
                if (nc > 1) xb = xb + a.c[st][1] * o[1];
                if (nc > 2) xb = xb + a.c[st][2] * o[2];
                res[jj] = inb ? xin : xb;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) tgt[jj] = res[jj];
        }
        const bool valid = lane >= 1 && lane <= 62 && k0 < half;
        if (FW) {
            T so[4], dO[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { so[j] = s[j] * a.norm1; dO[j] = d[j] * a.norm2; }     // normalize!
            if (valid) {
                stN<T, 4>(a.o0 + line * a.o0_ls + k0, so);
                stN<T, 4>(a.o1 + line * a.o1_ls + k0, dO);
            }
        } else {
            T v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = s[j]; v[2 * j + 1] = d[j]; }               // Util.merge!
            if (valid) stN<T, 8>(a.o0 + line * a.o0_ls + 2 * k0, v);
        }
    }
}


// --------------------------------------------------------------------------------------------------
// Three forward lifting levels per pass over HBM, entirely in registers: after level t the lane keeps
// its normalised approximations (4, then 2, then 1 pairs per lane) and runs level t+1 on them; the
// out-of-lane operands again come from the neighbouring lanes by DPP (distance up to 2 lanes at the
// last level).  The dependency cone costs 4 lanes at each wave edge (lanes 4..59 own 224 pairs).
// Traffic: read n, write n for three levels (level by level: 3.5 n) and a third of the launches.
constexpr __host__ __device__ int l_floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

template <typename T, int PPL>
__device__ __forceinline__ T lane_operand(const T (&op)[PPL], int off)
{
    // value of op at pair offset `off` relative to this lane's first pair (off is a compile-time constant
    // at every call site after unrolling)
    const int delta = l_floordiv(off, PPL);
    const int e = off - delta * PPL;
    T v = op[e];
    if (delta == 1) v = l_next(v);
    else if (delta == 2) v = l_next(l_next(v));
    else if (delta == -1) v = l_prev(v);
    else if (delta == -2) v = l_prev(l_prev(v));
    return v;
}

// FAST: the caller has established that no operand of any lane wraps around the ends of the line (every update takes the
// in-bounds form): the boundary form and the 64-bit position tests disappear
template <typename T, int ID, int PPL, bool FAST = false>
__device__ __forceinline__ void lift_steps_lane(T (&s)[PPL], T (&d)[PPL], const T (&c)[LIFT_FAST_STEPS][WL_MAX_NCOEF],
                                                int64_t kfirst, int64_t half)
{
    typedef Shape<ID> SH;
#pragma unroll
    for (int st = 0; st < SH::NS; ++st) {
        const int upd = SH::S[st].upd, nc = SH::S[st].nc, sh = SH::S[st].sh;
        T res[PPL];
#pragma unroll
        for (int jj = 0; jj < PPL; ++jj) {
            T o[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                o[kk] = (T)0;
                if (kk < nc) o[kk] = upd ? lane_operand<T, PPL>(s, jj + kk - sh) : lane_operand<T, PPL>(d, jj + kk - sh);
            }
            const int64_t jg = kfirst + jj - sh;
            const bool inb = (jg >= 0) && (jg + nc - 1 <= half - 1);
            const T x = upd ? d[jj] : s[jj];
            T acc = c[st][0] * o[0];
            if (nc > 1) acc = acc + c[st][1] * o[1];
            if (nc > 2) acc = acc + c[st][2] * o[2];
            const T xin = x + acc;
            if constexpr (FAST) {
                res[jj] = xin;
            } else {
                T xb = x + c[st][0] * o[0];
                if (nc > 1) xb = xb + c[st][1] * o[1];
                if (nc > 2) xb = xb + c[st][2] * o[2];
                res[jj] = inb ? xin : xb;
            }
        }
#pragma unroll
        for (int jj = 0; jj < PPL; ++jj) { if (upd) d[jj] = res[jj]; else s[jj] = res[jj]; }
    }
}

template <typename T>
struct Lift3Args {
    const T *src; int64_t src_ls;
    T *y; int64_t y_ls;             // details: level t (1..3) at y[(n >> t) + k]
    T *d1; int64_t d1_ls;           // level-1 detail destination (y + n/2, or a staging buffer)
    T *sdst; int64_t s_ls;          // approximation after three levels
    int64_t n;
    int64_t ntiles;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

// (LVL1 only gives the launch that consumes the full-size input its own symbol: rocprofv3 --stats then reports it separately)
template <typename T, int ID, int LVL1 = 0>
__global__ void __launch_bounds__(256) k_lift1d_fwd3(Lift3Args<T> a)
{
    constexpr int ML = 4, VP = (64 - 2 * ML) * 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t n = a.n, half = n >> 1;
    const int64_t line = blockIdx.y;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k0 = tile * VP + (int64_t)(lane - ML) * 4;
        int64_t kw = k0;
        if (kw < 0) kw += half;
        if (kw >= half) kw -= half;
        T v[8];
        ldg_pol<WL_P_LIFT3_LD != 0, T, 8>(a.src + line * a.src_ls + 2 * kw, v);
        T s1[4], d1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[j] = v[2 * j]; d1[j] = v[2 * j + 1]; }
        lift_steps_lane<T, ID, 4>(s1, d1, a.c, kw, half);
        T s2[2], d2[2], dO1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) dO1[j] = d1[j] * a.norm2;
#pragma unroll
        for (int j = 0; j < 2; ++j) { s2[j] = s1[2 * j] * a.norm1; d2[j] = s1[2 * j + 1] * a.norm1; }
        lift_steps_lane<T, ID, 2>(s2, d2, a.c, kw >> 1, half >> 1);
        T s3[1], d3[1], dO2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) dO2[j] = d2[j] * a.norm2;
        s3[0] = s2[0] * a.norm1;
        d3[0] = s2[1] * a.norm1;
        lift_steps_lane<T, ID, 1>(s3, d3, a.c, kw >> 2, half >> 2);
        if (lane >= ML && lane < 64 - ML && k0 < half) {
            stg_pol<WL_P_LIFT3_ST != 0, T, 4>(a.d1 + line * a.d1_ls + k0, dO1);
            typedef T V2 __attribute__((ext_vector_type(2)));
            V2 t2; t2[0] = dO2[0]; t2[1] = dO2[1];
            *reinterpret_cast<V2 *>(a.y + line * a.y_ls + (n >> 2) + (k0 >> 1)) = t2;
            a.y[line * a.y_ls + (n >> 3) + (k0 >> 2)] = d3[0] * a.norm2;
            a.sdst[line * a.s_ls + (k0 >> 2)] = s3[0] * a.norm1;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Three INVERSE lifting levels per pass over HBM (mirror of k_lift1d_fwd3): the lane starts from one pair of the
// deepest level, reconstructs 2 then 4 approximations in registers, merges in the details of the two shallower
// levels on the way and stores 8 samples.  Traffic: read n, write n for three levels (level by level: 3.5 n).
template <typename T>
struct LiftInv3Args {
    const T *s3; int64_t s3_ls;     // approximation of the deepest of the three levels (n/8 per line)
    const T *x; int64_t x_ls;       // coefficient lines: d3 at x[n/8 + k], d2 at x[n/4 + k], d1 at x[n/2 + k]
    T *dst; int64_t o_ls;           // output lines (n samples)
    int64_t n;                      // OUTPUT line length
    int64_t ntiles;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;                 // already inverted by make_scheme
};

template <typename T, int ID>
__global__ void __launch_bounds__(256) k_lift1d_inv3(LiftInv3Args<T> a)
{
    constexpr int ML = 4, VP3 = 64 - 2 * ML;          // deepest-level pairs owned by a wave tile (one per lane)
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t n = a.n, h3 = n >> 3;
    const int64_t line = blockIdx.y;
    const T *x = a.x + line * a.x_ls;
    for (int64_t tile = wave; tile < a.ntiles; tile += nwaves) {
        const int64_t k3 = tile * VP3 + (lane - ML);
        int64_t kw = k3;
        if (kw < 0) kw += h3;
        if (kw >= h3) kw -= h3;
        T s3[1], d3[1], d2[2], d1[4];
        s3[0] = a.norm1 * a.s3[line * a.s3_ls + kw];
        d3[0] = a.norm2 * x[h3 + kw];
        ldg_pol<WL_P_LIFTI3_LD != 0, T, 2>(x + 2 * h3 + 2 * kw, d2);
        ldg_pol<WL_P_LIFTI3_LD != 0, T, 4>(x + 4 * h3 + 4 * kw, d1);
        lift_steps_lane<T, ID, 1>(s3, d3, a.c, kw, h3);
        T s2[2], d2n[2];
        s2[0] = a.norm1 * s3[0]; s2[1] = a.norm1 * d3[0];                 // merge!, then normalize! of the next level
#pragma unroll
        for (int j = 0; j < 2; ++j) d2n[j] = a.norm2 * d2[j];
        lift_steps_lane<T, ID, 2>(s2, d2n, a.c, 2 * kw, 2 * h3);
        T s1[4], d1n[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) { s1[2 * j] = a.norm1 * s2[j]; s1[2 * j + 1] = a.norm1 * d2n[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) d1n[j] = a.norm2 * d1[j];
        lift_steps_lane<T, ID, 4>(s1, d1n, a.c, 4 * kw, 4 * h3);
        T out[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { out[2 * j] = s1[j]; out[2 * j + 1] = d1n[j]; }
        if (lane >= ML && lane < 64 - ML && k3 < h3) stg_pol<WL_P_LIFTI3_ST != 0, T, 8>(a.dst + line * a.o_ls + 8 * k3, out);
    }
}

// --------------------------------------------------------------------------------------------------
// LDS tail: all remaining levels of one line per workgroup (any scheme)
template <typename T>
struct LiftTailArgs {
    const T *src; int64_t src_item;     // fw: the line to transform (n0 samples)
    T *y; int64_t y_item;               // output line; fw: details + final approximation; inv: samples
    const T *ll; int64_t ll_item;       // inv only: approximation source of the deepest level (may equal y's LL)
    int n0;                             // fw: input length;  inv: OUTPUT length of the last (shallowest) level done here
    int nlev;
    int cap;
};

// Synchronisation inside the tail kernels: a workgroup barrier while several waves work on the line,
// just "my own LDS traffic has landed" once only wave 0 is left (small levels: the other waves have
// returned, and a barrier round trip per lifting step would dominate the run time).
__device__ __forceinline__ void tail_sync(bool multi)
{
    if (multi) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void tail_sync_vm(bool multi)
{
    if (multi) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <typename T>
__device__ __forceinline__ void tail_lift_steps(T *w, int half, const LiftScheme<T> &sc, int tid, int nthr, bool multi)
{
    for (int st = 0; st < sc.nsteps; ++st) {
        const LiftStep<T> &sp = sc.step[st];
        T *tgt = w + (sp.is_update ? half : 0);
        const T *op = w + (sp.is_update ? 0 : half);
        for (int j = tid; j < half; j += nthr) {
            const int j0 = j - sp.shift;
            const bool inb = (j0 >= 0) && (j0 + sp.nc - 1 <= half - 1);
            T x = tgt[j];
            if (inb) {
                T acc = sp.c[0] * op[j0];
                if (sp.nc > 1) acc = acc + sp.c[1] * op[j0 + 1];
                if (sp.nc > 2) acc = acc + sp.c[2] * op[j0 + 2];
                x = x + acc;
            } else {
                for (int k = 0; k < sp.nc; ++k) {
                    int i = j0 + k;
                    while (i < 0) i += half;
                    while (i >= half) i -= half;
                    x = x + sp.c[k] * op[i];
                }
            }
            tgt[j] = x;
        }
        tail_sync(multi);
    }
}

constexpr int kSingleWavePairs = 128;     // levels with at most this many pairs run on wave 0 alone

template <typename T, int FW>
__global__ void __launch_bounds__(1024) k_tail_lift(LiftTailArgs<T> a, LiftScheme<T> sc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *A = reinterpret_cast<T *>(smem_raw);       // current approximation (natural order)
    T *W = A + a.cap;                             // [s ; d] work line
    const int tid = threadIdx.x;
    int nthr = blockDim.x;
    bool multi = nthr > 64;
    T *y = a.y + (int64_t)blockIdx.x * a.y_item;
    if (FW) {
        const T *src = a.src + (int64_t)blockIdx.x * a.src_item;
        int n = a.n0;
        for (int i0 = tid; i0 < n; i0 += 8 * nthr) {       // eight independent loads in flight per thread
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * nthr < n) v[u] = src[i0 + u * nthr];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * nthr < n) A[i0 + u * nthr] = v[u];
        }
        tail_sync_vm(multi);
        for (int lev = 0; lev < a.nlev; ++lev) {
            const int half = n >> 1;
            if (multi && half <= kSingleWavePairs) {      // hand the rest to wave 0 (all earlier LDS writes are synced)
                if (tid >= 64) return;
                multi = false; nthr = 64;
            }
            for (int j = tid; j < half; j += nthr) { W[j] = A[2 * j]; W[half + j] = A[2 * j + 1]; }
            tail_sync(multi);
            tail_lift_steps<T>(W, half, sc, tid, nthr, multi);
            const bool last = (lev == a.nlev - 1);
            for (int j = tid; j < half; j += nthr) {
                T s = W[j] * sc.norm1, d = W[half + j] * sc.norm2;
                y[half + j] = d;
                if (last) y[j] = s;
                else A[j] = s;
            }
            tail_sync(multi);
            n = half;
        }
    } else {
        // deepest level first: output lengths n0 >> (nlev-1), ..., n0.  Small levels come FIRST here, so
        // wave 0 starts alone and the other waves join (after one barrier) when the level is big enough.
        const T *ll = a.ll + (int64_t)blockIdx.x * a.ll_item;
        const T *src = a.src + (int64_t)blockIdx.x * a.src_item;     // detail coefficients live in src (x)
        int n = a.n0 >> (a.nlev - 1);
        const bool have_multi = multi;
        bool solo = have_multi && (n >> 1) <= kSingleWavePairs;
        if (!solo || tid < 64) {
            const int nt0 = solo ? 64 : nthr;
            for (int j = tid; j < (n >> 1); j += nt0) A[j] = ll[j];
        }
        tail_sync_vm(have_multi);
        for (int lev = 0; lev < a.nlev; ++lev) {
            const int half = n >> 1;
            solo = have_multi && half <= kSingleWavePairs;
            const bool last = (lev == a.nlev - 1);
            if (!solo || tid < 64) {
                const int nt = solo ? 64 : nthr;
                const bool m = have_multi && !solo;
                for (int j = tid; j < half; j += nt) { W[j] = sc.norm1 * A[j]; W[half + j] = sc.norm2 * src[half + j]; }
                tail_sync_vm(m);
                tail_lift_steps<T>(W, half, sc, tid, nt, m);
                for (int j = tid; j < half; j += nt) {
                    if (last) { y[2 * j] = W[j]; y[2 * j + 1] = W[half + j]; }
                    else { A[2 * j] = W[j]; A[2 * j + 1] = W[half + j]; }
                }
                tail_sync(m);
            }
            // when the NEXT level switches from solo to all waves, everybody meets here once
            const bool next_solo = have_multi && (half << 1) <= kSingleWavePairs;
            if (solo && !next_solo && !last) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            n <<= 1;
        }
    }
}


// --------------------------------------------------------------------------------------------------
// Register-resident forward tail: ALL remaining levels of a power-of-two line of <= 4096 (Float32) / 2048 (Float64)
// samples in ONE WAVE, known scheme shapes.  The LDS tail below pays a barrier per pass -- split, every lifting step,
// normalize: six per cdf9/7 level, 66 for the last eleven levels of a long line, ~0.5 us each.  Here the wave holds the whole
// line: lane L owns pairs PPL*L .. PPL*L + PPL-1, the steps run in registers exactly as in the streaming kernels, the
// out-of-lane operands come from the neighbouring lane by a ROTATING DPP shift (lane 63's neighbour is lane 0: the periodic
// boundary of the reference, transforms_lifting.jl:437-451), and the next level's pairs (s[2i], s[2i+1]) are already in the
// same lane.  Once a lane is down to one pair (64 pairs left) the line is finished with one sample per lane and
// ds_bpermute gathers (six levels, 32 .. 1 pairs).  No LDS storage, no barrier.
__device__ __forceinline__ int l_dpp_rol(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x134, 0xf, 0xf, false); }   // lane i <- lane i+1 (mod 64)
__device__ __forceinline__ int l_dpp_ror(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xf, 0xf, false); }   // lane i <- lane i-1 (mod 64)
__device__ __forceinline__ float l_rol(float v) { return __int_as_float(l_dpp_rol(__float_as_int(v))); }
__device__ __forceinline__ float l_ror(float v) { return __int_as_float(l_dpp_ror(__float_as_int(v))); }
__device__ __forceinline__ double l_rol(double v) { return __hiloint2double(l_dpp_rol(__double2hiint(v)), l_dpp_rol(__double2loint(v))); }
__device__ __forceinline__ double l_ror(double v) { return __hiloint2double(l_dpp_ror(__double2hiint(v)), l_dpp_ror(__double2loint(v))); }
__device__ __forceinline__ float l_gather(float v, int srclane) { return __int_as_float(__builtin_amdgcn_ds_bpermute(srclane << 2, __float_as_int(v))); }
__device__ __forceinline__ double l_gather(double v, int srclane)
{
    return __hiloint2double(__builtin_amdgcn_ds_bpermute(srclane << 2, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(srclane << 2, __double2loint(v)));
}

template <typename T, int PPL>
__device__ __forceinline__ T lane_operand_rot(const T (&op)[PPL], int off)
{
    const int delta = l_floordiv(off, PPL);
    const int e = off - delta * PPL;
    T v = op[e];
    if (delta == 1) v = l_rol(v);
    else if (delta == 2) v = l_rol(l_rol(v));
    else if (delta == -1) v = l_ror(v);
    else if (delta == -2) v = l_ror(l_ror(v));
    return v;
}

// all steps of one level on the whole line held by the wave (pairs PPL*lane + jj); same arithmetic as lift_steps_lane
template <typename T, int ID, int PPL>
__device__ __forceinline__ void lift_steps_wave(T (&s)[PPL], T (&d)[PPL], const T (&c)[LIFT_FAST_STEPS][WL_MAX_NCOEF], int lane)
{
    typedef Shape<ID> SH;
    const int kfirst = PPL * lane, half = 64 * PPL;
#pragma unroll
    for (int st = 0; st < SH::NS; ++st) {
        const int upd = SH::S[st].upd, nc = SH::S[st].nc, sh = SH::S[st].sh;
        T res[PPL];
#pragma unroll
        for (int jj = 0; jj < PPL; ++jj) {
            T o[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                o[kk] = (T)0;
                if (kk < nc) o[kk] = upd ? lane_operand_rot<T, PPL>(s, jj + kk - sh) : lane_operand_rot<T, PPL>(d, jj + kk - sh);
            }
            const int jg = kfirst + jj - sh;
            const bool inb = (jg >= 0) && (jg + nc - 1 <= half - 1);
            const T x = upd ? d[jj] : s[jj];
            T acc = c[st][0] * o[0];
            if (nc > 1) acc = acc + c[st][1] * o[1];
            if (nc > 2) acc = acc + c[st][2] * o[2];
            const T xin = x + acc;
            T xb = x + c[st][0] * o[0];
            if (nc > 1) xb = xb + c[st][1] * o[1];
            if (nc > 2) xb = xb + c[st][2] * o[2];
            res[jj] = inb ? xin : xb;
        }
#pragma unroll
        for (int jj = 0; jj < PPL; ++jj) { if (upd) d[jj] = res[jj]; else s[jj] = res[jj]; }
    }
}

template <typename T>
struct LiftRegArgs {
    const T *src; int64_t src_item;
    T *y; int64_t y_item;
    int n0, nlev;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

// one sample per lane: levels of <= 32 pairs
template <typename T, int ID>
__device__ __forceinline__ void lift_reg_small(T v, int m, int nlev, const LiftRegArgs<T> &a, T *y, int lane)
{
    typedef Shape<ID> SH;
    for (int lev = 0; lev < nlev; ++lev) {
        const int half = m >> 1;
        T s = l_gather(v, (2 * lane) & 63), d = l_gather(v, (2 * lane + 1) & 63);       // Util.split!
#pragma unroll
        for (int st = 0; st < SH::NS; ++st) {
            const int upd = SH::S[st].upd, nc = SH::S[st].nc, sh = SH::S[st].sh;
            const T opv = upd ? s : d;
            T o[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                o[kk] = (T)0;
                if (kk < nc) o[kk] = l_gather(opv, (lane + kk - sh) & (half - 1));
            }
            const int jg = lane - sh;
            const bool inb = (jg >= 0) && (jg + nc - 1 <= half - 1);
            const T x = upd ? d : s;
            T acc = a.c[st][0] * o[0];
            if (nc > 1) acc = acc + a.c[st][1] * o[1];
            if (nc > 2) acc = acc + a.c[st][2] * o[2];
            const T xin = x + acc;
            T xb = x + a.c[st][0] * o[0];
            if (nc > 1) xb = xb + a.c[st][1] * o[1];
            if (nc > 2) xb = xb + a.c[st][2] * o[2];
            const T r = inb ? xin : xb;
            if (upd) d = r; else s = r;
        }
        if (lane < half) y[half + lane] = d * a.norm2;
        v = s * a.norm1;
        if (lev == nlev - 1) { if (lane < half) y[lane] = v; }
        m = half;
    }
}

template <typename T, int ID, int PPL>
__device__ __forceinline__ void lift_reg_levels(T (&s)[PPL], T (&d)[PPL], int nlev, const LiftRegArgs<T> &a, T *y, int lane)
{
    lift_steps_wave<T, ID, PPL>(s, d, a.c, lane);
    constexpr int half = 64 * PPL;
    T dO[PPL], sO[PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) { dO[j] = d[j] * a.norm2; sO[j] = s[j] * a.norm1; }
    stv_l<T, PPL>(y + half + PPL * lane, dO);
    if (nlev == 1) { stv_l<T, PPL>(y + PPL * lane, sO); return; }
    if constexpr (PPL > 1) {
        T s2[PPL / 2], d2[PPL / 2];
#pragma unroll
        for (int j = 0; j < PPL / 2; ++j) { s2[j] = sO[2 * j]; d2[j] = sO[2 * j + 1]; }
        lift_reg_levels<T, ID, PPL / 2>(s2, d2, nlev - 1, a, y, lane);
    } else {
        lift_reg_small<T, ID>(sO[0], 64, nlev - 1, a, y, lane);
    }
}

template <typename T, int ID>
__global__ void __launch_bounds__(64) k_tail_lift_reg(LiftRegArgs<T> a)
{
    const int lane = threadIdx.x;
    const T *src = a.src + (int64_t)blockIdx.x * a.src_item;
    T *y = a.y + (int64_t)blockIdx.x * a.y_item;
    const int n = a.n0;
    if (n <= 64) {
        const T v = (lane < n) ? src[lane] : (T)0;
        lift_reg_small<T, ID>(v, n, a.nlev, a, y, lane);
        return;
    }
#define WL_REG_CASE(PPL_)                                                              \
    case PPL_: {                                                                       \
        T v[2 * PPL_], s[PPL_], d[PPL_];                                               \
        ldv_l<T, 2 * PPL_>(src + 2 * PPL_ * lane, v);                                  \
        _Pragma("unroll") for (int j = 0; j < PPL_; ++j) { s[j] = v[2 * j]; d[j] = v[2 * j + 1]; } \
        lift_reg_levels<T, ID, PPL_>(s, d, a.nlev, a, y, lane);                        \
    } break
    switch (n >> 7) {
        WL_REG_CASE(1);
        WL_REG_CASE(2);
        WL_REG_CASE(4);
        WL_REG_CASE(8);
        WL_REG_CASE(16);
    default:
        if constexpr (sizeof(T) == 4) {
            T v[64], s[32], d[32];
            ldv_l<T, 64>(src + 64 * lane, v);
#pragma unroll
            for (int j = 0; j < 32; ++j) { s[j] = v[2 * j]; d[j] = v[2 * j + 1]; }
            lift_reg_levels<T, ID, 32>(s, d, a.nlev, a, y, lane);
        }
        break;
    }
#undef WL_REG_CASE
}

// The inverse of the register tail: the deepest `nlev` levels of a reconstruction whose last output here is a power-of-two
// line of n0 <= 4096 / 2048 samples, one wave.  Levels run from the short end: one sample per lane (ds_bpermute operands,
// merge! = a gather) up to 64 samples, then a lane's merged pairs (s[j], d[j]) ARE the next level's approximation values
// PPL' = 2 PPL -- nothing moves between lanes except the rotating-DPP step operands; details come straight from the
// coefficient line.  normalize! -> steps -> merge! per level, as the reference (transforms_lifting.jl:57-72).
template <typename T, int ID, int PPL>
__device__ __forceinline__ void lift_reg_up(T (&s)[PPL], const LiftRegArgs<T> &a, const T *src, T *out, int lane)
{
    constexpr int half = 64 * PPL, m = 2 * half;
    T d[PPL];
    ldv_l<T, PPL>(src + half + PPL * lane, d);
#pragma unroll
    for (int j = 0; j < PPL; ++j) { s[j] = a.norm1 * s[j]; d[j] = a.norm2 * d[j]; }
    lift_steps_wave<T, ID, PPL>(s, d, a.c, lane);
    T x[2 * PPL];
#pragma unroll
    for (int j = 0; j < PPL; ++j) { x[2 * j] = s[j]; x[2 * j + 1] = d[j]; }
    if (a.n0 == m) { stv_l<T, 2 * PPL>(out + 2 * PPL * lane, x); return; }
    if constexpr (2 * PPL <= (sizeof(T) == 4 ? 32 : 16)) lift_reg_up<T, ID, 2 * PPL>(x, a, src, out, lane);
}

template <typename T, int ID>
__global__ void __launch_bounds__(64) k_tail_lift_reg_inv(LiftRegArgs<T> a)
{
    typedef Shape<ID> SH;
    const int lane = threadIdx.x;
    const T *src = a.src + (int64_t)blockIdx.x * a.src_item;       // coefficient line
    T *out = a.y + (int64_t)blockIdx.x * a.y_item;                 // reconstruction of length n0
    const int n = a.n0, mstart = n >> a.nlev;                      // deepest approximation: mstart <= 64 samples
    T v = (lane < mstart) ? src[lane] : (T)0;
    const int msmall = n < 64 ? n : 64;
    for (int m = 2 * mstart; m <= msmall; m <<= 1) {
        const int half = m >> 1;
        T sv = a.norm1 * v;
        T dv = a.norm2 * ((lane < half) ? src[half + lane] : (T)0);
#pragma unroll
        for (int st = 0; st < SH::NS; ++st) {
            const int upd = SH::S[st].upd, nc = SH::S[st].nc, sh = SH::S[st].sh;
            const T opv = upd ? sv : dv;
            T o[3];
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                o[kk] = (T)0;
                if (kk < nc) o[kk] = l_gather(opv, (lane + kk - sh) & (half - 1));
            }
            const int jg = lane - sh;
            const bool inb = (jg >= 0) && (jg + nc - 1 <= half - 1);
            const T x = upd ? dv : sv;
            T acc = a.c[st][0] * o[0];
            if (nc > 1) acc = acc + a.c[st][1] * o[1];
            if (nc > 2) acc = acc + a.c[st][2] * o[2];
            const T xin = x + acc;
            T xb = x + a.c[st][0] * o[0];
            if (nc > 1) xb = xb + a.c[st][1] * o[1];
            if (nc > 2) xb = xb + a.c[st][2] * o[2];
            const T r = inb ? xin : xb;
            if (upd) dv = r; else sv = r;
        }
        const T gs = l_gather(sv, lane >> 1), gd = l_gather(dv, lane >> 1);        // Util.merge!
        v = (lane & 1) ? gd : gs;
    }
    if (n <= 64) {
        if (lane < n) out[lane] = v;
        return;
    }
    T s1[1] = {v};
    lift_reg_up<T, ID, 1>(s1, a, src, out, lane);
}

template <typename T>
static bool lift_reg_inv_ok(int id, int64_t n0, int nlev, const T *src, int64_t src_item, const T *out, int64_t out_item)
{
    if (id != 1 && id != 3 && id != 5) return false;                       // inverse shapes
    if (n0 < 2 || (n0 & (n0 - 1)) != 0 || n0 > (sizeof(T) == 4 ? 4096 : 2048)) return false;
    if (nlev < 1 || ((int64_t)1 << nlev) > n0 || (n0 >> nlev) > 64) return false;
    constexpr int VEC = 16 / sizeof(T);
    if (n0 > 64 && ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (src_item % VEC) || (out_item % VEC))) return false;
    return true;
}

template <typename T>
static bool lift_reg_ok(int id, int64_t n, int nlev, const T *src, int64_t src_item, const T *y, int64_t y_item)
{
    if (id != 0 && id != 2 && id != 4) return false;                       // forward shapes
    if (n < 2 || (n & (n - 1)) != 0 || n > (sizeof(T) == 4 ? 4096 : 2048)) return false;
    if (nlev < 1 || ((int64_t)1 << nlev) > n) return false;
    constexpr int VEC = 16 / sizeof(T);
    // the wide loads / stores of the first level want 16-byte aligned lines (lines of < 2*VEC... samples use scalar accesses)
    if (n > 64 && ((reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) || (src_item % VEC) || (y_item % VEC))) return false;
    return true;
}

// --------------------------------------------------------------------------------------------------
// LDS tail for SQUARE 2-D blocks (n0 <= 64): every remaining level of the 2-D lifting transform inside one workgroup
// (any scheme).  Forward: per level rows (dim 2) then columns (dim 1), each as split -> steps -> normalize on all
// lines at once; inverse: the coefficient corner is staged once, per level columns then rows.  W holds the lines
// of the current pass contiguously ([s ; d] per line) so the step loops are conflict-free.
template <typename T>
struct LiftTail2DArgs {
    const T *src; int64_t lds;      // fw: block to transform; inv: coefficient array (its n0 x n0 corner)
    T *y; int64_t ldy;              // fw: coefficient array; inv: reconstruction of the shallowest level done here
    int n0;                         // fw: block size; inv: OUTPUT size of the shallowest level done here
    int nlev;
    int ld;                         // LDS leading dimension (n0 | 1)
    int cap;                        // elements per LDS buffer
};

// e -> (e / d, e % d) without the emulated integer division when d is a power of two (the usual case)
__device__ __forceinline__ void l_split_idx(int e, int d, int &q, int &r)
{
    if ((d & (d - 1)) == 0) { const int lg = 31 - __clz(d); q = e >> lg; r = e & (d - 1); }
    else { q = e / d; r = e - q * d; }
}

// all lifting steps on `nlines` lines of [s(half) ; d(half)] stored contiguously in w (line l at w + l*2*half)
template <typename T>
__device__ __forceinline__ void tail_lift_steps_lines(T *w, int half, int nlines, const LiftScheme<T> &sc, int tid, int nthr, bool multi)
{
    const int m = 2 * half;
    for (int st = 0; st < sc.nsteps; ++st) {
        const LiftStep<T> &sp = sc.step[st];
        const int nc = sp.nc, shift = sp.shift, toff = sp.is_update ? half : 0, ooff = sp.is_update ? 0 : half;
        const T c0 = sp.c[0], c1 = sp.c[1], c2 = sp.c[2];
        for (int e = tid; e < half * nlines; e += nthr) {
            int l, j;
            l_split_idx(e, half, l, j);
            T *tgt = w + l * m + toff;
            const T *op = w + l * m + ooff;
            const int j0 = j - shift;
            const bool inb = (j0 >= 0) && (j0 + nc - 1 <= half - 1);
            T x = tgt[j];
            if (inb) {
                T acc = c0 * op[j0];
                if (nc > 1) acc = acc + c1 * op[j0 + 1];
                if (nc > 2) acc = acc + c2 * op[j0 + 2];
                x = x + acc;
            } else {
                for (int k = 0; k < nc; ++k) {
                    int i = j0 + k;
                    while (i < 0) i += half;
                    while (i >= half) i -= half;
                    x = x + (k == 0 ? c0 : (k == 1 ? c1 : c2)) * op[i];
                }
            }
            tgt[j] = x;
        }
        tail_sync(multi);
    }
}

template <typename T, int FW>
__global__ void __launch_bounds__(1024) k_tail_lift2d(LiftTail2DArgs<T> a, LiftScheme<T> sc)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);       // the block (leading dimension ld)
    T *W = P + a.cap;                             // work lines
    T *Q = W + a.cap;                             // inverse only: columns-pass result
    const int tid = threadIdx.x;
    int nthr = blockDim.x;
    bool multi = nthr > 64;
    const int ld = a.ld;
    for (int e = tid; e < a.n0 * a.n0; e += nthr) {
        int j, i;
        l_split_idx(e, a.n0, j, i);
        P[i + j * ld] = a.src[i + (int64_t)j * a.lds];
    }
    tail_sync_vm(multi);
    if (FW) {
        int m = a.n0;
        for (int lev = 0; lev < a.nlev; ++lev) {
            const int h = m >> 1;
            const bool last = (lev == a.nlev - 1);
            if (multi && m * m <= 256) {              // hand the rest to wave 0
                if (tid >= 64) return;
                multi = false; nthr = 64;
            }
            // rows (dim 2): line = row i, W[i*m + p] = P[i, 2p], W[i*m + h + p] = P[i, 2p+1]
            for (int e = tid; e < m * h; e += nthr) {
                int p, i;
                l_split_idx(e, m, p, i);
                W[i * m + p] = P[i + (2 * p) * ld];
                W[i * m + h + p] = P[i + (2 * p + 1) * ld];
            }
            tail_sync(multi);
            tail_lift_steps_lines<T>(W, h, m, sc, tid, nthr, multi);
            for (int e = tid; e < m * h; e += nthr) {
                int p, i;
                l_split_idx(e, m, p, i);
                P[i + p * ld] = W[i * m + p] * sc.norm1;
                P[i + (h + p) * ld] = W[i * m + h + p] * sc.norm2;
            }
            tail_sync(multi);
            // columns (dim 1): line = column j, W[j*m + p] = P[2p, j], W[j*m + h + p] = P[2p+1, j]
            for (int e = tid; e < m * h; e += nthr) {
                int j, p;
                l_split_idx(e, h, j, p);
                W[j * m + p] = P[2 * p + j * ld];
                W[j * m + h + p] = P[2 * p + 1 + j * ld];
            }
            tail_sync(multi);
            tail_lift_steps_lines<T>(W, h, m, sc, tid, nthr, multi);
            for (int e = tid; e < m * h; e += nthr) {
                int j, p;
                l_split_idx(e, h, j, p);
                const T sv = W[j * m + p] * sc.norm1, dv = W[j * m + h + p] * sc.norm2;
                a.y[h + p + (int64_t)j * a.ldy] = dv;
                if (j < h && !last) P[p + j * ld] = sv;
                else a.y[p + (int64_t)j * a.ldy] = sv;
            }
            tail_sync(multi);
            m = h;
        }
    } else {
        int m = a.n0 >> (a.nlev - 1);
        const bool have_multi = multi;
        for (int lev = 0; lev < a.nlev; ++lev) {
            const int h = m >> 1;
            const bool last = (lev == a.nlev - 1);
            // small levels come first: wave 0 works alone until the level is big enough, the others wait at the barriers
            const bool solo = have_multi && m * m <= 256;
            const bool act = !solo || tid < 64;
            const int nt = solo ? 64 : nthr;
            const bool mm = have_multi && !solo;
            if (act) {
                // columns (dim 1) first: W[j*m + p] = norm1 * P[p, j], W[j*m + h + p] = norm2 * P[h + p, j]
                for (int e = tid; e < m * h; e += nt) {
                    int j, p;
                l_split_idx(e, h, j, p);
                    W[j * m + p] = sc.norm1 * P[p + j * ld];
                    W[j * m + h + p] = sc.norm2 * P[h + p + j * ld];
                }
                tail_sync(mm);
                tail_lift_steps_lines<T>(W, h, m, sc, tid, nt, mm);
                for (int e = tid; e < m * h; e += nt) {
                    int j, p;
                l_split_idx(e, h, j, p);
                    Q[2 * p + j * ld] = W[j * m + p];
                    Q[2 * p + 1 + j * ld] = W[j * m + h + p];
                }
                tail_sync(mm);
                // rows (dim 2): W[i*m + p] = norm1 * Q[i, p], W[i*m + h + p] = norm2 * Q[i, h + p]
                for (int e = tid; e < m * h; e += nt) {
                    int p, i;
                l_split_idx(e, m, p, i);
                    W[i * m + p] = sc.norm1 * Q[i + p * ld];
                    W[i * m + h + p] = sc.norm2 * Q[i + (h + p) * ld];
                }
                tail_sync(mm);
                tail_lift_steps_lines<T>(W, h, m, sc, tid, nt, mm);
                for (int e = tid; e < m * h; e += nt) {
                    int p, i;
                l_split_idx(e, m, p, i);
                    const T v0 = W[i * m + p], v1 = W[i * m + h + p];
                    if (last) { a.y[i + (int64_t)(2 * p) * a.ldy] = v0; a.y[i + (int64_t)(2 * p + 1) * a.ldy] = v1; }
                    else { P[i + (2 * p) * ld] = v0; P[i + (2 * p + 1) * ld] = v1; }
                }
                tail_sync(mm);
            }
            const bool next_solo = have_multi && (4 * m * m) <= 256;
            if (solo && !next_solo && !last) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            m <<= 1;
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Register tail for 2-D lifting: every remaining level of a power-of-two block of <= 64 x 64 in ONE WAVE.  A lane owns a
// whole line (a row in the dim-2 pass, a column in the dim-1 pass) and keeps it in registers as s[H], d[H]; every step of
// the scheme is then straight-line code with compile-time indices -- the periodic wrap and the reference's two summation
// forms (in-bounds: x + (c1 a + c2 b); boundary: (x + c1 a) + c2 b, transforms_lifting.jl:366-483) are decided per element
// at compile time.  The block lives in LDS between the passes (leading dimension M | 1: conflict-free rows and columns);
// a single wave needs no barrier, only the LDS wait.  64 x 64 cdf9/7, 6 levels: ~10 us (LDS workgroup tail: 53 us).
template <typename T>
struct LiftTailRegArgs {
    const T *src; int64_t lds;
    T *y; int64_t ldy;
    int m0, nlev;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <typename T, int ID, int H>
__device__ __forceinline__ void reg_line_steps(T (&s)[H], T (&d)[H], const T (&c)[LIFT_FAST_STEPS][WL_MAX_NCOEF])
{
    typedef Shape<ID> SH;
#pragma unroll
    for (int k = 0; k < SH::NS; ++k) {
        const int upd = SH::S[k].upd, nc = SH::S[k].nc, sh = SH::S[k].sh;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int j0 = j - sh;
            const bool inb = (j0 >= 0) && (j0 + nc - 1 <= H - 1);
            T x = upd ? d[j] : s[j];
            if (inb) {
                T acc = c[k][0] * (upd ? s[((j0 % H) + H) % H] : d[((j0 % H) + H) % H]);
                if (nc > 1) acc = acc + c[k][1] * (upd ? s[(j0 + 1) % H] : d[(j0 + 1) % H]);
                if (nc > 2) acc = acc + c[k][2] * (upd ? s[(j0 + 2) % H] : d[(j0 + 2) % H]);
                x = x + acc;
            } else {
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    if (kk < nc) {
                        const int i = (((j0 + kk) % H) + H) % H;
                        x = x + c[k][kk] * (upd ? s[i] : d[i]);
                    }
            }
            if (upd) d[j] = x; else s[j] = x;
        }
    }
}
__device__ __forceinline__ void reg_tail_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// one forward level of the M x M block in P (split -> steps -> normalize per line; rows pass, then columns pass), details
// (and the approximation, when this is the last level) copied out with coalesced stores
template <typename T, int ID, int M>
__device__ __forceinline__ void reg_tail_fwd_level(T *P, const LiftTailRegArgs<T> &a, bool last, int lane)
{
    constexpr int H = M / 2, ld = 64 | 1;
    if (lane < M) {
        T s[H], d[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = P[lane + (2 * k) * ld]; d[k] = P[lane + (2 * k + 1) * ld]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { P[lane + k * ld] = s[k] * a.norm1; P[lane + (H + k) * ld] = d[k] * a.norm2; }
    }
    reg_tail_sync();
    if (lane < M) {
        T s[H], d[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = P[2 * k + lane * ld]; d[k] = P[2 * k + 1 + lane * ld]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { P[k + lane * ld] = s[k] * a.norm1; P[H + k + lane * ld] = d[k] * a.norm2; }
    }
    reg_tail_sync();
#pragma unroll
    for (int t = 0; t < (M * M + 63) / 64; ++t) {
        const int idx = lane + 64 * t, i = idx % M, j = idx / M;
        if (idx < M * M && (last || i >= H || j >= H)) a.y[i + (int64_t)j * a.ldy] = P[i + j * ld];
    }
}
template <typename T, int ID, int M>
__device__ __forceinline__ void reg_tail_fwd_from(T *P, const LiftTailRegArgs<T> &a, int m0, int nlev, int lane)
{
    if (m0 == M) {
        reg_tail_fwd_level<T, ID, M>(P, a, nlev == 1, lane);
        if constexpr (M >= 4) {
            if (nlev > 1) reg_tail_fwd_from<T, ID, M / 2>(P, a, M / 2, nlev - 1, lane);
        }
    } else {
        if constexpr (M >= 4) reg_tail_fwd_from<T, ID, M / 2>(P, a, m0, nlev, lane);
    }
}
// one inverse level with output M x M in P (normalize -> steps -> merge per line; columns pass, then rows pass)
template <typename T, int ID, int M>
__device__ __forceinline__ void reg_tail_inv_level(T *P, const LiftTailRegArgs<T> &a, int lane)
{
    constexpr int H = M / 2, ld = 64 | 1;
    if (lane < M) {
        T s[H], d[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * P[k + lane * ld]; d[k] = a.norm2 * P[H + k + lane * ld]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { P[2 * k + lane * ld] = s[k]; P[2 * k + 1 + lane * ld] = d[k]; }
    }
    reg_tail_sync();
    if (lane < M) {
        T s[H], d[H];
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * P[lane + k * ld]; d[k] = a.norm2 * P[lane + (H + k) * ld]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { P[lane + (2 * k) * ld] = s[k]; P[lane + (2 * k + 1) * ld] = d[k]; }
    }
    reg_tail_sync();
}
// levels with outputs m0 >> (nlev-1), ..., m0 (smallest first)
template <typename T, int ID, int M>
__device__ __forceinline__ void reg_tail_inv_upto(T *P, const LiftTailRegArgs<T> &a, int m0, int nlev, int lane)
{
    // sizes handled by this instance: M, if m0 >> (nlev-1) <= M <= m0
    if constexpr (M >= 4) reg_tail_inv_upto<T, ID, M / 2>(P, a, m0, nlev, lane);
    if (M <= m0 && M >= (m0 >> (nlev - 1))) reg_tail_inv_level<T, ID, M>(P, a, lane);
}

template <typename T, int ID, int FW>
__global__ void __launch_bounds__(64) k_tail_lift2d_reg(LiftTailRegArgs<T> a)
{
    constexpr int ld = 64 | 1;
    __shared__ T P[ld * 64];
    const int lane = threadIdx.x;
    const int m0 = a.m0, lg = 31 - __clz(m0);
    for (int idx = lane; idx < m0 * m0; idx += 64) {
        const int i = idx & (m0 - 1), j = idx >> lg;
        P[i + j * ld] = a.src[i + (int64_t)j * a.lds];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (FW) {
        reg_tail_fwd_from<T, ID, 64>(P, a, m0, a.nlev, lane);
    } else {
        reg_tail_inv_upto<T, ID, 64>(P, a, m0, a.nlev, lane);
        for (int idx = lane; idx < m0 * m0; idx += 64) {
            const int i = idx & (m0 - 1), j = idx >> lg;
            a.y[i + (int64_t)j * a.ldy] = P[i + j * ld];
        }
    }
}
template <typename T>
static bool tail_lift2d_reg_ok(int id, int n0) { return id >= 0 && id <= 5 && n0 >= 2 && n0 <= 64 && (n0 & (n0 - 1)) == 0; }
template <typename T, int FW>
static hipError_t launch_tail_lift2d_reg(int id, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy,
                                         int n0, int nlev)
{
    LiftTailRegArgs<T> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.m0 = n0; a.nlev = nlev;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    if (FW) {
        if (id == 0) hipLaunchKernelGGL((k_tail_lift2d_reg<T, 0, 1>), dim3(1), dim3(64), 0, st, a);
        else if (id == 2) hipLaunchKernelGGL((k_tail_lift2d_reg<T, 2, 1>), dim3(1), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_tail_lift2d_reg<T, 4, 1>), dim3(1), dim3(64), 0, st, a);
    } else {
        if (id == 1) hipLaunchKernelGGL((k_tail_lift2d_reg<T, 1, 0>), dim3(1), dim3(64), 0, st, a);
        else if (id == 3) hipLaunchKernelGGL((k_tail_lift2d_reg<T, 3, 0>), dim3(1), dim3(64), 0, st, a);
        else hipLaunchKernelGGL((k_tail_lift2d_reg<T, 5, 0>), dim3(1), dim3(64), 0, st, a);
    }
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// LDS tail for 2-D lifting, a thread per line, several waves (round 4): every remaining level of a power-of-two block of
// <= 128 x 128 (Float32; 64 x 64 Float64) in one workgroup.  Same scheme as the register tail above -- a thread holds a whole
// row / column in registers, straight-line steps with compile-time wrap and summation forms -- but the block is staged, and
// its details stored, by 256 threads, and a 128-sample line is two waves' worth of threads: the 128 x 128 level no longer
// needs a tile launch of its own (8192^2 cdf9/7: 128^2 tile 9.7 us + 64^2 register tail 13.5 us -> one launch).
__device__ __forceinline__ void tail2l_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail2l_fwd_level(T *P, const LiftTailRegArgs<T> &a, bool last, int tid, int nthr)
{
    constexpr int H = M / 2, LD = MM + 1;
    if (tid < M) {                                   // rows (dim 2): the line of row tid runs along the columns
        T s[H], d[H];
        T *q = P + tid;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = q[(2 * k) * LD]; d[k] = q[(2 * k + 1) * LD]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[k * LD] = s[k] * a.norm1; q[(H + k) * LD] = d[k] * a.norm2; }
    }
    tail2l_barrier();
    if (tid < M) {                                   // columns (dim 1)
        T s[H], d[H];
        T *q = P + tid * LD;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = q[2 * k]; d[k] = q[2 * k + 1]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[k] = s[k] * a.norm1; q[H + k] = d[k] * a.norm2; }
    }
    tail2l_barrier();
    for (int idx = tid; idx < M * M; idx += nthr) {
        const int i = idx % M, j = idx / M;
        if (last || i >= H || j >= H) a.y[i + (int64_t)j * a.ldy] = P[i + j * LD];
    }
}
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail2l_fwd_from(T *P, const LiftTailRegArgs<T> &a, int m0, int nlev, int tid, int nthr)
{
    if (m0 == M) {
        tail2l_fwd_level<T, ID, M, MM>(P, a, nlev == 1, tid, nthr);
        if constexpr (M >= 4) {
            if (nlev > 1) tail2l_fwd_from<T, ID, M / 2, MM>(P, a, M / 2, nlev - 1, tid, nthr);
        }
    } else {
        if constexpr (M >= 4) tail2l_fwd_from<T, ID, M / 2, MM>(P, a, m0, nlev, tid, nthr);
    }
}
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail2l_inv_level(T *P, const LiftTailRegArgs<T> &a, int tid)
{
    constexpr int H = M / 2, LD = MM + 1;
    if (tid < M) {                                   // columns (dim 1) first
        T s[H], d[H];
        T *q = P + tid * LD;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * q[k]; d[k] = a.norm2 * q[H + k]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[2 * k] = s[k]; q[2 * k + 1] = d[k]; }
    }
    tail2l_barrier();
    if (tid < M) {                                   // rows (dim 2)
        T s[H], d[H];
        T *q = P + tid;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * q[k * LD]; d[k] = a.norm2 * q[(H + k) * LD]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[(2 * k) * LD] = s[k]; q[(2 * k + 1) * LD] = d[k]; }
    }
    tail2l_barrier();
}
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail2l_inv_upto(T *P, const LiftTailRegArgs<T> &a, int m0, int nlev, int tid)
{
    if constexpr (M >= 4) tail2l_inv_upto<T, ID, M / 2, MM>(P, a, m0, nlev, tid);
    if (M <= m0 && M >= (m0 >> (nlev - 1))) tail2l_inv_level<T, ID, M, MM>(P, a, tid);
}
template <typename T, int ID, int FW, int MM>
__global__ void __launch_bounds__(256) k_tail_lift2d_lds(LiftTailRegArgs<T> a)
{
    constexpr int LD = MM + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m0 = a.m0, lg = 31 - __clz(m0);
    for (int idx = tid; idx < m0 * m0; idx += nthr) {
        const int i = idx & (m0 - 1), j = idx >> lg;
        P[i + j * LD] = a.src[i + (int64_t)j * a.lds];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (FW) {
        tail2l_fwd_from<T, ID, MM, MM>(P, a, m0, a.nlev, tid, nthr);
    } else {
        tail2l_inv_upto<T, ID, MM, MM>(P, a, m0, a.nlev, tid);
        for (int idx = tid; idx < m0 * m0; idx += nthr) {
            const int i = idx & (m0 - 1), j = idx >> lg;
            a.y[i + (int64_t)j * a.ldy] = P[i + j * LD];
        }
    }
}
template <typename T> constexpr int tail2l_max() { return sizeof(T) == 4 ? 128 : 64; }
template <typename T>
static bool tail_lift2d_lds_ok(int id, int64_t n) { return id >= 0 && id <= 5 && n >= 2 && n <= tail2l_max<T>() && (n & (n - 1)) == 0; }
template <typename T, int ID, int FW>
static hipError_t launch_tail_lift2d_lds_id(hipStream_t st, const LiftTailRegArgs<T> &a)
{
    constexpr int MM = tail2l_max<T>();
    const size_t shmem = (size_t)(MM + 1) * MM * sizeof(T);
    static thread_local int attr_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool done = false;
    for (int i = 0; i < 8; ++i) done = done || attr_dev[i] == dev;
    if (!done && shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail_lift2d_lds<T, ID, FW, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        for (int i = 0; i < 8; ++i) if (attr_dev[i] < 0) { attr_dev[i] = dev; break; }
    }
    hipLaunchKernelGGL((k_tail_lift2d_lds<T, ID, FW, MM>), dim3(1), dim3(256), shmem, st, a);
    return hipGetLastError();
}
template <typename T, int FW>
static hipError_t launch_tail_lift2d_lds(int id, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy,
                                         int n0, int nlev)
{
    LiftTailRegArgs<T> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.m0 = n0; a.nlev = nlev;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    if (FW) {
        if (id == 0) return launch_tail_lift2d_lds_id<T, 0, 1>(st, a);
        if (id == 2) return launch_tail_lift2d_lds_id<T, 2, 1>(st, a);
        return launch_tail_lift2d_lds_id<T, 4, 1>(st, a);
    }
    if (id == 1) return launch_tail_lift2d_lds_id<T, 1, 0>(st, a);
    if (id == 3) return launch_tail_lift2d_lds_id<T, 3, 0>(st, a);
    return launch_tail_lift2d_lds_id<T, 5, 0>(st, a);
}

// --------------------------------------------------------------------------------------------------
// LDS tail for 3-D lifting (round 4): every remaining level of a power-of-two cube of <= MM^3 (32^3 Float32 = 132 KiB of LDS,
// 16^3 Float64) in ONE workgroup.  As in the 2-D register tail a thread owns a whole line of the current pass and keeps it in
// registers (s[H], d[H]: compile-time indices, the periodic wrap and the reference's two summation forms decided per element
// at compile time); the cube lives in LDS between the passes (leading dimensions MM + 1 and (MM + 1) MM: the three line
// directions all read / write conflict-free), M^2 lines per pass = one line per thread of the 1024.  Pass order of the
// reference (transforms_lifting.jl:228-268): forward planes (dim 3), rows (dim 2), columns (dim 1); inverse columns, rows, planes.
// Replaces three launches per level (k_lift_axis_stream x 2 + k_lift_short_lines) on the levels that hold 0.2 % of the data:
// 256^3 cdf9/7, 8 levels: 22 launches -> 8.
__device__ __forceinline__ void tail3_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename T>
struct LiftTail3Args {
    const T *src; int64_t s1, s2;   // fw: the level's input cube;  inv: the coefficient array (its m0^3 corner)
    T *y; int64_t y1, y2;           // fw: the coefficient array;   inv: the m0^3 result
    int m0, nlev;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail3_fwd_level(T *P, const LiftTail3Args<T> &a, bool last, int tid, int nthr)
{
    constexpr int H = M / 2, L1 = MM + 1, L2 = (MM + 1) * MM;
    const int u = tid % M, v = tid / M;                      // the two coordinates a line does not run along
    const bool act = tid < M * M;
    // ---- planes: lines along dim 3 at (i, j) = (u, v) ----
    if (act) {
        T s[H], d[H];
        T *q = P + u + v * L1;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = q[(2 * k) * L2]; d[k] = q[(2 * k + 1) * L2]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[k * L2] = s[k] * a.norm1; q[(H + k) * L2] = d[k] * a.norm2; }
    }
    tail3_barrier();
    // ---- rows: lines along dim 2 at (i, k) = (u, v) ----
    if (act) {
        T s[H], d[H];
        T *q = P + u + v * L2;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = q[(2 * k) * L1]; d[k] = q[(2 * k + 1) * L1]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[k * L1] = s[k] * a.norm1; q[(H + k) * L1] = d[k] * a.norm2; }
    }
    tail3_barrier();
    // ---- columns: lines along dim 1 at (j, k) = (u, v) ----
    if (act) {
        T s[H], d[H];
        T *q = P + u * L1 + v * L2;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = q[2 * k]; d[k] = q[2 * k + 1]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[k] = s[k] * a.norm1; q[H + k] = d[k] * a.norm2; }
    }
    tail3_barrier();
    // the seven detail octants are final (the approximation octant too after the last level): lanes along dim 1
    for (int idx = tid; idx < M * M * M; idx += nthr) {
        const int i = idx % M, j = (idx / M) % M, k = idx / (M * M);
        if (last || i >= H || j >= H || k >= H) a.y[i + (int64_t)j * a.y1 + (int64_t)k * a.y2] = P[i + j * L1 + k * L2];
    }
}
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail3_fwd_from(T *P, const LiftTail3Args<T> &a, int m0, int nlev, int tid, int nthr)
{
    if (m0 == M) {
        tail3_fwd_level<T, ID, M, MM>(P, a, nlev == 1, tid, nthr);
        if constexpr (M >= 4) {
            if (nlev > 1) tail3_fwd_from<T, ID, M / 2, MM>(P, a, M / 2, nlev - 1, tid, nthr);
        }
    } else {
        if constexpr (M >= 4) tail3_fwd_from<T, ID, M / 2, MM>(P, a, m0, nlev, tid, nthr);
    }
}
// one inverse level with output M^3 in P (normalize -> steps -> merge per line; columns, rows, planes)
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail3_inv_level(T *P, const LiftTail3Args<T> &a, int tid)
{
    constexpr int H = M / 2, L1 = MM + 1, L2 = (MM + 1) * MM;
    const int u = tid % M, v = tid / M;
    const bool act = tid < M * M;
    if (act) {
        T s[H], d[H];
        T *q = P + u * L1 + v * L2;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * q[k]; d[k] = a.norm2 * q[H + k]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[2 * k] = s[k]; q[2 * k + 1] = d[k]; }
    }
    tail3_barrier();
    if (act) {
        T s[H], d[H];
        T *q = P + u + v * L2;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * q[k * L1]; d[k] = a.norm2 * q[(H + k) * L1]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[(2 * k) * L1] = s[k]; q[(2 * k + 1) * L1] = d[k]; }
    }
    tail3_barrier();
    if (act) {
        T s[H], d[H];
        T *q = P + u + v * L1;
#pragma unroll
        for (int k = 0; k < H; ++k) { s[k] = a.norm1 * q[k * L2]; d[k] = a.norm2 * q[(H + k) * L2]; }
        reg_line_steps<T, ID, H>(s, d, a.c);
#pragma unroll
        for (int k = 0; k < H; ++k) { q[(2 * k) * L2] = s[k]; q[(2 * k + 1) * L2] = d[k]; }
    }
    tail3_barrier();
}
// levels with outputs m0 >> (nlev-1), ..., m0 (smallest first)
template <typename T, int ID, int M, int MM>
__device__ __forceinline__ void tail3_inv_upto(T *P, const LiftTail3Args<T> &a, int m0, int nlev, int tid)
{
    if constexpr (M >= 4) tail3_inv_upto<T, ID, M / 2, MM>(P, a, m0, nlev, tid);
    if (M <= m0 && M >= (m0 >> (nlev - 1))) tail3_inv_level<T, ID, M, MM>(P, a, tid);
}

template <typename T, int ID, int FW, int MM>
__global__ void __launch_bounds__(1024) k_tail_lift3d(LiftTail3Args<T> a)
{
    constexpr int L1 = MM + 1, L2 = (MM + 1) * MM;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *P = reinterpret_cast<T *>(smem_raw);
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m0 = a.m0, lg = 31 - __clz(m0);
    for (int idx = tid; idx < m0 * m0 * m0; idx += nthr) {
        const int i = idx & (m0 - 1), j = (idx >> lg) & (m0 - 1), k = idx >> (2 * lg);
        P[i + j * L1 + k * L2] = a.src[i + (int64_t)j * a.s1 + (int64_t)k * a.s2];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (FW) {
        tail3_fwd_from<T, ID, MM, MM>(P, a, m0, a.nlev, tid, nthr);
    } else {
        tail3_inv_upto<T, ID, MM, MM>(P, a, m0, a.nlev, tid);
        for (int idx = tid; idx < m0 * m0 * m0; idx += nthr) {
            const int i = idx & (m0 - 1), j = (idx >> lg) & (m0 - 1), k = idx >> (2 * lg);
            a.y[i + (int64_t)j * a.y1 + (int64_t)k * a.y2] = P[i + j * L1 + k * L2];
        }
    }
}
template <typename T> constexpr int tail3_max() { return sizeof(T) == 4 ? 32 : 16; }
template <typename T>
static bool tail_lift3d_ok(int id, int64_t n) { return id >= 0 && id <= 5 && n >= 2 && n <= tail3_max<T>() && (n & (n - 1)) == 0; }
template <typename T, int ID, int FW>
static hipError_t launch_tail_lift3d_id(hipStream_t st, const LiftTail3Args<T> &a)
{
    constexpr int MM = tail3_max<T>();
    const size_t shmem = (size_t)(MM + 1) * MM * MM * sizeof(T);
    static thread_local int attr_dev[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int dev = 0;
    (void)hipGetDevice(&dev);
    bool done = false;
    for (int i = 0; i < 8; ++i) done = done || attr_dev[i] == dev;
    if (!done && shmem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail_lift3d<T, ID, FW, MM>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        for (int i = 0; i < 8; ++i) if (attr_dev[i] < 0) { attr_dev[i] = dev; break; }
    }
    const int lines = a.m0 * a.m0;
    const int threads = lines >= 1024 ? 1024 : (lines >= 256 ? 256 : 64);
    hipLaunchKernelGGL((k_tail_lift3d<T, ID, FW, MM>), dim3(1), dim3(threads), shmem, st, a);
    return hipGetLastError();
}
template <typename T, int FW>
static hipError_t launch_tail_lift3d(int id, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t s1, int64_t s2, T *y, int64_t y1,
                                     int64_t y2, int m0, int nlev)
{
    LiftTail3Args<T> a;
    a.src = src; a.s1 = s1; a.s2 = s2; a.y = y; a.y1 = y1; a.y2 = y2; a.m0 = m0; a.nlev = nlev;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    if (FW) {
        if (id == 0) return launch_tail_lift3d_id<T, 0, 1>(st, a);
        if (id == 2) return launch_tail_lift3d_id<T, 2, 1>(st, a);
        return launch_tail_lift3d_id<T, 4, 1>(st, a);
    }
    if (id == 1) return launch_tail_lift3d_id<T, 1, 0>(st, a);
    if (id == 3) return launch_tail_lift3d_id<T, 3, 0>(st, a);
    return launch_tail_lift3d_id<T, 5, 0>(st, a);
}

// --------------------------------------------------------------------------------------------------
// One 2-D lifting level of a block of ANY even size in one launch (known scheme shapes): the levels the streaming / register
// kernels decline (sizes that are not multiples of 8, small non-power-of-two blocks: 1000 x 1000 and its 500, 250 levels)
// ran as twelve one-thread-per-element launches per level.  One WAVE per tile: 64 x 64 samples of the block incl. a halo of
// HP pairs on every side (the dependency cone of the scheme), staged to LDS with periodic wrap; exactly as in the register
// tail a lane holds a whole tile row (then a whole tile column) in registers and runs split -> steps -> normalize (or the
// inverse order) as straight-line code -- only the choice between the reference's in-bounds and boundary summation forms
// depends on the global position and is a wave-uniform select.  Tile edges inside the halo are garbage and never stored.
template <typename T>
struct LiftGTileArgs {
    const T *src; int64_t lds;      // fw: block n x n;  inv: coefficient array
    T *y; int64_t ldy;              // fw: coefficient array;  inv: result block
    T *ll; int64_t ldl;             // fw: approximation destination or nullptr (-> y);  inv: approximation source or nullptr (-> src)
    int n;                          // block size (even)
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

// all steps on an open line of 32 pairs whose first pair has the global (periodic) index kg0 of a line with `half` pairs
template <typename T, int ID>
__device__ __forceinline__ void tile_line_steps(T (&s)[32], T (&d)[32], const T (&c)[LIFT_FAST_STEPS][WL_MAX_NCOEF], int kg0, int half)
{
    typedef Shape<ID> SH;
#pragma unroll
    for (int k = 0; k < SH::NS; ++k) {
        const int upd = SH::S[k].upd, nc = SH::S[k].nc, sh = SH::S[k].sh;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int j0 = j - sh;
            if (j0 < 0 || j0 + nc - 1 > 31) continue;              // operands outside the tile: this pair is halo by then
            int jg = kg0 + j;                                       // wave-uniform
            while (jg >= half) jg -= half;
            const bool inb = (jg - sh >= 0) && (jg - sh + nc - 1 <= half - 1);
            const T x = upd ? d[j] : s[j];
            const T o0 = upd ? s[j0] : d[j0];
            const T o1 = (nc > 1) ? (upd ? s[(j0 + 1) & 31] : d[(j0 + 1) & 31]) : (T)0;
            const T o2 = (nc > 2) ? (upd ? s[(j0 + 2) & 31] : d[(j0 + 2) & 31]) : (T)0;
            T acc = c[k][0] * o0;
            if (nc > 1) acc = acc + c[k][1] * o1;
            if (nc > 2) acc = acc + c[k][2] * o2;
            const T xin = x + acc;
            T xb = x + c[k][0] * o0;
            if (nc > 1) xb = xb + c[k][1] * o1;
            if (nc > 2) xb = xb + c[k][2] * o2;
            const T r = inb ? xin : xb;
            if (upd) d[j] = r; else s[j] = r;
        }
    }
}

template <typename T, int ID, int FW>
__global__ void __launch_bounds__(64) k_lift2d_gtile(LiftGTileArgs<T> a)
{
    constexpr int HP = LiftReach<ID>::HP, OWN = 32 - 2 * HP, ld = 65;
    static_assert(OWN >= 8, "scheme reach too large for the tile");
    __shared__ T P[ld * 64];
    const int lane = threadIdx.x;
    const int n = a.n, h = n >> 1;
    int pr0 = (int)blockIdx.x * OWN - HP, pc0 = (int)blockIdx.y * OWN - HP;       // first pair of the tile per dimension (periodic)
    while (pr0 < 0) pr0 += h;
    while (pc0 < 0) pc0 += h;
    const int own_r0 = (int)blockIdx.x * OWN, own_c0 = (int)blockIdx.y * OWN;      // first OWNED pair
    auto wrapp = [&](int i) __attribute__((always_inline)) { while (i >= h) i -= h; return i; };
    if (FW) {
        // stage: P[i + j*ld] = x[(2 pr0 + i) mod n, (2 pc0 + j) mod n]; lanes along the rows
        {
            int gi = 2 * pr0 + lane;
            while (gi >= n) gi -= n;
            int gj = 2 * pc0;
            for (int j = 0; j < 64; ++j) {
                P[lane + j * ld] = a.src[gi + (int64_t)gj * a.lds];
                if (++gj >= n) gj -= n;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        {   // dim 2: lane = tile row, line along the columns
            T s[32], d[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) { s[k] = P[lane + (2 * k) * ld]; d[k] = P[lane + (2 * k + 1) * ld]; }
            tile_line_steps<T, ID>(s, d, a.c, pc0, h);
#pragma unroll
            for (int k = 0; k < 32; ++k) { P[lane + k * ld] = s[k] * a.norm1; P[lane + (32 + k) * ld] = d[k] * a.norm2; }
        }
        reg_tail_sync();
        {   // dim 1: lane = tile column (32 s-columns, 32 d-columns), line along the rows
            T s[32], d[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) { s[k] = P[2 * k + lane * ld]; d[k] = P[2 * k + 1 + lane * ld]; }
            tile_line_steps<T, ID>(s, d, a.c, pr0, h);
#pragma unroll
            for (int k = 0; k < 32; ++k) { P[k + lane * ld] = s[k] * a.norm1; P[32 + k + lane * ld] = d[k] * a.norm2; }
        }
        reg_tail_sync();
        // store the owned pairs: lanes along the rows (s rows then d rows), one tile column per iteration
        T *const lld = a.ll ? a.ll : a.y;
        const int64_t ldl = a.ll ? a.ldl : a.ldy;
        const int kr = lane & 31, rdet = lane >> 5;                 // local row pair, 0: s row, 1: d row
        const int gk = own_r0 + (kr - HP);
        const bool row_ok = kr >= HP && kr < 32 - HP && gk < h;
        for (int c = 0; c < 64; ++c) {
            const int kc = c & 31, cdet = c >> 5;
            const int gc = own_c0 + (kc - HP);
            if (kc >= HP && kc < 32 - HP && gc < h && row_ok) {
                const T v = P[lane + c * ld];
                if (!rdet && !cdet) lld[gk + (int64_t)gc * ldl] = v;
                else a.y[(rdet ? h : 0) + gk + (int64_t)((cdet ? h : 0) + gc) * a.ldy] = v;
            }
        }
    } else {
        // stage the four quadrant pieces: P[i + j*ld], i < 32: s row pr0 + i, i >= 32: d row pr0 + i - 32 (columns alike)
        {
            const int kr = lane & 31, rdet = lane >> 5;
            const int gr = wrapp(pr0 + kr);
            const T *lls = a.ll ? a.ll : a.src;
            const int64_t ldl = a.ll ? a.ldl : a.lds;
            int gc = pc0;
            for (int j = 0; j < 32; ++j) {
                P[lane + j * ld] = rdet ? a.src[h + gr + (int64_t)gc * a.lds] : lls[gr + (int64_t)gc * ldl];
                P[lane + (32 + j) * ld] = a.src[(rdet ? h : 0) + gr + (int64_t)(h + gc) * a.lds];
                if (++gc >= h) gc -= h;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        {   // dim 1 first: lane = tile column, line along the rows: normalize -> steps -> merge
            T s[32], d[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) { s[k] = a.norm1 * P[k + lane * ld]; d[k] = a.norm2 * P[32 + k + lane * ld]; }
            tile_line_steps<T, ID>(s, d, a.c, pr0, h);
#pragma unroll
            for (int k = 0; k < 32; ++k) { P[2 * k + lane * ld] = s[k]; P[2 * k + 1 + lane * ld] = d[k]; }
        }
        reg_tail_sync();
        {   // dim 2: lane = tile row, line along the columns
            T s[32], d[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) { s[k] = a.norm1 * P[lane + k * ld]; d[k] = a.norm2 * P[lane + (32 + k) * ld]; }
            tile_line_steps<T, ID>(s, d, a.c, pc0, h);
#pragma unroll
            for (int k = 0; k < 32; ++k) { P[lane + (2 * k) * ld] = s[k]; P[lane + (2 * k + 1) * ld] = d[k]; }
        }
        reg_tail_sync();
        // sample (i, j) of the tile is x[2 (own_r0 - HP) + i, 2 (own_c0 - HP) + j]; owned: pairs HP .. 32-HP-1
        const int i = lane, kr = i >> 1;
        const int gi = 2 * own_r0 + (i - 2 * HP);
        const bool row_ok = kr >= HP && kr < 32 - HP && gi < n;
        for (int j = 0; j < 64; ++j) {
            const int kc = j >> 1;
            const int gj = 2 * own_c0 + (j - 2 * HP);
            if (kc >= HP && kc < 32 - HP && gj < n && row_ok) a.y[gi + (int64_t)gj * a.ldy] = P[i + j * ld];
        }
    }
}

template <typename T>
static bool lift2d_gtile_ok(int id, int64_t n) { return id >= 0 && id <= 5 && n >= 2 && (n % 2) == 0 && n < ((int64_t)1 << 30); }
template <typename T, int FW>
static hipError_t launch_lift2d_gtile(int id, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy, T *ll,
                                      int64_t ldl, int64_t n)
{
    LiftGTileArgs<T> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.ll = ll; a.ldl = ldl; a.n = (int)n;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
    const int64_t h = n >> 1;
#define WL_LGT(ID_)                                                                                             \
    {                                                                                                           \
        constexpr int OWN = 32 - 2 * LiftReach<ID_>::HP;                                                        \
        const unsigned g = (unsigned)((h + OWN - 1) / OWN);                                                     \
        if (g > 65535) return hipErrorInvalidValue;                                                             \
        hipLaunchKernelGGL((k_lift2d_gtile<T, ID_, FW>), dim3(g, g), dim3(64), 0, st, a);                       \
    }
    if (FW) {
        if (id == 0) WL_LGT(0) else if (id == 2) WL_LGT(2) else WL_LGT(4)
    } else {
        if (id == 1) WL_LGT(1) else if (id == 3) WL_LGT(3) else WL_LGT(5)
    }
#undef WL_LGT
    return hipGetLastError();
}

template <typename T, int FW>
static hipError_t launch_tail_lift2d(hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy, int n0, int nlev)
{
    LiftTail2DArgs<T> a;
    a.src = src; a.lds = lds; a.y = y; a.ldy = ldy; a.n0 = n0; a.nlev = nlev;
    a.ld = n0 | 1;
    a.cap = (a.ld * n0 + 15) & ~15;
    const size_t shmem = (size_t)(FW ? 2 : 3) * a.cap * sizeof(T);
    static unsigned char attr_set[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tail_lift2d<T, FW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = 1;
    }
    const int work = n0 * n0;
    int threads = work >= 2048 ? 1024 : (work >= 512 ? 256 : 64);
    if (opt("WL_LIFT_TAIL_THREADS", 0) >= 64) threads = (int)opt("WL_LIFT_TAIL_THREADS", 0);
    hipLaunchKernelGGL((k_tail_lift2d<T, FW>), dim3(1), dim3(threads), shmem, st, a, sc);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
static inline int l_env(const char *name, int dflt) { return (int)opt(name, dflt); }   // per-context options

template <typename T>
constexpr int lift_tail_cap() { return sizeof(T) == 4 ? 16384 : 8192; }

template <typename T, int ID, int FW>
static void launch_stream_id(hipStream_t st, const Lift1DArgs<T> &a, int64_t nlines, int cu_count)
{
    int64_t gx = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 8;
    if (gx > cap) gx = cap;
    const int64_t slab = (l_env("WL_SLAB_LINES", 32768) > 0) ? l_env("WL_SLAB_LINES", 32768) : 32768;
    for (int64_t l0 = 0; l0 < nlines; l0 += slab) {      // gridDim.y <= 65535
        const int64_t nl = (nlines - l0 < slab) ? (nlines - l0) : slab;
        Lift1DArgs<T> b = a;
        b.a = a.a + l0 * a.a_ls; b.b = a.b ? a.b + l0 * a.b_ls : nullptr;
        b.o0 = a.o0 + l0 * a.o0_ls; b.o1 = a.o1 ? a.o1 + l0 * a.o1_ls : nullptr;
        hipLaunchKernelGGL((k_lift1d_stream<T, ID, FW>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
    }
}

template <typename T, int ID>
static void launch_fwd3_id(hipStream_t st, const Lift3Args<T> &a, int64_t nlines, int cu_count, bool lvl1 = false)
{
    int64_t gx = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 8;
    if (gx > cap) gx = cap;
    const int64_t slab = (l_env("WL_SLAB_LINES", 32768) > 0) ? l_env("WL_SLAB_LINES", 32768) : 32768;
    for (int64_t l0 = 0; l0 < nlines; l0 += slab) {
        const int64_t nl = (nlines - l0 < slab) ? (nlines - l0) : slab;
        Lift3Args<T> b = a;
        b.src = a.src + l0 * a.src_ls; b.y = a.y + l0 * a.y_ls; b.d1 = a.d1 + l0 * a.d1_ls; b.sdst = a.sdst + l0 * a.s_ls;
        if (lvl1) hipLaunchKernelGGL((k_lift1d_fwd3<T, ID, 1>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
        else hipLaunchKernelGGL((k_lift1d_fwd3<T, ID, 0>), dim3((unsigned)gx, (unsigned)nl), dim3(256), 0, st, b);
    }
}

template <typename T>
static int match_shape(const LiftScheme<T> &sc)
{
    int upd[WL_MAX_STEPS], nc[WL_MAX_STEPS], sh[WL_MAX_STEPS];
    for (int i = 0; i < sc.nsteps; ++i) { upd[i] = sc.step[i].is_update; nc[i] = sc.step[i].nc; sh[i] = sc.step[i].shift; }
    if (shape_matches<0>(sc.nsteps, upd, nc, sh)) return 0;
    if (shape_matches<1>(sc.nsteps, upd, nc, sh)) return 1;
    if (shape_matches<2>(sc.nsteps, upd, nc, sh)) return 2;
    if (shape_matches<3>(sc.nsteps, upd, nc, sh)) return 3;
    if (shape_matches<4>(sc.nsteps, upd, nc, sh)) return 4;
    if (shape_matches<5>(sc.nsteps, upd, nc, sh)) return 5;
    return -1;
}

static inline bool al16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Forward or inverse lifting transform of `nlines` lines (1-D vector: nlines = 1; batched columns).
// Returns 1 in *handled when the whole transform was enqueued by the fast kernels.
// inverse shapes only (IDs 1, 3, 5); false = not an inverse shape
template <typename T>
static bool launch_inv3_id(int id, hipStream_t st, const LiftInv3Args<T> &a, int64_t nlines, int cu_count)
{
    if (id != 1 && id != 3 && id != 5) return false;
    int64_t gx = (a.ntiles + 3) / 4;
    const int64_t cap = (int64_t)cu_count * 8;
    if (gx > cap) gx = cap;
    for (int64_t l0 = 0; l0 < nlines; l0 += 32768) {
        const int64_t nl = (nlines - l0 < 32768) ? (nlines - l0) : 32768;
        LiftInv3Args<T> b = a;
        b.s3 = a.s3 + l0 * a.s3_ls; b.x = a.x + l0 * a.x_ls; b.dst = a.dst + l0 * a.o_ls;
        const dim3 grid((unsigned)gx, (unsigned)nl), block(256);
        if (id == 1) hipLaunchKernelGGL((k_lift1d_inv3<T, 1>), grid, block, 0, st, b);
        else if (id == 3) hipLaunchKernelGGL((k_lift1d_inv3<T, 3>), grid, block, 0, st, b);
        else hipLaunchKernelGGL((k_lift1d_inv3<T, 5>), grid, block, 0, st, b);
    }
    return true;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is sticky per (function, device) and costs tens of microseconds:
// raise the limit to the LDS size once per device instead of on every call
static hipError_t lift_max_lds_once(const void *fn, unsigned char (&done)[64])
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (done[dev]) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done[dev] = 1;
    return e;
}

// One 1-D lifting level of lines of ANY even length (known shapes): a lane owns OWN = 32 - 2 HP consecutive pairs, loads the
// 32-pair window around them (periodic wrap per sample) straight into registers and runs the level exactly like a tile row of
// k_lift2d_gtile.  Fallback of the lines the streaming kernels decline (lengths that are not multiples of 8, short
// non-power-of-two lines) -- a 10^6-sample cdf9/7 transform ran as six one-thread-per-element launches per level before.
// Arguments as for k_lift1d_stream (fw: a = src, o0 = s destination, o1 = d destination; inv: a = s source, b = d source,
// o0 = destination).
template <typename T, int ID, int FW>
__global__ void __launch_bounds__(256) k_lift1d_gtile(Lift1DArgs<T> a)
{
    constexpr int HP = LiftReach<ID>::HP, OWN = 32 - 2 * HP;
    const int64_t n = a.n, h = n >> 1;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t own0 = t * OWN;
    if (own0 >= h) return;
    const int64_t line = blockIdx.y;
    int64_t pr0 = own0 - HP;
    while (pr0 < 0) pr0 += h;
    T s[32], d[32];
    if (FW) {
        const T *src = a.a + line * a.a_ls;
        int64_t g = 2 * pr0;
        while (g >= n) g -= n;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            s[k] = src[g];
            if (++g >= n) g -= n;
            d[k] = src[g];
            if (++g >= n) g -= n;
        }
    } else {
        const T *ss = a.a + line * a.a_ls, *ds = a.b + line * a.b_ls;
        int64_t g = pr0;
        while (g >= h) g -= h;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            s[k] = a.norm1 * ss[g];
            d[k] = a.norm2 * ds[g];
            if (++g >= h) g -= h;
        }
    }
    // the periodic index of the window's first pair fits an int for the step function only when h < 2^31
    int kg0;
    {
        int64_t p = pr0;
        while (p >= h) p -= h;
        kg0 = (int)p;
    }
    tile_line_steps<T, ID>(s, d, a.c, kg0, (int)h);
    if (FW) {
        T *so = a.o0 + line * a.o0_ls, *dO = a.o1 + line * a.o1_ls;
#pragma unroll
        for (int k = HP; k < 32 - HP; ++k) {
            const int64_t gk = own0 + (k - HP);
            if (gk < h) { so[gk] = s[k] * a.norm1; dO[gk] = d[k] * a.norm2; }
        }
    } else {
        T *out = a.o0 + line * a.o0_ls;
#pragma unroll
        for (int k = HP; k < 32 - HP; ++k) {
            const int64_t gk = own0 + (k - HP);
            if (gk < h) { out[2 * gk] = s[k]; out[2 * gk + 1] = d[k]; }
        }
    }
}
template <typename T, int FW>
static hipError_t launch_lift1d_gtile(int id, hipStream_t st, const Lift1DArgs<T> &a, int64_t nlines)
{
    const int64_t h = a.n >> 1;
    if (h >= ((int64_t)1 << 31)) return hipErrorInvalidValue;
#define WL_L1G(ID_)                                                                                              \
    {                                                                                                            \
        constexpr int OWN = 32 - 2 * LiftReach<ID_>::HP;                                                         \
        const int64_t nthreads = (h + OWN - 1) / OWN;                                                            \
        for (int64_t l0 = 0; l0 < nlines; l0 += 32768) {                                                         \
            const int64_t nb = (nlines - l0 < 32768) ? (nlines - l0) : 32768;                                    \
            Lift1DArgs<T> b = a;                                                                                 \
            b.a = a.a + l0 * a.a_ls; b.b = a.b ? a.b + l0 * a.b_ls : nullptr;                                    \
            b.o0 = a.o0 + l0 * a.o0_ls; b.o1 = a.o1 ? a.o1 + l0 * a.o1_ls : nullptr;                             \
            hipLaunchKernelGGL((k_lift1d_gtile<T, ID_, FW>), dim3((unsigned)((nthreads + 255) / 256), (unsigned)nb), dim3(256), 0, st, b); \
        }                                                                                                        \
    }
    if (FW) {
        if (id == 0) WL_L1G(0) else if (id == 2) WL_L1G(2) else WL_L1G(4)
    } else {
        if (id == 1) WL_L1G(1) else if (id == 3) WL_L1G(3) else WL_L1G(5)
    }
#undef WL_L1G
    return hipGetLastError();
}

// One lifting pass (split -> steps -> normalize, or normalize -> steps -> merge) along ANY axis of a box of any even extent,
// known shapes: what the rank-generic driver (wl_api.hip) runs instead of 6 one-thread-per-element launches per axis and level
// (3-D volumes of any size, in-place 2-D level 1, ...).  Same register-window scheme as k_lift1d_gtile, arbitrary strides;
// the approximation of the low corner goes to / comes from the next level's buffer exactly as in the generic kernels.
template <typename T>
struct LiftAnyArgs {
    const T *src; Strides3 sst;
    T *dst; Strides3 dst_st;
    T *ll; Strides3 ll_st;          // fw: destination of the low corner's s half (or nullptr); inv: its source (or nullptr)
    int n[3], lo[3];
    int axis;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <typename T, int ID, int FW>
__global__ void __launch_bounds__(256) k_lift_any(LiftAnyArgs<T> a)
{
    constexpr int HP = LiftReach<ID>::HP, OWN = 32 - 2 * HP;
    const int axis = a.axis;
    const int nax = a.n[axis], h = nax >> 1, ntile = (h + OWN - 1) / OWN;
    int e[3] = {a.n[0], a.n[1], a.n[2]};
    e[axis] = ntile;
    const int64_t sa = a.sst.s[axis], da = a.dst_st.s[axis];
    for (int i2 = blockIdx.z; i2 < e[2]; i2 += gridDim.z)
        for (int i1 = blockIdx.y * blockDim.y + threadIdx.y; i1 < e[1]; i1 += gridDim.y * blockDim.y)
            for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < e[0]; i0 += gridDim.x * blockDim.x) {
                int c[3] = {i0, i1, i2};
                const int own0 = c[axis] * OWN;
                int64_t base = 0, dbase = 0, lbase = 0;
                bool low = (a.ll != nullptr);
#pragma unroll
                for (int d = 0; d < 3; ++d)
                    if (d != axis) {
                        base += (int64_t)c[d] * a.sst.s[d];
                        dbase += (int64_t)c[d] * a.dst_st.s[d];
                        lbase += (int64_t)c[d] * a.ll_st.s[d];
                        low = low && (c[d] < a.lo[d]);
                    }
                int pr0 = own0 - HP;
                while (pr0 < 0) pr0 += h;
                while (pr0 >= h) pr0 -= h;
                T s[32], d[32];
                if (FW) {
                    const T *p = a.src + base;
                    int g = 2 * pr0;
#pragma unroll
                    for (int k = 0; k < 32; ++k) {
                        s[k] = p[(int64_t)g * sa];
                        if (++g >= nax) g -= nax;
                        d[k] = p[(int64_t)g * sa];
                        if (++g >= nax) g -= nax;
                    }
                } else {
                    const T *ps = low ? (a.ll + lbase) : (a.src + base);
                    const int64_t ss = low ? a.ll_st.s[axis] : sa;
                    const T *pd = a.src + base + (int64_t)h * sa;
                    int g = pr0;
#pragma unroll
                    for (int k = 0; k < 32; ++k) {
                        s[k] = a.norm1 * ps[(int64_t)g * ss];
                        d[k] = a.norm2 * pd[(int64_t)g * sa];
                        if (++g >= h) g -= h;
                    }
                }
                tile_line_steps<T, ID>(s, d, a.c, pr0, h);
#pragma unroll
                for (int k = HP; k < 32 - HP; ++k) {
                    const int gk = own0 + (k - HP);
                    if (gk < h) {
                        if (FW) {
                            const T sv = s[k] * a.norm1, dv = d[k] * a.norm2;
                            if (low) a.ll[lbase + (int64_t)gk * a.ll_st.s[axis]] = sv;
                            else a.dst[dbase + (int64_t)gk * da] = sv;
                            a.dst[dbase + (int64_t)(h + gk) * da] = dv;
                        } else {
                            a.dst[dbase + (int64_t)(2 * gk) * da] = s[k];
                            a.dst[dbase + (int64_t)(2 * gk + 1) * da] = d[k];
                        }
                    }
                }
            }
}

// returns false when the scheme shape / extents are not covered (the caller then uses the one-thread-per-element kernels)
template <typename T>
bool lift_any_pass(hipStream_t st, const LiftScheme<T> &sc, int fw, const T *src, Strides3 sst, T *dst, Strides3 dst_st, T *ll,
                   Strides3 ll_st, Extent3 n, int axis, Extent3 lo, hipError_t *err)
{
    *err = hipSuccess;
    const int id = match_shape<T>(sc);
    if (id < 0 || (fw ? (id & 1) : !(id & 1))) return false;
    for (int d = 0; d < 3; ++d)
        if (n.n[d] < 1 || n.n[d] >= ((int64_t)1 << 30)) return false;
    if (n.n[axis] < 2 || (n.n[axis] % 2) != 0 || opt("WL_LIFT_ANY", 1) == 0) return false;
    LiftAnyArgs<T> a;
    a.src = src; a.sst = sst; a.dst = dst; a.dst_st = dst_st; a.ll = ll; a.ll_st = ll_st; a.axis = axis;
    for (int d = 0; d < 3; ++d) { a.n[d] = (int)n.n[d]; a.lo[d] = (int)lo.n[d]; }
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;
#define WL_LANY(ID_, FW_)                                                                                          \
    {                                                                                                              \
        constexpr int OWN = 32 - 2 * LiftReach<ID_>::HP;                                                           \
        int64_t e[3] = {n.n[0], n.n[1], n.n[2]};                                                                   \
        e[axis] = ((n.n[axis] >> 1) + OWN - 1) / OWN;                                                              \
        int bx = 256;                                                                                              \
        while (bx > 1 && (bx >> 1) >= e[0]) bx >>= 1;                                                              \
        const int by = 256 / bx;                                                                                   \
        int64_t gx = (e[0] + bx - 1) / bx, gy = (e[1] + by - 1) / by, gz = e[2];                                   \
        if (gy > 65535) gy = 65535;                                                                                \
        if (gz > 65535) gz = 65535;                                                                                \
        while (gx * gy * gz > 8192) {                                                                              \
            if (gz > 1 && gz >= gy && gz >= gx) gz = (gz + 1) / 2;                                                 \
            else if (gy > 1 && gy >= gx) gy = (gy + 1) / 2;                                                        \
            else gx = (gx + 1) / 2;                                                                                \
        }                                                                                                          \
        hipLaunchKernelGGL((k_lift_any<T, ID_, FW_>), dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3((unsigned)bx, (unsigned)by, 1), 0, \
                           st, a);                                                                                 \
    }
    switch (id) {
    case 0: WL_LANY(0, 1) break;
    case 2: WL_LANY(2, 1) break;
    case 4: WL_LANY(4, 1) break;
    case 1: WL_LANY(1, 0) break;
    case 3: WL_LANY(3, 0) break;
    default: WL_LANY(5, 0) break;
    }
#undef WL_LANY
    *err = hipGetLastError();
    return true;
}
template bool lift_any_pass<float>(hipStream_t, const LiftScheme<float> &, int, const float *, Strides3, float *, Strides3, float *, Strides3,
                                   Extent3, int, Extent3, hipError_t *);
template bool lift_any_pass<double>(hipStream_t, const LiftScheme<double> &, int, const double *, Strides3, double *, Strides3, double *, Strides3,
                                    Extent3, int, Extent3, hipError_t *);

template <typename T>
int lifting_lines_fast(void *ws, int cu_count, hipStream_t st, int64_t n, int64_t nlines, int64_t ld,
                       T *y, const T *x, const LiftScheme<T> &sc, int L, int fw,
                       int *handled, const char **kernel_name, int *hip_err)
{
    *handled = 0;
    constexpr int VEC = 16 / sizeof(T);
    const int id = match_shape<T>(sc);
    if (L < 1 || !al16(x) || !al16(y)) return WL_OK;
    if (nlines > 1 && (ld % VEC) != 0) return WL_OK;
    // the LDS tail can hold up to lift_tail_cap samples; a single line (or a few) hands over later, at 2048
    // samples, because one workgroup is slow on a long line while the streaming kernels use the whole chip
    int cap = (nlines >= 32) ? lift_tail_cap<T>() : 2048;
    // known forward shapes on power-of-two lines: the single-wave register tail takes over at 4096 / 2048 samples
    const bool reg_tail = fw && l_env("WL_LIFT_REGTAIL", 1) && (id == 0 || id == 2 || id == 4) && (n & (n - 1)) == 0;
    const bool reg_tail_inv = !fw && l_env("WL_LIFT_REGTAIL", 1) && (id == 1 || id == 3 || id == 5) && (n & (n - 1)) == 0 &&
                              (n >> L) <= 64;
    if (reg_tail || reg_tail_inv) cap = (sizeof(T) == 4) ? 4096 : 2048;
    // every level must be either stream-able (known shape, n_l >= 512, n_l % 8 == 0) or inside the tail
    int l_tail = L + 1;                       // first level (1-based) handled by the tail (fw) ...
    for (int l = 1; l <= L; ++l) {
        const int64_t nl = n >> (l - 1);
        if (nl <= cap) { l_tail = l; break; }
        if (id < 0) return WL_OK;                 // (known shapes: lines that cannot stream -- nl < 512 or nl % 8 != 0 -- take k_lift1d_gtile)
    }
    const int64_t N = n * nlines;
    Work<T> w = carve<T>(ws, N);
    const char *dom = nullptr;
#define WL_LAUNCH_ID(FWV)                                                                    \
    switch (id) {                                                                            \
    case 0: launch_stream_id<T, 0, FWV>(st, a, nlines, cu_count); break;                     \
    case 1: launch_stream_id<T, 1, FWV>(st, a, nlines, cu_count); break;                     \
    case 2: launch_stream_id<T, 2, FWV>(st, a, nlines, cu_count); break;                     \
    case 3: launch_stream_id<T, 3, FWV>(st, a, nlines, cu_count); break;                     \
    case 4: launch_stream_id<T, 4, FWV>(st, a, nlines, cu_count); break;                     \
    default: launch_stream_id<T, 5, FWV>(st, a, nlines, cu_count); break;                    \
    }
#define WL_CHECK_LAUNCH()                                                                    \
    do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) { if (hip_err) *hip_err = (int)e__; return WL_EHIP; } } while (0)

    Lift1DArgs<T> a;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
    a.norm1 = sc.norm1; a.norm2 = sc.norm2;

    if (fw) {
        const bool inplace = (y == x);
        const T *cur = x;
        int64_t cur_ls = ld;
        int pp = 0;
        for (int l = 1; l < l_tail && l <= L; ++l) {
            const int64_t nl = n >> (l - 1), hl = nl >> 1;
            const bool last = (l == L);
            // three levels per launch while the line is long (forward shapes 0/2/4), not for the in-place
            // first level (its outputs would land in regions other waves still read)
            if ((id == 0 || id == 2 || id == 4) && l_env("WL_LIFT3", 1) && (L - l + 1) >= 3 && (l + 2 < l_tail) &&
                nl >= 4096 && (nl % 32) == 0 && !(inplace && l == 1)) {
                const bool last3 = (l + 2 == L);
                T *llbuf3 = pp ? w.B : w.A;
                Lift3Args<T> a3;
                for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                    for (int k = 0; k < WL_MAX_NCOEF; ++k) a3.c[i][k] = a.c[i][k];
                a3.norm1 = a.norm1; a3.norm2 = a.norm2;
                a3.src = cur; a3.src_ls = cur_ls; a3.y = y; a3.y_ls = ld; a3.d1 = y + hl; a3.d1_ls = ld;
                a3.sdst = last3 ? y : llbuf3; a3.s_ls = last3 ? ld : (nl >> 3);
                a3.n = nl; a3.ntiles = (hl + 223) / 224;
                if (id == 0) launch_fwd3_id<T, 0>(st, a3, nlines, cu_count, l == 1);
                else if (id == 2) launch_fwd3_id<T, 2>(st, a3, nlines, cu_count, l == 1);
                else launch_fwd3_id<T, 4>(st, a3, nlines, cu_count, l == 1);
                WL_CHECK_LAUNCH();
                if (!dom) dom = "k_lift1d_fwd3";
                cur = llbuf3; cur_ls = nl >> 3; pp ^= 1;
                l += 2;
                continue;
            }
            T *llbuf = pp ? w.B : w.A;
            a.a = cur; a.a_ls = cur_ls; a.b = nullptr; a.b_ls = 0;
            // in place, level 1 reads all of y while other waves would already write s / d into it:
            // stage both halves in the workspace and copy them back afterwards
            const bool stage = inplace && (l == 1);
            a.o0 = (last && !stage) ? y : llbuf; a.o0_ls = (last && !stage) ? ld : hl;
            a.o1 = stage ? w.W : (y + hl); a.o1_ls = stage ? hl : ld;
            a.n = nl; a.ntiles = (hl + 247) / 248;
            const bool streamable = nl >= 512 && (nl % 8) == 0;
            if (streamable) { WL_LAUNCH_ID(1); }
            else { hipError_t eg = launch_lift1d_gtile<T, 1>(id, st, a, nlines); if (eg != hipSuccess) { if (hip_err) *hip_err = (int)eg; return WL_EHIP; } }
            WL_CHECK_LAUNCH();
            if (!streamable && !dom) dom = "k_lift1d_gtile";
            if (stage) {
                Extent3 e = {{hl, nlines, 1}};
                Strides3 s0 = {{1, hl, hl * nlines}}, s1 = {{1, ld, ld * nlines}};
                hipError_t e2 = generic_copy_box<T>(st, w.W, s0, y + hl, s1, e);
                if (e2 == hipSuccess && last) e2 = generic_copy_box<T>(st, llbuf, s0, y, s1, e);
                if (e2 != hipSuccess) { if (hip_err) *hip_err = (int)e2; return WL_EHIP; }
            }
            if (!dom) dom = "k_lift1d_stream";
            cur = llbuf; cur_ls = hl; pp ^= 1;
        }
        if (l_tail <= L && reg_tail &&
            lift_reg_ok<T>(id, n >> (l_tail - 1), L - l_tail + 1, cur, cur_ls, y, ld)) {
            LiftRegArgs<T> r;
            r.src = cur; r.src_item = cur_ls; r.y = y; r.y_item = ld; r.n0 = (int)(n >> (l_tail - 1)); r.nlev = L - l_tail + 1;
            for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                for (int k = 0; k < WL_MAX_NCOEF; ++k) r.c[i][k] = a.c[i][k];
            r.norm1 = a.norm1; r.norm2 = a.norm2;
            if (id == 0) hipLaunchKernelGGL((k_tail_lift_reg<T, 0>), dim3((unsigned)nlines), dim3(64), 0, st, r);
            else if (id == 2) hipLaunchKernelGGL((k_tail_lift_reg<T, 2>), dim3((unsigned)nlines), dim3(64), 0, st, r);
            else hipLaunchKernelGGL((k_tail_lift_reg<T, 4>), dim3((unsigned)nlines), dim3(64), 0, st, r);
            WL_CHECK_LAUNCH();
            if (!dom) dom = "k_tail_lift_reg";
        } else if (l_tail <= L) {
            LiftTailArgs<T> t;
            const int64_t nl = n >> (l_tail - 1);
            t.src = cur; t.src_item = cur_ls; t.y = y; t.y_item = ld; t.ll = nullptr; t.ll_item = 0;
            t.n0 = (int)nl; t.nlev = L - l_tail + 1; t.cap = (int)((nl + 15) & ~15);
            const size_t shmem = 2 * (size_t)t.cap * sizeof(T);
            static unsigned char attr_fw[64] = {0};
            hipError_t e = lift_max_lds_once(reinterpret_cast<const void *>(&k_tail_lift<T, 1>), attr_fw);
            if (e != hipSuccess) { if (hip_err) *hip_err = (int)e; return WL_EHIP; }
            int threads = nl >= 4096 ? 1024 : (nl >= 512 ? 256 : 64);
            hipLaunchKernelGGL((k_tail_lift<T, 1>), dim3((unsigned)nlines), dim3(threads), shmem, st, t, sc);
            WL_CHECK_LAUNCH();
            if (!dom) dom = "k_tail_lift";
        }
    } else {
        // inverse: levels L..1; the tail does the deepest levels (outputs up to `cap` samples)
        const T *llsrc = x;                  // approximation source of the current level
        int64_t ll_ls = ld;
        int l = L;
        int pp = 0;
        // deepest levels whose OUTPUT fits the tail: output length of level l is n >> (l-1)
        int l_hi = L;                        // tail covers levels L .. l_lo
        int l_lo = L + 1;
        for (int q = L; q >= 1; --q) { if ((n >> (q - 1)) <= cap) l_lo = q; else break; }
        if (l_lo <= L && l_env("WL_LIFT_REGTAIL", 1) != 0 &&
            lift_reg_inv_ok<T>(id, n >> (l_lo - 1), L - l_lo + 1, x, ld, (l_lo == 1) ? y : (pp ? w.B : w.A), (l_lo == 1) ? ld : (n >> (l_lo - 1)))) {
            // known inverse shapes on power-of-two lines: the single-wave register tail
            const int64_t nout = n >> (l_lo - 1);
            const bool to_y = (l_lo == 1);
            T *out = to_y ? y : (pp ? w.B : w.A);
            LiftRegArgs<T> r;
            r.src = x; r.src_item = ld; r.y = out; r.y_item = to_y ? ld : nout; r.n0 = (int)nout; r.nlev = L - l_lo + 1;
            for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                for (int k = 0; k < WL_MAX_NCOEF; ++k) r.c[i][k] = a.c[i][k];
            r.norm1 = a.norm1; r.norm2 = a.norm2;
            if (id == 1) hipLaunchKernelGGL((k_tail_lift_reg_inv<T, 1>), dim3((unsigned)nlines), dim3(64), 0, st, r);
            else if (id == 3) hipLaunchKernelGGL((k_tail_lift_reg_inv<T, 3>), dim3((unsigned)nlines), dim3(64), 0, st, r);
            else hipLaunchKernelGGL((k_tail_lift_reg_inv<T, 5>), dim3((unsigned)nlines), dim3(64), 0, st, r);
            WL_CHECK_LAUNCH();
            if (!dom) dom = "k_tail_lift_reg_inv";
            llsrc = out; ll_ls = r.y_item; pp ^= 1;
            l = l_lo - 1;
        } else if (l_lo <= L) {
            LiftTailArgs<T> t;
            const int64_t nout = n >> (l_lo - 1);
            const bool to_y = (l_lo == 1);
            T *out = to_y ? y : (pp ? w.B : w.A);
            t.src = x; t.src_item = ld; t.ll = x; t.ll_item = ld;
            t.y = out; t.y_item = to_y ? ld : nout;
            t.n0 = (int)nout; t.nlev = l_hi - l_lo + 1; t.cap = (int)((nout + 15) & ~15);
            const size_t shmem = 2 * (size_t)t.cap * sizeof(T);
            static unsigned char attr_inv[64] = {0};
            hipError_t e = lift_max_lds_once(reinterpret_cast<const void *>(&k_tail_lift<T, 0>), attr_inv);
            if (e != hipSuccess) { if (hip_err) *hip_err = (int)e; return WL_EHIP; }
            int threads = nout >= 4096 ? 1024 : (nout >= 512 ? 256 : 64);
            hipLaunchKernelGGL((k_tail_lift<T, 0>), dim3((unsigned)nlines), dim3(threads), shmem, st, t, sc);
            WL_CHECK_LAUNCH();
            if (!dom) dom = "k_tail_lift";
            llsrc = out; ll_ls = t.y_item; pp ^= 1;
            l = l_lo - 1;
        }
        for (; l >= 1; --l) {
            // three levels (l, l-1, l-2) per launch while the output of level l-2 is a long line
            if (l >= 3 && l_env("WL_NO_LIFT_INV3", 0) == 0) {
                const int64_t n3 = n >> (l - 3);                 // output length of level l-2
                if (n3 >= 4096 && (n3 % 64) == 0 && (nlines == 1 || ((ld % VEC) == 0 && (ll_ls % VEC) == 0))) {
                    const bool to_y3 = (l - 2 == 1);
                    T *out3 = to_y3 ? y : (pp ? w.B : w.A);
                    const bool stage3 = to_y3 && (y == x);       // in place: the details of y are still being read
                    LiftInv3Args<T> q;
                    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                        for (int k = 0; k < WL_MAX_NCOEF; ++k) q.c[i][k] = a.c[i][k];
                    q.norm1 = a.norm1; q.norm2 = a.norm2;
                    q.s3 = llsrc; q.s3_ls = ll_ls; q.x = x; q.x_ls = ld;
                    q.dst = stage3 ? w.W : out3; q.o_ls = stage3 ? n3 : (to_y3 ? ld : n3);
                    q.n = n3; q.ntiles = ((n3 >> 3) + 55) / 56;
                    if (launch_inv3_id<T>(id, st, q, nlines, cu_count)) {
                        WL_CHECK_LAUNCH();
                        if (stage3) {
                            Extent3 e = {{n3, nlines, 1}};
                            Strides3 s0 = {{1, n3, n3 * nlines}}, s1 = {{1, ld, ld * nlines}};
                            hipError_t e2 = generic_copy_box<T>(st, w.W, s0, y, s1, e);
                            if (e2 != hipSuccess) { if (hip_err) *hip_err = (int)e2; return WL_EHIP; }
                        }
                        dom = "k_lift1d_inv3";
                        llsrc = out3; ll_ls = n3; pp ^= 1;
                        l -= 2;
                        continue;
                    }
                }
            }
            const int64_t nl = n >> (l - 1), hl = nl >> 1;
            const bool to_y = (l == 1);
            T *out = to_y ? y : (pp ? w.B : w.A);
            a.a = llsrc; a.a_ls = ll_ls; a.b = x + hl; a.b_ls = ld;
            // in place (y == x), level 1 writes all of y while detail d1 = y[n/2..n) is still being read
            // by other waves: stage the output in the work buffer and copy back
            const bool stage = to_y && (y == x);
            a.o0 = stage ? w.W : out; a.o0_ls = stage ? nl : (to_y ? ld : nl);
            a.o1 = nullptr; a.o1_ls = 0;
            a.n = nl; a.ntiles = (hl + 247) / 248;
            const bool streamable = nl >= 512 && (nl % 8) == 0;
            if (streamable) { WL_LAUNCH_ID(0); }
            else { hipError_t eg = launch_lift1d_gtile<T, 0>(id, st, a, nlines); if (eg != hipSuccess) { if (hip_err) *hip_err = (int)eg; return WL_EHIP; } }
            WL_CHECK_LAUNCH();
            if (stage) {
                Extent3 e = {{nl, nlines, 1}};
                Strides3 s0 = {{1, nl, nl * nlines}}, s1 = {{1, ld, ld * nlines}};
                hipError_t e2 = generic_copy_box<T>(st, w.W, s0, y, s1, e);
                if (e2 != hipSuccess) { if (hip_err) *hip_err = (int)e2; return WL_EHIP; }
            }
            dom = streamable ? "k_lift1d_stream" : "k_lift1d_gtile";
            llsrc = out; ll_ls = nl; pp ^= 1;
        }
    }
#undef WL_LAUNCH_ID
#undef WL_CHECK_LAUNCH
    *handled = 1;
    if (kernel_name) *kernel_name = dom ? dom : "none";
    return WL_OK;
}

template int lifting_lines_fast<float>(void *, int, hipStream_t, int64_t, int64_t, int64_t, float *, const float *,
                                       const LiftScheme<float> &, int, int, int *, const char **, int *);
template int lifting_lines_fast<double>(void *, int, hipStream_t, int64_t, int64_t, int64_t, double *, const double *,
                                        const LiftScheme<double> &, int, int, int *, const char **, int *);

// --------------------------------------------------------------------------------------------------
// One lifting level along a STRIDED axis (the dim-2 pass of a 2-D level) without transposes: lanes own 16 bytes of
// consecutive rows, the wave marches along the axis pair by pair.  The steps run as a software cascade on a
// register ring of (s, d) pairs: at time tau step k updates pair tau - D[k], with the delays D chosen so that
// every operand is exactly in the state the sequential algorithm would see (all earlier steps done there, no later
// step yet).  A chunk starts VM pairs early (the left reach of the dependency cone) and runs D_last pairs past
// its end; nothing crosses lanes.  Rounding as in k_lift1d_stream (in-bounds vs wrapped form per element; the
// pair index is uniform across the wave).
struct CascadeTab { int D[WL_MAX_STEPS]; int DL, VM, AMIN; };
template <int ID>
struct Cascade {
    typedef Shape<ID> SH;
    static constexpr int NS = SH::NS;
    static constexpr int a(int k) { return -SH::S[k].sh; }
    static constexpr int b(int k) { return SH::S[k].nc - 1 - SH::S[k].sh; }
    static constexpr CascadeTab make()
    {
        CascadeTab t = {};
        // delay of step k: see DESIGN.md ("lifting along a strided axis")
        for (int k = 0; k < NS; ++k) {
            int d = (k > 0) ? t.D[k - 1] : 0;
            int bo = b(k);                                  // raw operands: pair tau must be loaded
            for (int m = 0; m < k; ++m) {
                if (SH::S[m].upd != SH::S[k].upd) {         // m writes k's operand array and reads k's target array
                    if (t.D[m] + b(k) > bo) bo = t.D[m] + b(k);
                    if (t.D[m] - a(m) > d) d = t.D[m] - a(m);
                }
            }
            if (bo > d) d = bo;
            t.D[k] = d < 0 ? 0 : d;
        }
        t.DL = t.D[NS - 1];
        // first valid output pair relative to the first raw pair loaded
        int cur_s = 0, cur_d = 0;
        for (int k = 0; k < NS; ++k) {
            const int o = SH::S[k].upd ? cur_s : cur_d, tg = SH::S[k].upd ? cur_d : cur_s;
            int v = -t.D[k];
            if (o - a(k) > v) v = o - a(k);
            if (tg > v) v = tg;
            if (SH::S[k].upd) cur_d = v; else cur_s = v;
        }
        t.VM = cur_s > cur_d ? cur_s : cur_d;
        t.AMIN = 0;
        for (int k = 0; k < NS; ++k) if (a(k) < t.AMIN) t.AMIN = a(k);
        return t;
    }
    static constexpr CascadeTab TAB = make();
};

template <typename T>
struct LiftAxisArgs {
    const T *src; int64_t lds; int64_t bs_src;     // column stride, batch stride (blockIdx.y)
    T *dst; int64_t ldd; int64_t bs_dst;
    int64_t R, C;                                  // rows (contiguous), axis length
    int TP;                                        // output pairs per chunk (multiple of 8)
    int nstrips, nchunks;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <typename T, int ID, int FW, int RPL>
__global__ void __launch_bounds__(64) k_lift_axis_stream(LiftAxisArgs<T> a)
{
    typedef Shape<ID> SH;
    typedef Cascade<ID> CS;
    constexpr int R = 8, DL = CS::TAB.DL, VM = CS::TAB.VM, PF = R - DL + CS::TAB.AMIN - 1;
    static_assert(PF >= 2, "ring too small for this scheme");
    const int lane = threadIdx.x;
    const int strip = (int)(blockIdx.x % (unsigned)a.nstrips);
    const int chunk = (int)(blockIdx.x / (unsigned)a.nstrips);
    const int64_t row = ((int64_t)strip * 64 + lane) * RPL;
    const bool valid = row < a.R;
    const int64_t rr = valid ? row : 0;
    const int64_t half = a.C >> 1;
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < half) ? (p0 + a.TP) : half;
    const T *base = a.src + (int64_t)blockIdx.y * a.bs_src + rr;
    T *out = a.dst + (int64_t)blockIdx.y * a.bs_dst + rr;
    T rs[R][RPL], rd[R][RPL];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int q = 0; q < RPL; ++q) { rs[i][q] = (T)0; rd[i][q] = (T)0; }
    const int64_t tau0 = p0 - VM;                       // first raw pair of this chunk (may be negative: wraps)
    // periodic pair index: the arguments stay within a few ring lengths of [0, half), so a subtraction loop (zero or one trip
    // unless the level is tiny) replaces the emulated 64-bit division that dominated the scalar instruction stream
    auto wrapi = [&](int64_t i) __attribute__((always_inline)) {
        int j = (int)i;
        const int hh = (int)half;
        while (j < 0) j += hh;
        while (j >= hh) j -= hh;
        return (int64_t)j;
    };
    // pair index tau lives in ring slot (tau - tau0) mod R: static once the loop is unrolled by R
    auto load_pair = [&](const int64_t tau, const int slot) __attribute__((always_inline)) {
        const int64_t iw = wrapi(tau);
        if (FW) {
            ldv_l<T, RPL>(base + (2 * iw) * a.lds, rs[slot]);
            ldv_l<T, RPL>(base + (2 * iw + 1) * a.lds, rd[slot]);
        } else {
            T sv[RPL], dv[RPL];
            ldv_l<T, RPL>(base + iw * a.lds, sv);
            ldv_l<T, RPL>(base + (half + iw) * a.lds, dv);
#pragma unroll
            for (int q = 0; q < RPL; ++q) { rs[slot][q] = a.norm1 * sv[q]; rd[slot][q] = a.norm2 * dv[q]; }
        }
    };
#pragma unroll
    for (int c = 0; c < PF; ++c) load_pair(tau0 + c, c % R);
    auto step = [&](const int64_t t, const int u) __attribute__((always_inline)) {
        const int64_t tau = tau0 + t;
        load_pair(tau + PF, (u + PF) % R);
#pragma unroll
        for (int k = 0; k < SH::NS; ++k) {
            const int upd = SH::S[k].upd, nc = SH::S[k].nc, ak = -SH::S[k].sh, Dk = CS::TAB.D[k];
            const int slot = ((u - Dk) % R + R) % R;
            const int64_t iw = wrapi(tau - Dk);
            const bool inb = (iw + ak >= 0) && (iw + ak + nc - 1 <= half - 1);
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                T o[3] = {(T)0, (T)0, (T)0};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    if (kk < nc) o[kk] = upd ? rs[((u - Dk + ak + kk) % R + R) % R][q] : rd[((u - Dk + ak + kk) % R + R) % R][q];
                const T x = upd ? rd[slot][q] : rs[slot][q];
                T acc = a.c[k][0] * o[0];
                if (nc > 1) acc = acc + a.c[k][1] * o[1];
                if (nc > 2) acc = acc + a.c[k][2] * o[2];
                const T xin = x + acc;
                T xb = x + a.c[k][0] * o[0];
                if (nc > 1) xb = xb + a.c[k][1] * o[1];
                if (nc > 2) xb = xb + a.c[k][2] * o[2];
                const T res = inb ? xin : xb;
                if (upd) rd[slot][q] = res; else rs[slot][q] = res;
            }
        }
        const int64_t io = tau - DL;
        if (valid && io >= p0 && io < pend) {
            const int slot = ((u - DL) % R + R) % R;
            if (FW) {
                T so[RPL], dO[RPL];
#pragma unroll
                for (int q = 0; q < RPL; ++q) { so[q] = rs[slot][q] * a.norm1; dO[q] = rd[slot][q] * a.norm2; }
                stv_l<T, RPL>(out + io * a.ldd, so);
                stv_l<T, RPL>(out + (half + io) * a.ldd, dO);
            } else {
                stv_l<T, RPL>(out + (2 * io) * a.ldd, rs[slot]);
                stv_l<T, RPL>(out + (2 * io + 1) * a.ldd, rd[slot]);
            }
        }
    };
    const int64_t nstep = (pend - p0) + VM + DL;
    for (int64_t t0 = 0; t0 < nstep; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u);
    }
}

// --------------------------------------------------------------------------------------------------
// One whole 2-D lifting level in a single pass over HBM (forward): the dim-2 pass is the register cascade of
// k_lift_axis_stream (lanes own 4 consecutive rows, the wave marches along the columns); every finished column pair
// (its s-column and its d-column) is then lifted along dim 1 ACROSS the lanes (2 row pairs per lane, neighbours by
// DPP exactly as in k_lift1d_stream) and scattered to the four quadrants.  Strips overlap by one lane on each side.
// halo lanes on either side of a strip: 4, so that the strip pitch is 224 rows = 7 x 128 bytes of Float32 -- strips whose pitch is
// not a multiple of the 128-byte line were measured ~15 % slower in pure data movement (tools/probes/march_probe.hip: 134 vs
// 111-118 us for 8192^2), which is what the 240- / 248-row pitches of the first version cost (level 1: 141 -> see DESIGN)
constexpr int kLift2dML = 4;
template <typename T>
struct Lift2DArgs {
    const T *src; int64_t lds;      // fw: block to transform          inv: coefficient array
    T *y; int64_t ldy;              // fw: coefficient array           inv: result block
    T *ll; int64_t ldl;             // fw: approximation quadrant destination (nullptr: y)   inv: approximation source (nullptr: src)
    int64_t n0, n1;
    int TP, nstrips, nchunks;
    // batch of independent blocks (blockIdx.y: the planes of a 3-D level); only the first nll blocks use ll
    int64_t bs_src, bs_y, bs_ll; int nll;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

// R: ring slots = steps per unrolled iteration; loads run PF = R - DL + AMIN - 1 column pairs ahead (R = 8 everywhere: 16 was
// measured on the small, latency-bound levels and lost).
template <typename T, int ID, int R, bool FAST>
__device__ __forceinline__ void lift2d_fwd_body(const Lift2DArgs<T> &a)
{
    typedef Shape<ID> SH;
    typedef Cascade<ID> CS;
    constexpr int RPL = 4, DL = CS::TAB.DL, VM = CS::TAB.VM, PF = R - DL + CS::TAB.AMIN - 1, ML = kLift2dML;   // even margin: lane pairs store together
    constexpr int VR = (64 - 2 * ML) * RPL;
    static_assert(PF >= 2, "ring too small for this scheme");
    const int lane = threadIdx.x;
    const int strip = (int)(blockIdx.x % (unsigned)a.nstrips);
    const int chunk = (int)(blockIdx.x / (unsigned)a.nstrips);
    const int64_t h0 = a.n0 >> 1, h1 = a.n1 >> 1;
    const int64_t gi = (int64_t)strip * VR + (int64_t)(lane - ML) * RPL;      // first row of this lane (may wrap)
    int64_t row = gi;
    if (row < 0) row += a.n0;
    if (row >= a.n0) row -= a.n0;
    const bool valid = lane >= ML && lane < 64 - ML && gi < a.n0;
    const int64_t kfirst = row >> 1, k0 = gi >> 1;
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < h1) ? (p0 + a.TP) : h1;
    const T *base = a.src + (int64_t)blockIdx.y * a.bs_src + row;
    T rs[R][RPL], rd[R][RPL];
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int q = 0; q < RPL; ++q) { rs[i][q] = (T)0; rd[i][q] = (T)0; }
    const int64_t tau0 = p0 - VM;
    // periodic pair index: the arguments stay within a few ring lengths of [0, h1), so a subtraction loop (zero or one trip
    // unless the level is tiny) replaces the emulated 64-bit division that dominated the scalar instruction stream
    auto wrapi = [&](int64_t i) __attribute__((always_inline)) {
        int j = (int)i;
        const int hh = (int)h1;
        while (j < 0) j += hh;
        while (j >= hh) j -= hh;
        return (int64_t)j;
    };
    auto load_pair = [&](const int64_t tau, const int slot) __attribute__((always_inline)) {
        const int64_t iw = wrapi(tau);
        ldg_pol<WL_P_LIFT2DF_LD != 0, T, RPL>(base + (2 * iw) * a.lds, rs[slot]);
        ldg_pol<WL_P_LIFT2DF_LD != 0, T, RPL>(base + (2 * iw + 1) * a.lds, rd[slot]);
    };
#pragma unroll
    for (int c = 0; c < PF; ++c) load_pair(tau0 + c, c % R);
    T *const yb = a.y + (int64_t)blockIdx.y * a.bs_y;
    const bool to_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    T *const llp = to_ll ? a.ll + (int64_t)blockIdx.y * a.bs_ll : yb;
    const int64_t ldl = to_ll ? a.ldl : a.ldy;
    auto step = [&](const int64_t t, const int u) __attribute__((always_inline)) {
        const int64_t tau = tau0 + t;
        load_pair(tau + PF, (u + PF) % R);
#pragma unroll
        for (int k = 0; k < SH::NS; ++k) {
            const int upd = SH::S[k].upd, nc = SH::S[k].nc, ak = -SH::S[k].sh, Dk = CS::TAB.D[k];
            const int slot = ((u - Dk) % R + R) % R;
            const int64_t iw = wrapi(tau - Dk);
            const bool inb = (iw + ak >= 0) && (iw + ak + nc - 1 <= h1 - 1);
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                T o[3] = {(T)0, (T)0, (T)0};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    if (kk < nc) o[kk] = upd ? rs[((u - Dk + ak + kk) % R + R) % R][q] : rd[((u - Dk + ak + kk) % R + R) % R][q];
                const T x = upd ? rd[slot][q] : rs[slot][q];
                T acc = a.c[k][0] * o[0];
                if (nc > 1) acc = acc + a.c[k][1] * o[1];
                if (nc > 2) acc = acc + a.c[k][2] * o[2];
                const T xin = x + acc;
                T res = xin;
                if constexpr (!FAST) {
                    T xb = x + a.c[k][0] * o[0];
                    if (nc > 1) xb = xb + a.c[k][1] * o[1];
                    if (nc > 2) xb = xb + a.c[k][2] * o[2];
                    res = inb ? xin : xb;
                }
                if (upd) rd[slot][q] = res; else rs[slot][q] = res;
            }
        }
        const int64_t io = tau - DL;
        if (io >= p0 && io < pend) {                      // uniform across the wave: the DPP exchanges below need every lane
            const int slot = ((u - DL) % R + R) % R;
            // dim-2 normalize!, then the dim-1 level on both columns (split -> steps -> normalize)
            T s1[2], d1[2], s2[2], d2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s1[j] = rs[slot][2 * j] * a.norm1; d1[j] = rs[slot][2 * j + 1] * a.norm1;
                s2[j] = rd[slot][2 * j] * a.norm2; d2[j] = rd[slot][2 * j + 1] * a.norm2;
            }
            lift_steps_lane<T, ID, 2, FAST>(s1, d1, a.c, kfirst, h0);
            lift_steps_lane<T, ID, 2, FAST>(s2, d2, a.c, kfirst, h0);
            // lane pairs (2i, 2i+1) own rows k0..k0+1 and k0+2..k0+3 of the same four sub-band columns: they swap
            // halves so that the even lane stores 4 rows of LL and HL (left column), the odd lane 4 rows of LH and HH
            const bool odd = (lane & 1) != 0;
            T ll_[2], hl_[2], lh_[2], hh_[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ll_[j] = s1[j] * a.norm1; hl_[j] = d1[j] * a.norm2;
                lh_[j] = s2[j] * a.norm1; hh_[j] = d2[j] * a.norm2;
            }
            T o0[4], o1[4];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const T ra = l_partner(odd ? ll_[j] : lh_[j]);      // even gets the partner's LL, odd the partner's LH
                const T rb = l_partner(odd ? hl_[j] : hh_[j]);
                o0[j] = odd ? ra : ll_[j];      o0[2 + j] = odd ? lh_[j] : ra;
                o1[j] = odd ? rb : hl_[j];      o1[2 + j] = odd ? hh_[j] : rb;
            }
            if (valid) {
                if (!odd) {
                    stg_pol<false, T, 4>(llp + k0 + io * ldl, o0);                                     // LL rows k0..k0+3
                    stg_pol<WL_P_LIFT2DF_ST != 0, T, 4>(yb + h0 + k0 + io * a.ldy, o1);                // HL
                } else {
                    stg_pol<WL_P_LIFT2DF_ST != 0, T, 4>(yb + (k0 - 2) + (h1 + io) * a.ldy, o0);        // LH rows k0-2..k0+1
                    stg_pol<WL_P_LIFT2DF_ST != 0, T, 4>(yb + h0 + (k0 - 2) + (h1 + io) * a.ldy, o1);   // HH
                }
            }
        }
    };
    const int64_t nstep = (pend - p0) + VM + DL;
    for (int64_t t0 = 0; t0 < nstep; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u);
    }
}

// interior workgroups (strip and chunk a few pairs away from the ends of the block in both dimensions: all but the border ones on the
// big levels) run the body without the boundary summation form -- this kernel is bound by VALU issue, not by HBM
template <typename T>
__device__ __forceinline__ bool lift2d_interior(const Lift2DArgs<T> &a)
{
    constexpr int VR = (64 - 2 * kLift2dML) * 4;
    const int strip = (int)(blockIdx.x % (unsigned)a.nstrips), chunk = (int)(blockIdx.x / (unsigned)a.nstrips);
    const int64_t gi0 = (int64_t)strip * VR - kLift2dML * 4, p0 = (int64_t)chunk * a.TP;
    return gi0 >= 16 && gi0 + 256 + 16 <= a.n0 && p0 >= 32 && p0 + a.TP + 32 <= (a.n1 >> 1);
}
template <typename T, int ID, int R = 8>
__global__ void __launch_bounds__(64) k_lift2d_fwd(Lift2DArgs<T> a)
{
    if (lift2d_interior<T>(a)) lift2d_fwd_body<T, ID, R, true>(a);
    else lift2d_fwd_body<T, ID, R, false>(a);
}

// The inverse: per step the raw coefficient column pair (left-half column p: approximation rows + detail rows, and
// right-half column p) is loaded four steps ahead, both columns are reconstructed along dim 1 across the lanes
// (normalize -> steps -> merge), and the results enter the dim-2 inverse cascade as its (s, d) pair.
template <typename T, int ID, int R, bool FAST>
__device__ __forceinline__ void lift2d_inv_body(const Lift2DArgs<T> &a)
{
    typedef Shape<ID> SH;
    typedef Cascade<ID> CS;
    constexpr int RPL = 4, DL = CS::TAB.DL, VM = CS::TAB.VM, PF = R - 4, ML = kLift2dML;
    constexpr int VR = (64 - 2 * ML) * RPL;
    static_assert(R - DL + CS::TAB.AMIN - 1 >= 1, "ring too small for this scheme");
    const int lane = threadIdx.x;
    const int strip = (int)(blockIdx.x % (unsigned)a.nstrips);
    const int chunk = (int)(blockIdx.x / (unsigned)a.nstrips);
    const int64_t h0 = a.n0 >> 1, h1 = a.n1 >> 1;
    const int64_t gi = (int64_t)strip * VR + (int64_t)(lane - ML) * RPL;
    int64_t row = gi;
    if (row < 0) row += a.n0;
    if (row >= a.n0) row -= a.n0;
    const bool valid = lane >= ML && lane < 64 - ML && gi < a.n0;
    const int64_t kw = row >> 1;                           // this lane's first row pair (wrapped)
    const int64_t p0 = (int64_t)chunk * a.TP;
    const int64_t pend = (p0 + a.TP < h1) ? (p0 + a.TP) : h1;
    // left-half columns take their approximation rows from ll when given
    const T *xb = a.src + (int64_t)blockIdx.y * a.bs_src;
    const bool from_ll = (a.ll != nullptr) && ((int)blockIdx.y < a.nll);
    const T *ls_base = (from_ll ? a.ll + (int64_t)blockIdx.y * a.bs_ll : xb) + kw;
    const int64_t ls_ld = from_ll ? a.ldl : a.lds;
    const T *ld_base = xb + h0 + kw;
    const T *rs_base = xb + h1 * a.lds + kw;
    const T *rd_base = rs_base + h0;
    T *const yb = a.y + (int64_t)blockIdx.y * a.bs_y;
    T rs[R][RPL], rd[R][RPL];                              // dim-2 cascade rings (dim-1-reconstructed columns)
    T qLs[R][2], qLd[R][2], qRs[R][2], qRd[R][2];          // raw coefficient columns in flight (slot = pair index mod R)
#pragma unroll
    for (int i = 0; i < R; ++i)
#pragma unroll
        for (int q = 0; q < RPL; ++q) { rs[i][q] = (T)0; rd[i][q] = (T)0; }
    const int64_t tau0 = p0 - VM;
    // periodic pair index: the arguments stay within a few ring lengths of [0, h1), so a subtraction loop (zero or one trip
    // unless the level is tiny) replaces the emulated 64-bit division that dominated the scalar instruction stream
    auto wrapi = [&](int64_t i) __attribute__((always_inline)) {
        int j = (int)i;
        const int hh = (int)h1;
        while (j < 0) j += hh;
        while (j >= hh) j -= hh;
        return (int64_t)j;
    };
    auto load_raw = [&](const int64_t tau, const int slot) __attribute__((always_inline)) {
        const int64_t iw = wrapi(tau);
        ldg_pol<WL_P_LIFT2D_LD != 0, T, 2>(ls_base + iw * ls_ld, qLs[slot]);
        ldg_pol<WL_P_LIFT2D_LD != 0, T, 2>(ld_base + iw * a.lds, qLd[slot]);
        ldg_pol<WL_P_LIFT2D_LD != 0, T, 2>(rs_base + iw * a.lds, qRs[slot]);
        ldg_pol<WL_P_LIFT2D_LD != 0, T, 2>(rd_base + iw * a.lds, qRd[slot]);
    };
#pragma unroll
    for (int c = 0; c < PF; ++c) load_raw(tau0 + c, c % R);
    auto step = [&](const int64_t t, const int u) __attribute__((always_inline)) {
        const int64_t tau = tau0 + t;
        // dim-1 reconstruction of the two raw columns of pair tau -> the cascade's (s, d) pair, dim-2 normalize! applied
        {
            T s1[2], d1[2], s2[2], d2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                s1[j] = a.norm1 * qLs[u][j]; d1[j] = a.norm2 * qLd[u][j];
                s2[j] = a.norm1 * qRs[u][j]; d2[j] = a.norm2 * qRd[u][j];
            }
            lift_steps_lane<T, ID, 2, FAST>(s1, d1, a.c, kw, h0);
            lift_steps_lane<T, ID, 2, FAST>(s2, d2, a.c, kw, h0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                rs[u][2 * j] = a.norm1 * s1[j]; rs[u][2 * j + 1] = a.norm1 * d1[j];          // merge!, then normalize! of dim 2
                rd[u][2 * j] = a.norm2 * s2[j]; rd[u][2 * j + 1] = a.norm2 * d2[j];
            }
        }
        load_raw(tau + PF, (u + PF) % R);
#pragma unroll
        for (int k = 0; k < SH::NS; ++k) {
            const int upd = SH::S[k].upd, nc = SH::S[k].nc, ak = -SH::S[k].sh, Dk = CS::TAB.D[k];
            const int slot = ((u - Dk) % R + R) % R;
            const int64_t iw = wrapi(tau - Dk);
            const bool inb = (iw + ak >= 0) && (iw + ak + nc - 1 <= h1 - 1);
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                T o[3] = {(T)0, (T)0, (T)0};
#pragma unroll
                for (int kk = 0; kk < 3; ++kk)
                    if (kk < nc) o[kk] = upd ? rs[((u - Dk + ak + kk) % R + R) % R][q] : rd[((u - Dk + ak + kk) % R + R) % R][q];
                const T x = upd ? rd[slot][q] : rs[slot][q];
                T acc = a.c[k][0] * o[0];
                if (nc > 1) acc = acc + a.c[k][1] * o[1];
                if (nc > 2) acc = acc + a.c[k][2] * o[2];
                const T xin = x + acc;
                T res = xin;
                if constexpr (!FAST) {
                    T xb = x + a.c[k][0] * o[0];
                    if (nc > 1) xb = xb + a.c[k][1] * o[1];
                    if (nc > 2) xb = xb + a.c[k][2] * o[2];
                    res = inb ? xin : xb;
                }
                if (upd) rd[slot][q] = res; else rs[slot][q] = res;
            }
        }
        const int64_t io = tau - DL;
        if (valid && io >= p0 && io < pend) {
            const int slot = ((u - DL) % R + R) % R;
            stg_pol<WL_P_LIFT2D_ST != 0, T, RPL>(yb + gi + (2 * io) * a.ldy, rs[slot]);
            stg_pol<WL_P_LIFT2D_ST != 0, T, RPL>(yb + gi + (2 * io + 1) * a.ldy, rd[slot]);
        }
    };
    const int64_t nstep = (pend - p0) + VM + DL;
    for (int64_t t0 = 0; t0 < nstep; t0 += R) {
#pragma unroll
        for (int u = 0; u < R; ++u) step(t0 + u, u);
    }
}

template <typename T, int ID, int R = 8>
__global__ void __launch_bounds__(64) k_lift2d_inv(Lift2DArgs<T> a)
{
    if (lift2d_interior<T>(a)) lift2d_inv_body<T, ID, R, true>(a);
    else lift2d_inv_body<T, ID, R, false>(a);
}

template <typename T, int ID>
static hipError_t launch_lift2d_inv(hipStream_t st, Lift2DArgs<T> a, int cu_count, int64_t nbatch = 1)
{
    constexpr int VR = (64 - 2 * kLift2dML) * 4;
    a.nstrips = (int)((a.n0 + VR - 1) / VR);
    const int64_t h1 = a.n1 >> 1;
    // chunk length: these kernels are latency-bound per wave (8-slot ring: loads run three or four steps ahead), so what counts is
    // how many waves are resident -- 40 pairs per chunk puts 3.7 waves on every SIMD for 8192^2 (64: 2.3 waves, 136 -> 125 us;
    // 32: the same time with more cone columns; 24 and below: a second round of workgroups)
    int TP = 40;
    if ((int64_t)a.nstrips * ((h1 + TP - 1) / TP) * nbatch < (int64_t)cu_count * 3) {
        TP = 64;
        while (TP > 8 && (int64_t)a.nstrips * ((h1 + TP - 1) / TP) * nbatch < (int64_t)cu_count * 8) TP >>= 1;
    }
    const int tpo = (int)opt("WL_LIFT_TP", 0);
    if (tpo >= 8 && (tpo % 8) == 0) TP = tpo;
    a.TP = TP;
    a.nchunks = (int)((h1 + TP - 1) / TP);
    // (a 16-slot ring -- loads 12 pairs ahead -- was measured on the small, latency-bound levels: slower, 13.4 vs 12.1 us)
    hipLaunchKernelGGL((k_lift2d_inv<T, ID, 8>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)nbatch), dim3(64), 0, st, a);
    return hipGetLastError();
}

template <typename T, int ID>
static hipError_t launch_lift2d_fwd(hipStream_t st, Lift2DArgs<T> a, int cu_count, int64_t nbatch = 1)
{
    constexpr int VR = (64 - 2 * kLift2dML) * 4;
    a.nstrips = (int)((a.n0 + VR - 1) / VR);
    const int64_t h1 = a.n1 >> 1;
    // chunk length: these kernels are latency-bound per wave (8-slot ring: loads run three or four steps ahead), so what counts is
    // how many waves are resident -- 40 pairs per chunk puts 3.7 waves on every SIMD for 8192^2 (64: 2.3 waves, 136 -> 125 us;
    // 32: the same time with more cone columns; 24 and below: a second round of workgroups)
    int TP = 40;
    if ((int64_t)a.nstrips * ((h1 + TP - 1) / TP) * nbatch < (int64_t)cu_count * 3) {
        TP = 64;
        while (TP > 8 && (int64_t)a.nstrips * ((h1 + TP - 1) / TP) * nbatch < (int64_t)cu_count * 8) TP >>= 1;
    }
    const int tpo = (int)opt("WL_LIFT_TP", 0);
    if (tpo >= 8 && (tpo % 8) == 0) TP = tpo;
    a.TP = TP;
    a.nchunks = (int)((h1 + TP - 1) / TP);
    hipLaunchKernelGGL((k_lift2d_fwd<T, ID, 8>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)nbatch), dim3(64), 0, st, a);
    return hipGetLastError();
}

template <typename T, int ID, int FW, int RPL>
static hipError_t launch_lift_axis_r(hipStream_t st, LiftAxisArgs<T> a, int64_t batch, int cu_count)
{
    a.nstrips = (int)((a.R + 64 * RPL - 1) / (64 * RPL));
    const int64_t half = a.C >> 1;
    int TP = 64;
    while (TP > 8 && (int64_t)a.nstrips * ((half + TP - 1) / TP) * batch < (int64_t)cu_count * 8) TP >>= 1;
    const int tpo = (int)opt("WL_LIFT_TP", 0);          // test knob: force the chunk length (multiple of 8)
    if (tpo >= 8 && (tpo % 8) == 0) TP = tpo;
    a.TP = TP;
    a.nchunks = (int)((half + TP - 1) / TP);
    for (int64_t b0 = 0; b0 < batch; b0 += 32768) {
        const int64_t nb = (batch - b0 < 32768) ? (batch - b0) : 32768;
        LiftAxisArgs<T> b = a;
        b.src = a.src + b0 * a.bs_src; b.dst = a.dst + b0 * a.bs_dst;
        hipLaunchKernelGGL((k_lift_axis_stream<T, ID, FW, RPL>), dim3((unsigned)(a.nstrips * a.nchunks), (unsigned)nb), dim3(64), 0, st, b);
    }
    return hipGetLastError();
}
// 16 bytes of rows per lane when the geometry allows it, single rows otherwise (tiny blocks)
template <typename T, int ID, int FW>
static hipError_t launch_lift_axis(hipStream_t st, const LiftAxisArgs<T> &a, int64_t batch, int cu_count)
{
    constexpr int VEC = 16 / sizeof(T);
    const bool vec = (a.R % VEC) == 0 && (a.lds % VEC) == 0 && (a.ldd % VEC) == 0 && (a.bs_src % VEC) == 0 && (a.bs_dst % VEC) == 0 &&
                     al16(a.src) && al16(a.dst);
    if (vec) return launch_lift_axis_r<T, ID, FW, VEC>(st, a, batch, cu_count);
    return launch_lift_axis_r<T, ID, FW, 1>(st, a, batch, cu_count);
}
template <typename T, int FWV>
static hipError_t launch_lift_axis_id(int id, hipStream_t st, const LiftAxisArgs<T> &a, int64_t batch, int cu_count)
{
    switch (id) {
    case 0: return launch_lift_axis<T, 0, FWV>(st, a, batch, cu_count);
    case 1: return launch_lift_axis<T, 1, FWV>(st, a, batch, cu_count);
    case 2: return launch_lift_axis<T, 2, FWV>(st, a, batch, cu_count);
    case 3: return launch_lift_axis<T, 3, FWV>(st, a, batch, cu_count);
    case 4: return launch_lift_axis<T, 4, FWV>(st, a, batch, cu_count);
    default: return launch_lift_axis<T, 5, FWV>(st, a, batch, cu_count);
    }
}

// --------------------------------------------------------------------------------------------------
// One lifting level of SHORT contiguous lines (n <= 512: the dim-1 pass of small 2-D blocks and of 3-D cubes):
// G = n/(2*PPL) lanes hold one whole line (PPL (s, d) pairs per lane), 64/G lines per wave; a step's operands
// outside the lane come from the neighbouring lane OF THE SAME GROUP by ds_bpermute with true periodic wrap, so
// there is no overlap and no boundary recomputation.  Lines are addressed as a 2-D grid (i2 < c2, i3 < c3).
template <typename T>
struct LiftShortArgs {
    const T *a; int64_t a2, a3;      // fw: src          inv: approximation source
    const T *b; int64_t b2, b3;      // fw: unused       inv: detail source
    T *o0; int64_t o02, o03;         // fw: s dest       inv: dst
    T *o1; int64_t o12, o13;         // fw: d dest       inv: unused
    // low-low corner (i2 < l2, i3 < l3): fw: s goes to ll instead of o0; inv: the approximation comes from ll
    T *ll; int64_t ll2, ll3; int l2; int64_t l3;
    int n, G, c2;
    int64_t nlines;
    T c[LIFT_FAST_STEPS][WL_MAX_NCOEF];
    T norm1, norm2;
};

template <typename T, int ID, int FW, int PPL>
__global__ void __launch_bounds__(256) k_lift_short_lines(LiftShortArgs<T> a)
{
    typedef Shape<ID> SH;
    const int lane = threadIdx.x & 63;
    const int G = a.G, lpw = 64 / G;
    const int g = lane / G, r = lane - g * G;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t li = wave * lpw + g;
    const bool valid = li < a.nlines;
    const int64_t lc = valid ? li : 0;
    const int64_t i3 = lc / a.c2, i2 = lc - i3 * a.c2;
    const int64_t half = a.n >> 1;
    const bool corner = a.ll != nullptr && i2 < a.l2 && i3 < a.l3;
    T s[PPL], d[PPL];
    if (FW) {
        T v[2 * PPL];
        ldv_l<T, 2 * PPL>(a.a + i2 * a.a2 + i3 * a.a3 + 2 * PPL * r, v);
#pragma unroll
        for (int j = 0; j < PPL; ++j) { s[j] = v[2 * j]; d[j] = v[2 * j + 1]; }                 // Util.split!
    } else {
        T sv[PPL], dv[PPL];
        ldv_l<T, PPL>((corner ? a.ll + i2 * a.ll2 + i3 * a.ll3 : a.a + i2 * a.a2 + i3 * a.a3) + PPL * r, sv);
        ldv_l<T, PPL>(a.b + i2 * a.b2 + i3 * a.b3 + PPL * r, dv);
#pragma unroll
        for (int j = 0; j < PPL; ++j) { s[j] = a.norm1 * sv[j]; d[j] = a.norm2 * dv[j]; }       // normalize! (inverse first)
    }
#pragma unroll
    for (int st = 0; st < SH::NS; ++st) {
        const int upd = SH::S[st].upd, nc = SH::S[st].nc, sh = SH::S[st].sh;
        T res[PPL];
#pragma unroll
        for (int jj = 0; jj < PPL; ++jj) {
            T o[3] = {(T)0, (T)0, (T)0};
#pragma unroll
            for (int kk = 0; kk < 3; ++kk) {
                if (kk < nc) {
                    const int off = jj + kk - sh;
                    const int delta = l_floordiv(off, PPL), e = off - delta * PPL;
                    const T v = upd ? s[e] : d[e];
                    if (delta == 0) o[kk] = v;
                    else {
                        int rs = r + delta;                     // |delta| <= 3 (PPL >= 1, |off| <= 3): wrap by repeated add
                        while (rs < 0) rs += G;
                        while (rs >= G) rs -= G;
                        o[kk] = __shfl(v, g * G + rs, 64);
                    }
                }
            }
            const int64_t jg = (int64_t)PPL * r + jj - sh;
            const bool inb = (jg >= 0) && (jg + nc - 1 <= half - 1);
            const T x = upd ? d[jj] : s[jj];
            T acc = a.c[st][0] * o[0];
            if (nc > 1) acc = acc + a.c[st][1] * o[1];
            if (nc > 2) acc = acc + a.c[st][2] * o[2];
            const T xin = x + acc;
            T xb = x + a.c[st][0] * o[0];
            if (nc > 1) xb = xb + a.c[st][1] * o[1];
            if (nc > 2) xb = xb + a.c[st][2] * o[2];
            res[jj] = inb ? xin : xb;
        }
#pragma unroll
        for (int jj = 0; jj < PPL; ++jj) { if (upd) d[jj] = res[jj]; else s[jj] = res[jj]; }
    }
    if (FW) {
        T so[PPL], dO[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) { so[j] = s[j] * a.norm1; dO[j] = d[j] * a.norm2; }        // normalize!
        if (valid) {
            stv_l<T, PPL>((corner ? a.ll + i2 * a.ll2 + i3 * a.ll3 : a.o0 + i2 * a.o02 + i3 * a.o03) + PPL * r, so);
            stv_l<T, PPL>(a.o1 + i2 * a.o12 + i3 * a.o13 + PPL * r, dO);
        }
    } else {
        T v[2 * PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) { v[2 * j] = s[j]; v[2 * j + 1] = d[j]; }                  // Util.merge!
        if (valid) stv_l<T, 2 * PPL>(a.o0 + i2 * a.o02 + i3 * a.o03 + 2 * PPL * r, v);
    }
}

// n in {2, 4} (one lane per line) or {8, 16, ..., 512}
static inline bool short_lift_ok(int64_t n) { return n == 2 || n == 4 || (n >= 8 && n <= 512 && (n & (n - 1)) == 0); }

template <typename T, int ID, int FW>
static hipError_t launch_lift_short(hipStream_t st, LiftShortArgs<T> a, int n, int c2, int64_t c3)
{
    a.n = n; a.c2 = c2; a.nlines = (int64_t)c2 * c3;
    if (a.nlines <= 0) return hipSuccess;
    const int ppl = (n >= 8) ? 4 : n / 2;
    a.G = (n / 2) / ppl;
    const int lpw = 64 / a.G;
    const int64_t nwaves = (a.nlines + lpw - 1) / lpw;
    const dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
    if (ppl == 4) hipLaunchKernelGGL((k_lift_short_lines<T, ID, FW, 4>), grid, block, 0, st, a);
    else if (ppl == 2) hipLaunchKernelGGL((k_lift_short_lines<T, ID, FW, 2>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_lift_short_lines<T, ID, FW, 1>), grid, block, 0, st, a);
    return hipGetLastError();
}
template <typename T, int FWV>
static hipError_t launch_lift_short_id(int id, hipStream_t st, const LiftShortArgs<T> &a, int n, int c2, int64_t c3)
{
    switch (id) {
    case 0: return launch_lift_short<T, 0, FWV>(st, a, n, c2, c3);
    case 1: return launch_lift_short<T, 1, FWV>(st, a, n, c2, c3);
    case 2: return launch_lift_short<T, 2, FWV>(st, a, n, c2, c3);
    case 3: return launch_lift_short<T, 3, FWV>(st, a, n, c2, c3);
    case 4: return launch_lift_short<T, 4, FWV>(st, a, n, c2, c3);
    default: return launch_lift_short<T, 5, FWV>(st, a, n, c2, c3);
    }
}

template <typename T, int FWV>
static void launch_lines_id(int id, hipStream_t st, const Lift1DArgs<T> &a, int64_t nlines, int cu_count)
{
    switch (id) {
    case 0: launch_stream_id<T, 0, FWV>(st, a, nlines, cu_count); break;
    case 1: launch_stream_id<T, 1, FWV>(st, a, nlines, cu_count); break;
    case 2: launch_stream_id<T, 2, FWV>(st, a, nlines, cu_count); break;
    case 3: launch_stream_id<T, 3, FWV>(st, a, nlines, cu_count); break;
    case 4: launch_stream_id<T, 4, FWV>(st, a, nlines, cu_count); break;
    default: launch_stream_id<T, 5, FWV>(st, a, nlines, cu_count); break;
    }
}

// 2-D (square) lifting transform: per level the dim-2 pass streams along the strided axis (k_lift_axis_stream) and
// the dim-1 pass runs on contiguous lines (k_lift1d_stream from 512 rows, k_lift_short_lines below); levels whose
// size fits neither (not a power of two below 512) use the generic kernels.
template <typename T>
int lifting_2d_fast(void *ws, int cu_count, hipStream_t st, int64_t n0, int64_t ldy, T *y, const T *x,
                    const LiftScheme<T> &sc, int L, int fw, int *handled, const char **kernel_name, int *hip_err)
{
    *handled = 0;
    constexpr int VEC = 16 / sizeof(T);
    const int id = match_shape<T>(sc);
    if (id < 0 || L < 1 || n0 < 2) return WL_OK;
    // the streaming / line kernels use 16-byte accesses; the tile and tail kernels take any leading dimension
    const bool aligned = (ldy % VEC) == 0 && al16(x) && al16(y);
#define WL_E(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) { if (hip_err) *hip_err = (int)e__; return WL_EHIP; } } while (0)
#define WL_EL() WL_E(hipGetLastError())
    const int64_t N = n0 * n0;
    Work<T> w = carve<T>(ws, N);
    Lift1DArgs<T> a;
    LiftAxisArgs<T> ax;
    LiftShortArgs<T> sa;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) {
            a.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
            ax.c[i][k] = a.c[i][k];
            sa.c[i][k] = a.c[i][k];
        }
    a.norm1 = ax.norm1 = sa.norm1 = sc.norm1;
    a.norm2 = ax.norm2 = sa.norm2 = sc.norm2;
    Strides3 full = {{1, ldy, ldy * n0}};
    auto lines_ok = [](int64_t n) { return n >= 512 && (n % 64) == 0; };
    auto fused_ok = [](int64_t n) { return n >= 128 && (n % 8) == 0; };     // k_lift2d_*: a lane's 4 rows wrap at most once
    bool any_fast = false, fused = false, gtile = false, tiled = false, ldstail = false;

    if (fw) {
        const T *cur = x;
        int64_t cur_ls = ldy;
        int pp = 0;
        for (int l = 1; l <= L; ++l) {
            const int64_t n = n0 >> (l - 1), h = n >> 1;
            const bool last = (l == L);
            T *llbuf = pp ? w.B : w.A;
            T *lld = last ? y : llbuf;
            const int64_t ldd = last ? ldy : h;
            // every remaining level of a block <= 128 x 128 (Float32) in one workgroup's LDS, a thread per line (k_tail_lift2d_lds)
            // (measured r04, cdf9/7 forward: 64^2 all levels 11.2 us against 13.5 in the one-wave register tail; 128^2 24.4 against
            //  23.3 for a tile launch + the register tail -- a 128-sample line per thread keeps two waves busy, the tile kernel 16)
            if ((id == 0 || id == 2 || id == 4) && tail_lift2d_lds_ok<T>(id, n) && n >= l_env("WL_LIFT_LDSTAIL2D_FMIN", 64) &&
                n <= l_env("WL_LIFT_LDSTAIL2D_FMAX", 64) && l_env("WL_NO_LIFT_TAIL2D", 0) == 0 && l_env("WL_LIFT_LDSTAIL2D", 1) != 0) {
                WL_E((launch_tail_lift2d_lds<T, 1>(id, st, sc, cur, cur_ls, y, ldy, (int)n, L - l + 1)));
                any_fast = true; ldstail = (l == 1);          // (names the call only when the tail is the whole transform)
                break;
            }
            if (n <= 64 && l_env("WL_NO_LIFT_TAIL2D", 0) == 0) {        // every remaining level inside one workgroup / one wave
                if ((id == 0 || id == 2 || id == 4) && tail_lift2d_reg_ok<T>(id, (int)n) && l_env("WL_LIFT_REGTAIL2D", 1) != 0)
                    WL_E((launch_tail_lift2d_reg<T, 1>(id, st, sc, cur, cur_ls, y, ldy, (int)n, L - l + 1)));
                else
                    WL_E((launch_tail_lift2d<T, 1>(st, sc, cur, cur_ls, y, ldy, (int)n, L - l + 1)));
                any_fast = true;
                break;
            }
            // (round 5) ... two levels per launch where two are left, Float32, up to 1024 rows: the level-l approximation stays in LDS
            // (k_lift2d_tile2_fwd).  cdf9/7 full depth: 1024^2 42.5 -> 34.0 us, 2048^2 58.2 -> 50.2, 4096^2 99.5 -> 92.8, 8192^2 218.4 -> 216.4.
            // Not from 2048 rows (1024 tiles of 512 threads with a 1.9 x halo: 58 -> 65 us) and not for Float64 (79 KB of LDS: 51.5 -> 62.7).
            if (aligned && (sizeof(T) == 4 || l_env("WL_LIFT_TILE2_F64", 0) != 0) && l_env("WL_LIFT_TILE", 1) != 0 && l_env("WL_LIFT_TILE2", 1) != 0 && (L - l + 1) >= 2 && n <= l_env("WL_LIFT_TILE2_MAX", 1024) &&
                lift2d_tile2_ok(id, n) && (cur_ls % VEC) == 0 && al16(cur) && al16(llbuf) && cur != y) {
                const bool last2 = (l + 1 == L);
                WL_E((lift2d_tile2_fwd_launch<T>(id, st, sc, cur, cur_ls, y, ldy, last2 ? (T *)nullptr : llbuf, n >> 2, n)));
                any_fast = true; tiled = true;
                cur = llbuf; cur_ls = n >> 2; pp ^= 1;
                ++l;
                continue;
            }
            // cache-resident levels: 64 x 64 tiles, one launch per level without the marching kernels' latency chain
            if (aligned && l_env("WL_LIFT_TILE", 1) != 0 && n <= l_env("WL_LIFT_TILE_MAX", 2048) && lift2d_tile_ok(id, n) && (id == 0 || id == 2 || id == 4) &&
                (cur_ls % VEC) == 0 && al16(cur) && al16(llbuf) && cur != y) {
                WL_E((lift2d_tile_launch<T>(id, 1, st, sc, cur, cur_ls, y, ldy, last ? (T *)nullptr : llbuf, h, n)));
                any_fast = true; tiled = true;
                cur = llbuf; cur_ls = h; pp ^= 1;
                continue;
            }
            if (aligned && fused_ok(n) && n > l_env("WL_LIFT_GTILE_MAX", 0) && (id == 0 || id == 2 || id == 4) && l_env("WL_NO_LIFT2D_FUSED", 0) == 0 && (cur_ls % VEC) == 0 &&
                al16(cur) && al16(llbuf) && cur != y) {      // (in place, level 1 reads y while writing it: two passes via T0)
                // both passes of the level in one kernel: read the block once, write the four quadrants once
                Lift2DArgs<T> q2;
                for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                    for (int k = 0; k < WL_MAX_NCOEF; ++k) q2.c[i][k] = a.c[i][k];
                q2.norm1 = a.norm1; q2.norm2 = a.norm2;
                q2.src = cur; q2.lds = cur_ls; q2.y = y; q2.ldy = ldy; q2.ll = last ? (T *)nullptr : llbuf; q2.ldl = h; q2.n0 = n; q2.n1 = n;
                q2.bs_src = q2.bs_y = q2.bs_ll = 0; q2.nll = 1;
                if (id == 0) WL_E((launch_lift2d_fwd<T, 0>(st, q2, cu_count)));
                else if (id == 2) WL_E((launch_lift2d_fwd<T, 2>(st, q2, cu_count)));
                else WL_E((launch_lift2d_fwd<T, 4>(st, q2, cu_count)));
                any_fast = true; fused = true;
                cur = llbuf; cur_ls = h; pp ^= 1;
                continue;
            }
            if (aligned && n > l_env("WL_LIFT_GTILE_MAX", 0) && (lines_ok(n) || short_lift_ok(n))) {
                any_fast = true;
                // rows (dim 2): one streaming pass along the strided axis, T0 = [s-columns | d-columns]
                ax.src = cur; ax.lds = cur_ls; ax.bs_src = 0; ax.dst = w.T0; ax.ldd = n; ax.bs_dst = 0; ax.R = n; ax.C = n;
                WL_E((launch_lift_axis_id<T, 1>(id, st, ax, 1, cu_count)));
                if (lines_ok(n)) {
                    a.b = nullptr; a.b_ls = 0; a.n = n; a.ntiles = (h + 247) / 248;
                    // columns [0, h): s -> LL, d -> y[h.., j]
                    a.a = w.T0; a.a_ls = n; a.o0 = lld; a.o0_ls = ldd; a.o1 = y + h; a.o1_ls = ldy;
                    launch_lines_id<T, 1>(id, st, a, h, cu_count); WL_EL();
                    // columns [h, n): s -> y[0..h, j], d -> y[h.., j]
                    a.a = w.T0 + h * n; a.o0 = y + h * ldy; a.o0_ls = ldy; a.o1 = y + h * ldy + h; a.o1_ls = ldy;
                    launch_lines_id<T, 1>(id, st, a, h, cu_count); WL_EL();
                } else {
                    // every column of T0 in one launch; columns [0, h) send their approximation to the next level's buffer
                    sa.b = nullptr; sa.b2 = sa.b3 = 0; sa.a3 = sa.o03 = sa.o13 = sa.ll3 = 0;
                    sa.a = w.T0; sa.a2 = n; sa.o0 = y; sa.o02 = ldy; sa.o1 = y + h; sa.o12 = ldy;
                    sa.ll = last ? (T *)nullptr : llbuf; sa.ll2 = h; sa.l2 = (int)h; sa.l3 = 1;
                    WL_E((launch_lift_short_id<T, 1>(id, st, sa, (int)n, (int)n, 1)));
                }
            } else if ((id == 0 || id == 2 || id == 4) && lift2d_gtile_ok<T>(id, n) && cur != y && l_env("WL_LIFT_GTILE", 1) != 0) {
                // any even size: one tile launch instead of twelve one-thread-per-element launches
                WL_E((launch_lift2d_gtile<T, 1>(id, st, sc, cur, cur_ls, y, ldy, last ? (T *)nullptr : llbuf, h, n)));
                any_fast = true; gtile = true;
            } else {
                Extent3 ext = {{n, n, 1}}, lo = {{h, h, 1}};
                Strides3 box = {{1, n, n * n}}, cst = {{1, cur_ls, cur_ls * n}}, lst = {{1, ldd, ldd * h}};
                // rows (axis 1) then columns (axis 0), as generic_lifting_fwd
                WL_E(generic_lift_split<T>(st, cur, cst, w.W, box, ext, 1));
                for (int q = 0; q < sc.nsteps; ++q) WL_E(generic_lift_step<T>(st, sc.step[q], w.W, box, ext, 1));
                WL_E(generic_lift_finish_fwd<T>(st, sc.norm1, sc.norm2, w.W, box, w.T0, box, (T *)nullptr, box, ext, 1, lo));
                WL_E(generic_lift_split<T>(st, w.T0, box, w.W, box, ext, 0));
                for (int q = 0; q < sc.nsteps; ++q) WL_E(generic_lift_step<T>(st, sc.step[q], w.W, box, ext, 0));
                WL_E(generic_lift_finish_fwd<T>(st, sc.norm1, sc.norm2, w.W, box, y, full, last ? (T *)nullptr : llbuf, lst, ext, 0, lo));
            }
            cur = llbuf; cur_ls = h; pp ^= 1;
        }
    } else {
        const T *llsrc = nullptr;
        int64_t ll_ls = 0;
        int pp = 0;
        int l_start = L;
        if (l_env("WL_NO_LIFT_TAIL2D", 0) == 0) {
            int l_lo = L + 1;                       // shallowest level whose output still fits the LDS tail
            while (l_lo > 1 && (n0 >> (l_lo - 2)) <= 64) --l_lo;
            // ... or, a thread per line in several waves, 128 x 128 (Float32): k_tail_lift2d_lds (inverse 128^2 all levels: 19.1 us
            //     against 20.9 for the register tail + a tile launch)
            const bool lds128 = (id == 1 || id == 3 || id == 5) && l_env("WL_LIFT_LDSTAIL2D", 1) != 0 && l_lo >= 2 && l_lo <= L + 1 &&
                                tail_lift2d_lds_ok<T>(id, n0 >> (l_lo - 2)) && (n0 >> (l_lo - 2)) >= l_env("WL_LIFT_LDSTAIL2D_MIN", 128);
            if (lds128) --l_lo;
            if (l_lo <= L) {
                const int64_t n = n0 >> (l_lo - 1);
                T *out = (l_lo == 1) ? y : (pp ? w.B : w.A);
                const int64_t ldo = (l_lo == 1) ? ldy : n;
                if (lds128) {
                    WL_E((launch_tail_lift2d_lds<T, 0>(id, st, sc, x, ldy, out, ldo, (int)n, L - l_lo + 1)));
                    ldstail = (l_lo == 1);
                } else if ((id == 1 || id == 3 || id == 5) && tail_lift2d_reg_ok<T>(id, (int)n) && l_env("WL_LIFT_REGTAIL2D", 1) != 0)
                    WL_E((launch_tail_lift2d_reg<T, 0>(id, st, sc, x, ldy, out, ldo, (int)n, L - l_lo + 1)));
                else
                    WL_E((launch_tail_lift2d<T, 0>(st, sc, x, ldy, out, ldo, (int)n, L - l_lo + 1)));
                any_fast = true;
                llsrc = out; ll_ls = ldo; pp ^= 1;
                l_start = l_lo - 1;
            }
        }
        for (int l = l_start; l >= 1; --l) {
            const int64_t n = n0 >> (l - 1), h = n >> 1;
            T *out = (l == 1) ? y : (pp ? w.B : w.A);
            const int64_t ldo = (l == 1) ? ldy : n;
            // (round 5) two reconstruction levels per launch (output 2 n <= 1024 rows, Float32): the coarser level's result is produced in
            // LDS where the finer level's staging expects its approximation quadrant (k_lift2d_tile2_inv)
            if (aligned && (sizeof(T) == 4 || l_env("WL_LIFT_TILE2_F64", 0) != 0) && l_env("WL_LIFT_TILE", 1) != 0 && l_env("WL_LIFT_TILE2", 1) != 0 && l >= 2 &&
                2 * n <= l_env("WL_LIFT_TILE2_MAX", 1024) && lift2d_tile2_inv_ok(id, 2 * n) && (!llsrc || (al16(llsrc) && (ll_ls % 2) == 0))) {
                const int64_t nf = 2 * n;
                T *out2 = (l - 1 == 1) ? y : (pp ? w.B : w.A);
                const int64_t ldo2 = (l - 1 == 1) ? ldy : nf;
                if (out2 != x) {
                    WL_E((lift2d_tile2_inv_launch<T>(id, st, sc, x, ldy, out2, ldo2, llsrc, ll_ls, nf)));
                    any_fast = true; tiled = true;
                    llsrc = out2; ll_ls = ldo2; pp ^= 1;
                    --l;
                    continue;
                }
            }
            if (aligned && l_env("WL_LIFT_TILE", 1) != 0 && n <= l_env("WL_LIFT_TILE_MAX", 2048) && lift2d_tile_ok(id, n) && (id == 1 || id == 3 || id == 5) &&
                (!llsrc || (al16(llsrc) && (ll_ls % 2) == 0)) && out != x) {
                WL_E((lift2d_tile_launch<T>(id, 0, st, sc, x, ldy, out, ldo, const_cast<T *>(llsrc), ll_ls, n)));
                any_fast = true; tiled = true;
                llsrc = out; ll_ls = ldo; pp ^= 1;
                continue;
            }
            if (aligned && fused_ok(n) && n > l_env("WL_LIFT_GTILE_MAX", 0) && (id == 1 || id == 3 || id == 5) && l_env("WL_NO_LIFT2D_FUSED", 0) == 0 && (ldo % VEC) == 0 && al16(out) &&
                (!llsrc || (al16(llsrc) && (ll_ls % 2) == 0)) && out != x) {   // (in place, level 1 writes y while reading it)
                Lift2DArgs<T> q2;
                for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                    for (int k = 0; k < WL_MAX_NCOEF; ++k) q2.c[i][k] = a.c[i][k];
                q2.norm1 = a.norm1; q2.norm2 = a.norm2;
                q2.src = x; q2.lds = ldy; q2.y = out; q2.ldy = ldo; q2.ll = const_cast<T *>(llsrc); q2.ldl = ll_ls; q2.n0 = n; q2.n1 = n;
                q2.bs_src = q2.bs_y = q2.bs_ll = 0; q2.nll = 1;
                if (id == 1) WL_E((launch_lift2d_inv<T, 1>(st, q2, cu_count)));
                else if (id == 3) WL_E((launch_lift2d_inv<T, 3>(st, q2, cu_count)));
                else WL_E((launch_lift2d_inv<T, 5>(st, q2, cu_count)));
                any_fast = true; fused = true;
                llsrc = out; ll_ls = ldo; pp ^= 1;
                continue;
            }
            if (aligned && n > l_env("WL_LIFT_GTILE_MAX", 0) && (lines_ok(n) || short_lift_ok(n))) {
                any_fast = true;
                // columns: merged column j -> T0[:, j]
                if (lines_ok(n)) {
                    a.o0 = w.T0; a.o0_ls = n; a.o1 = nullptr; a.o1_ls = 0; a.n = n; a.ntiles = (h + 247) / 248;
                    if (llsrc) { a.a = llsrc; a.a_ls = ll_ls; } else { a.a = x; a.a_ls = ldy; }
                    a.b = x + h; a.b_ls = ldy;
                    launch_lines_id<T, 0>(id, st, a, h, cu_count); WL_EL();
                    a.a = x + h * ldy; a.a_ls = ldy; a.b = x + h * ldy + h; a.b_ls = ldy; a.o0 = w.T0 + h * n;
                    launch_lines_id<T, 0>(id, st, a, h, cu_count); WL_EL();
                } else {
                    sa.o1 = nullptr; sa.o12 = sa.o13 = 0; sa.a3 = sa.b3 = sa.o03 = sa.ll3 = 0;
                    sa.o0 = w.T0; sa.o02 = n; sa.a = x; sa.a2 = ldy; sa.b = x + h; sa.b2 = ldy;
                    sa.ll = const_cast<T *>(llsrc); sa.ll2 = ll_ls; sa.l2 = (int)h; sa.l3 = 1;
                    WL_E((launch_lift_short_id<T, 0>(id, st, sa, (int)n, (int)n, 1)));
                }
                // rows (dim 2): streaming pass along the strided axis straight into the result
                ax.src = w.T0; ax.lds = n; ax.bs_src = 0; ax.dst = out; ax.ldd = ldo; ax.bs_dst = 0; ax.R = n; ax.C = n;
                WL_E((launch_lift_axis_id<T, 0>(id, st, ax, 1, cu_count)));
            } else if ((id == 1 || id == 3 || id == 5) && lift2d_gtile_ok<T>(id, n) && out != x && l_env("WL_LIFT_GTILE", 1) != 0) {
                WL_E((launch_lift2d_gtile<T, 0>(id, st, sc, x, ldy, out, ldo, const_cast<T *>(llsrc), ll_ls, n)));
                any_fast = true; gtile = true;
            } else {
                Extent3 ext = {{n, n, 1}}, lo = {{h, h, 1}};
                Strides3 box = {{1, n, n * n}}, lst = {{1, ll_ls, ll_ls * h}}, ost = {{1, ldo, ldo * n}};
                WL_E(generic_lift_norm_inv<T>(st, sc.norm1, sc.norm2, x, full, llsrc, lst, w.W, box, ext, 0, lo));
                for (int q = 0; q < sc.nsteps; ++q) WL_E(generic_lift_step<T>(st, sc.step[q], w.W, box, ext, 0));
                WL_E(generic_lift_merge<T>(st, w.W, box, w.T0, box, ext, 0));
                WL_E(generic_lift_norm_inv<T>(st, sc.norm1, sc.norm2, w.T0, box, (const T *)nullptr, box, w.W, box, ext, 1, lo));
                for (int q = 0; q < sc.nsteps; ++q) WL_E(generic_lift_step<T>(st, sc.step[q], w.W, box, ext, 1));
                WL_E(generic_lift_merge<T>(st, w.W, box, out, ost, ext, 1));
            }
            llsrc = out; ll_ls = ldo; pp ^= 1;
        }
    }
    *handled = 1;
    if (kernel_name)
        *kernel_name = any_fast ? (fused ? (fw ? "k_lift2d_fwd" : "k_lift2d_inv") : (tiled ? "k_lift2d_tile" : (gtile ? "k_lift2d_gtile" : (ldstail ? "k_tail_lift2d_lds" : "k_lift_axis_stream+lines"))))
                                : (fw ? "k_generic_lift_fwd" : "k_generic_lift_inv");
    return WL_OK;
}

// 3-D (cube) lifting transform, cubes of 2^k <= 512 per side: planes and rows along the strided axes
// (k_lift_axis_stream, rows batched over the planes), columns as short lines with the LLL corner routed to the
// approximation buffer.  Other sizes: not handled (generic kernels).
template <typename T>
int lifting_3d_fast(void *ws, int cu_count, hipStream_t st, int64_t n0, T *y, const T *x,
                    const LiftScheme<T> &sc, int L, int fw, int *handled, const char **kernel_name, int *hip_err)
{
    *handled = 0;
    const int id = match_shape<T>(sc);
    if (id < 0 || L < 1 || n0 < 8 || n0 > 512 || (n0 & (n0 - 1)) != 0 || !al16(x) || !al16(y)) return WL_OK;
    if ((n0 >> (L - 1)) < 2) return WL_OK;
    const int64_t N = n0 * n0 * n0;
    Work<T> w = carve<T>(ws, N);
    LiftAxisArgs<T> ax;
    LiftShortArgs<T> sa;
    for (int i = 0; i < LIFT_FAST_STEPS; ++i)
        for (int k = 0; k < WL_MAX_NCOEF; ++k) {
            ax.c[i][k] = (i < sc.nsteps) ? sc.step[i].c[k] : (T)0;
            sa.c[i][k] = ax.c[i][k];
        }
    ax.norm1 = sa.norm1 = sc.norm1;
    ax.norm2 = sa.norm2 = sc.norm2;
    const int64_t y1 = n0, y2 = n0 * n0;
    if (fw) {
        const T *cur = x;
        int64_t c1 = n0, c2 = n0 * n0;
        int pp = 0;
        for (int l = 1; l <= L; ++l) {
            const int64_t n = n0 >> (l - 1), h = n >> 1;
            const bool last = (l == L);
            T *llbuf = pp ? w.B : w.A;
            // every remaining level inside one workgroup's LDS (k_tail_lift3d)
            if ((id == 0 || id == 2 || id == 4) && tail_lift3d_ok<T>(id, n) && l_env("WL_LIFT_TAIL3D", 1) != 0) {
                WL_E((launch_tail_lift3d<T, 1>(id, st, sc, cur, c1, c2, y, y1, y2, (int)n, L - l + 1)));
                break;
            }
            // planes (dim 3): the cube is an (n*n) x n matrix when its first two dims are dense
            if (c1 == n) {
                ax.src = cur; ax.lds = c2; ax.bs_src = 0; ax.dst = w.T0; ax.ldd = n * n; ax.bs_dst = 0; ax.R = n * n; ax.C = n;
                WL_E((launch_lift_axis_id<T, 1>(id, st, ax, 1, cu_count)));
            } else {       // (not reached: level 1 reads the dense cube, deeper levels the dense approximation buffer)
                return WL_OK;
            }
            if (n >= 128 && (id == 0 || id == 2 || id == 4) && l_env("WL_NO_LIFT2D_FUSED", 0) == 0) {
                // rows + columns of every plane in one launch (fused 2-D level kernel batched over the planes)
                Lift2DArgs<T> q2;
                for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                    for (int k = 0; k < WL_MAX_NCOEF; ++k) q2.c[i][k] = ax.c[i][k];
                q2.norm1 = ax.norm1; q2.norm2 = ax.norm2;
                q2.src = w.T0; q2.lds = n; q2.y = y; q2.ldy = y1; q2.ll = last ? (T *)nullptr : llbuf; q2.ldl = h; q2.n0 = n; q2.n1 = n;
                q2.bs_src = n * n; q2.bs_y = y2; q2.bs_ll = h * h; q2.nll = (int)h;
                if (id == 0) WL_E((launch_lift2d_fwd<T, 0>(st, q2, cu_count, n)));
                else if (id == 2) WL_E((launch_lift2d_fwd<T, 2>(st, q2, cu_count, n)));
                else WL_E((launch_lift2d_fwd<T, 4>(st, q2, cu_count, n)));
                cur = llbuf; c1 = h; c2 = h * h; pp ^= 1;
                continue;
            }
            // rows (dim 2): n matrices of n x n
            ax.src = w.T0; ax.lds = n; ax.bs_src = n * n; ax.dst = w.T1; ax.ldd = n; ax.bs_dst = n * n; ax.R = n; ax.C = n;
            WL_E((launch_lift_axis_id<T, 1>(id, st, ax, n, cu_count)));
            // columns (dim 1): the low-low corner sends its approximation on to the next level's buffer
            sa.b = nullptr; sa.b2 = sa.b3 = 0;
            sa.a = w.T1; sa.a2 = n; sa.a3 = n * n;
            sa.o0 = y; sa.o02 = y1; sa.o03 = y2; sa.o1 = y + h; sa.o12 = y1; sa.o13 = y2;
            sa.ll = last ? (T *)nullptr : llbuf; sa.ll2 = h; sa.ll3 = h * h; sa.l2 = (int)h; sa.l3 = h;
            WL_E((launch_lift_short_id<T, 1>(id, st, sa, (int)n, (int)n, n)));
            cur = llbuf; c1 = h; c2 = h * h; pp ^= 1;
        }
    } else {
        const T *llsrc = nullptr;
        int pp = 0;
        int l_top = L;
        // the deepest levels (outputs of <= 32^3 Float32 / 16^3 Float64) inside one workgroup's LDS (k_tail_lift3d)
        if ((id == 1 || id == 3 || id == 5) && l_env("WL_LIFT_TAIL3D", 1) != 0 && tail_lift3d_ok<T>(id, n0 >> (L - 1))) {
            int lt = L;
            while (lt > 1 && tail_lift3d_ok<T>(id, n0 >> (lt - 2))) --lt;
            const int64_t m0 = n0 >> (lt - 1);
            T *out = (lt == 1) ? y : (pp ? w.B : w.A);
            WL_E((launch_tail_lift3d<T, 0>(id, st, sc, x, y1, y2, out, (lt == 1) ? y1 : m0, (lt == 1) ? y2 : m0 * m0, (int)m0, L - lt + 1)));
            llsrc = out; pp ^= 1;
            l_top = lt - 1;
        }
        for (int l = l_top; l >= 1; --l) {
            const int64_t n = n0 >> (l - 1), h = n >> 1;
            T *out = (l == 1) ? y : (pp ? w.B : w.A);
            bool planes = false;
            if (n >= 128 && (id == 1 || id == 3 || id == 5) && l_env("WL_NO_LIFT2D_FUSED", 0) == 0) {
                // columns + rows of every plane in one launch
                Lift2DArgs<T> q2;
                for (int i = 0; i < LIFT_FAST_STEPS; ++i)
                    for (int k = 0; k < WL_MAX_NCOEF; ++k) q2.c[i][k] = ax.c[i][k];
                q2.norm1 = ax.norm1; q2.norm2 = ax.norm2;
                q2.src = x; q2.lds = y1; q2.y = w.T1; q2.ldy = n; q2.ll = const_cast<T *>(llsrc); q2.ldl = h; q2.n0 = n; q2.n1 = n;
                q2.bs_src = y2; q2.bs_y = n * n; q2.bs_ll = h * h; q2.nll = (int)h;
                if (id == 1) WL_E((launch_lift2d_inv<T, 1>(st, q2, cu_count, n)));
                else if (id == 3) WL_E((launch_lift2d_inv<T, 3>(st, q2, cu_count, n)));
                else WL_E((launch_lift2d_inv<T, 5>(st, q2, cu_count, n)));
                planes = true;
            }
            if (!planes) {
            // columns first: merged lines into T0 (dense n^3); the low-low corner reads the deeper reconstruction
            sa.o1 = nullptr; sa.o12 = sa.o13 = 0;
            sa.a = x; sa.a2 = y1; sa.a3 = y2; sa.b = x + h; sa.b2 = y1; sa.b3 = y2;
            sa.o0 = w.T0; sa.o02 = n; sa.o03 = n * n;
            sa.ll = const_cast<T *>(llsrc); sa.ll2 = h; sa.ll3 = h * h; sa.l2 = (int)h; sa.l3 = h;
            WL_E((launch_lift_short_id<T, 0>(id, st, sa, (int)n, (int)n, n)));
            // rows (dim 2)
            ax.src = w.T0; ax.lds = n; ax.bs_src = n * n; ax.dst = w.T1; ax.ldd = n; ax.bs_dst = n * n; ax.R = n; ax.C = n;
            WL_E((launch_lift_axis_id<T, 0>(id, st, ax, n, cu_count)));
            }
            // planes (dim 3); the result of level 1 goes to y (dense cube), deeper ones to the dense n^3 buffer
            ax.src = w.T1; ax.lds = n * n; ax.bs_src = 0; ax.dst = out; ax.ldd = (l == 1) ? y2 : n * n; ax.bs_dst = 0; ax.R = n * n; ax.C = n;
            WL_E((launch_lift_axis_id<T, 0>(id, st, ax, 1, cu_count)));
            llsrc = out; pp ^= 1;
        }
    }
#undef WL_E
#undef WL_EL
    *handled = 1;
    if (kernel_name) *kernel_name = "k_lift_axis_stream+k_lift_short_lines";
    return WL_OK;
}
template int lifting_3d_fast<float>(void *, int, hipStream_t, int64_t, float *, const float *, const LiftScheme<float> &, int, int, int *,
                                    const char **, int *);
template int lifting_3d_fast<double>(void *, int, hipStream_t, int64_t, double *, const double *, const LiftScheme<double> &, int, int,
                                     int *, const char **, int *);

template int lifting_2d_fast<float>(void *, int, hipStream_t, int64_t, int64_t, float *, const float *, const LiftScheme<float> &,
                                    int, int, int *, const char **, int *);
template int lifting_2d_fast<double>(void *, int, hipStream_t, int64_t, int64_t, double *, const double *,
                                     const LiftScheme<double> &, int, int, int *, const char **, int *);

}  // namespace wl
