// wl_generic.hip -- generic kernels: any box size (incl. lines shorter than the filter),
// any filter length <= WL_MAX_FLEN, any axis, Float32/Float64.
//
// These are the catch-all path: one thread per output (pair), operands read straight
// from global memory with true periodic indexing.  The fast paths (wl_fwd2d.hip,
// wl_fwd1d.hip, ...) cover the large power-of-two levels of the BASELINE configs and
// are required to be bit-identical to these kernels, which in turn are bit-identical
// to the CPU oracle.  Thread index runs fastest along dim 1 (contiguous) so global
// accesses of the strided-axis passes are coalesced.
#include "wl_internal.h"

namespace wl {

static constexpr int kBlock = 256;
static inline dim3 grid_for(int64_t total)
{
    int64_t g = (total + kBlock - 1) / kBlock;
    const int64_t cap = 256 * 64;   // 64 blocks per CU, grid-stride beyond
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return dim3((unsigned)g);
}

struct Idx3 { int64_t i[3]; };
__device__ __forceinline__ Idx3 unflatten(int64_t t, int64_t e0, int64_t e1)
{
    Idx3 r;
    r.i[0] = t % e0;
    int64_t q = t / e0;
    r.i[1] = q % e1;
    r.i[2] = q / e1;
    return r;
}
__device__ __forceinline__ bool in_low_corner(const Idx3 &c, int axis, const Extent3 &lo)
{
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d != axis) ok = ok && (c.i[d] < lo.n[d]);
    return ok;
}

// --------------------------------------------------------------------------------------
// forward filter pass: reference filtdown! x2 (transforms_filter.jl:387-433 as called at
// :73-75), closed form documented in wl_internal.h
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_fwd_filter(Taps<T> taps, const T *__restrict__ src, Strides3 sst,
                     T *__restrict__ dst, Strides3 dst_st, T *__restrict__ ll, Strides3 ll_st,
                     Extent3 n, int axis, Extent3 lo, const uint8_t *__restrict__ mask)
{
    const int F = taps.F;
    const int64_t nax = n.n[axis], nx = nax >> 1, sa = sst.s[axis];
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = nx;
    const int64_t total = e[0] * e[1] * e[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, e[0], e[1]);
        const int64_t k = c.i[axis];
        int64_t base = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) base += c.i[d] * sst.s[d];
        const T *p = src + base;
        if (mask != nullptr && !mask[c.i[1]]) {      // WPT: node not split -> copy the segment through
            int64_t o0 = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) o0 += (d == axis ? 0 : c.i[d]) * dst_st.s[d];
            dst[o0 + (2 * k) * dst_st.s[axis]] = p[(2 * k) * sa];
            dst[o0 + (2 * k + 1) * dst_st.s[axis]] = p[(2 * k + 1) * sa];
            continue;
        }
        // both branches walk the line upwards one sample per tap: one modulo for the start, then a conditional wrap
        // (64-bit % is an emulated division on the GPU; a line shorter than the filter simply wraps several times).
        // Taps are taken in blocks of 8 with the block's loads issued before its arithmetic: with one dependent load per tap
        // a pass over a 270 x 480 block took 19 us of pure memory latency.
        // scaling branch: m ascending, samples 2k, 2k+1, ...
        int64_t idx = 2 * k;                                  // < nax
        T s = taps.h[0] * p[idx * sa];
        for (int m0 = 1; m0 < F; m0 += 8) {
            T xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 + e < F) {
                    if (++idx == nax) idx = 0;
                    xv[e] = p[idx * sa];
                }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 + e < F) s = s + taps.h[m0 + e] * xv[e];
        }
        // detail branch: m descending, samples 2k+1-(F-1), ..., 2k+1
        idx = pmod(2 * k + 1 - (F - 1), nax);
        T dd = taps.g[F - 1] * p[idx * sa];
        for (int m0 = F - 2; m0 >= 0; m0 -= 8) {
            T xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 - e >= 0) {
                    if (++idx == nax) idx = 0;
                    xv[e] = p[idx * sa];
                }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m0 - e >= 0) dd = dd + taps.g[m0 - e] * xv[e];
        }

        int64_t off_s = 0, off_d = 0;
        if (ll != nullptr && in_low_corner(c, axis, lo)) {
            int64_t o = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) o += c.i[d] * ll_st.s[d];
            ll[o] = s;
        } else {
#pragma unroll
            for (int d = 0; d < 3; ++d) off_s += c.i[d] * dst_st.s[d];
            dst[off_s] = s;
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) off_d += (d == axis ? (nx + k) : c.i[d]) * dst_st.s[d];
        dst[off_d] = dd;
    }
}

// inverse filter pass: reference filtup! x2 (transforms_filter.jl:467-541 as called at :78-80)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_inv_filter(Taps<T> taps, const T *__restrict__ src, Strides3 sst,
                     const T *__restrict__ ll, Strides3 ll_st,
                     T *__restrict__ dst, Strides3 dst_st, Extent3 n, int axis, Extent3 lo,
                     const uint8_t *__restrict__ mask)
{
    const int F = taps.F;
    const int64_t nax = n.n[axis], nx = nax >> 1;
    const int64_t total = n.n[0] * n.n[1] * n.n[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, n.n[0], n.n[1]);
        const int64_t o = c.i[axis];
        int64_t base = 0, lbase = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) { base += c.i[d] * sst.s[d]; lbase += c.i[d] * ll_st.s[d]; }
        if (mask != nullptr && !mask[c.i[1]]) {
            int64_t off = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) off += c.i[d] * dst_st.s[d];
            dst[off] = src[base + o * sst.s[axis]];
            continue;
        }
        const bool use_ll = (ll != nullptr) && in_low_corner(c, axis, lo);
        const T *ps = use_ll ? (ll + lbase) : (src + base);
        const int64_t ss = use_ll ? ll_st.s[axis] : sst.s[axis];
        const T *pd = src + base + nx * sst.s[axis];
        const int64_t sd = sst.s[axis];

        // S: m descending over the taps with (o - m) even, coefficient index (o - m)/2 going up by one per term;
        // D: m ascending over the taps with (o + m - 1) even, index (o + m - 1)/2 going up by one per term.
        // One modulo for the start of S (it may be negative), conditional wraps afterwards.
        T S = (T)0, D = (T)0;
        {
            const int mt = (((F - 1 - o) & 1) == 0) ? F - 1 : F - 2;
            if (mt >= 0) {
                int64_t k = pmod((o - mt) / 2, nx);           // (o - mt) even => exact
                S = taps.h[mt] * ps[k * ss];
                for (int m0 = mt - 2; m0 >= 0; m0 -= 16) {    // blocks of 8 terms, loads before arithmetic (see the forward kernel)
                    T xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 - 2 * e >= 0) {
                            if (++k == nx) k = 0;
                            xv[e] = ps[k * ss];
                        }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 - 2 * e >= 0) S = S + taps.h[m0 - 2 * e] * xv[e];
                }
            }
        }
        {
            const int mb = (o & 1) ? 0 : 1;
            if (mb < F) {
                int64_t k = (o + mb - 1) / 2;                 // in [0, nx)
                D = taps.g[mb] * pd[k * sd];
                for (int m0 = mb + 2; m0 < F; m0 += 16) {
                    T xv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 + 2 * e < F) {
                            if (++k == nx) k = 0;
                            xv[e] = pd[k * sd];
                        }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (m0 + 2 * e < F) D = D + taps.g[m0 + 2 * e] * xv[e];
                }
            }
        }
        int64_t off = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) off += c.i[d] * dst_st.s[d];
        dst[off] = S + D;
    }
}

// --------------------------------------------------------------------------------------
// lifting building blocks
// split: Util.split! (util_main.jl:142-204): w[k] = src[2k], w[nx+k] = src[2k+1] along axis
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_lift_split(const T *__restrict__ src, Strides3 sst, T *__restrict__ w, Strides3 wst, Extent3 n, int axis,
                     const uint8_t *__restrict__ mask)
{
    const int64_t nx = n.n[axis] >> 1;
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = nx;
    const int64_t total = e[0] * e[1] * e[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, e[0], e[1]);
        const int64_t k = c.i[axis];
        int64_t sb = 0, wb = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) { sb += c.i[d] * sst.s[d]; wb += c.i[d] * wst.s[d]; }
        if (mask != nullptr && !mask[c.i[1]]) {      // copy through in natural order
            w[wb + (2 * k) * wst.s[axis]] = src[sb + (2 * k) * sst.s[axis]];
            w[wb + (2 * k + 1) * wst.s[axis]] = src[sb + (2 * k + 1) * sst.s[axis]];
            continue;
        }
        w[wb + k * wst.s[axis]] = src[sb + (2 * k) * sst.s[axis]];
        w[wb + (nx + k) * wst.s[axis]] = src[sb + (2 * k + 1) * sst.s[axis]];
    }
}

// one lifting step in place: lift! (transforms_lifting.jl:366-381) = lift_perboundary!
// (:437-451) on the wrapped indices + lift_inbounds! (:455-483) on the rest.  Element j is
// "in bounds" iff none of its nc operands wraps: 0 <= j-shift and j+nc-1-shift <= half-1
// (irlimits :383-390).  The two cases round differently for nc >= 2:
//   in bounds: x += (c1*a + c2*b [+ c3*c]);   boundary: x += c1*a; x += c2*b; ...
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_lift_step(LiftStep<T> st, T *__restrict__ w, Strides3 wst, Extent3 n, int axis,
                    const uint8_t *__restrict__ mask)
{
    const int64_t half = n.n[axis] >> 1;
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = half;
    const int64_t total = e[0] * e[1] * e[2];
    const int64_t sa = wst.s[axis];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, e[0], e[1]);
        const int64_t j = c.i[axis];
        if (mask != nullptr && !mask[c.i[1]]) continue;
        int64_t wb = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) wb += c.i[d] * wst.s[d];
        // Predict: target = s half, operands = d half.  Update: the other way round.
        T *tgt = w + wb + (st.is_update ? (half + j) : j) * sa;
        const T *opnd = w + wb + (st.is_update ? 0 : half) * sa;
        const int64_t j0 = j - st.shift;
        const bool inb = (j0 >= 0) && (j0 + st.nc - 1 <= half - 1);
        T x = *tgt;
        if (inb) {
            T acc = st.c[0] * opnd[j0 * sa];
            if (st.nc > 1) acc = acc + st.c[1] * opnd[(j0 + 1) * sa];
            if (st.nc > 2) acc = acc + st.c[2] * opnd[(j0 + 2) * sa];
            x = x + acc;
        } else {
            for (int k = 0; k < st.nc; ++k) x = x + st.c[k] * opnd[pmod(j0 + k, half) * sa];
        }
        *tgt = x;
    }
}

// forward finish: normalize! (transforms_lifting.jl:323-350) + scatter to destination
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_lift_finish_fwd(T n1, T n2, const T *__restrict__ w, Strides3 wst,
                          T *__restrict__ dst, Strides3 dst_st, T *__restrict__ ll, Strides3 ll_st,
                          Extent3 n, int axis, Extent3 lo, const uint8_t *__restrict__ mask)
{
    const int64_t nx = n.n[axis] >> 1;
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = nx;
    const int64_t total = e[0] * e[1] * e[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, e[0], e[1]);
        const int64_t k = c.i[axis];
        int64_t wb = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) wb += c.i[d] * wst.s[d];
        if (mask != nullptr && !mask[c.i[1]]) {
            int64_t o0 = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) o0 += (d == axis ? 0 : c.i[d]) * dst_st.s[d];
            dst[o0 + (2 * k) * dst_st.s[axis]] = w[wb + (2 * k) * wst.s[axis]];
            dst[o0 + (2 * k + 1) * dst_st.s[axis]] = w[wb + (2 * k + 1) * wst.s[axis]];
            continue;
        }
        T s = w[wb + k * wst.s[axis]] * n1;
        T dd = w[wb + (nx + k) * wst.s[axis]] * n2;
        if (ll != nullptr && in_low_corner(c, axis, lo)) {
            int64_t o = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) o += c.i[d] * ll_st.s[d];
            ll[o] = s;
        } else {
            int64_t o = 0;
#pragma unroll
            for (int d = 0; d < 3; ++d) o += c.i[d] * dst_st.s[d];
            dst[o] = s;
        }
        int64_t od = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) od += (d == axis ? (nx + k) : c.i[d]) * dst_st.s[d];
        dst[od] = dd;
    }
}

// inverse start: normalize! from the (strided) source into the work buffer
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_lift_norm_inv(T n1, T n2, const T *__restrict__ src, Strides3 sst,
                        const T *__restrict__ ll, Strides3 ll_st, T *__restrict__ w, Strides3 wst,
                        Extent3 n, int axis, Extent3 lo, const uint8_t *__restrict__ mask)
{
    const int64_t nx = n.n[axis] >> 1;
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = nx;
    const int64_t total = e[0] * e[1] * e[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, e[0], e[1]);
        const int64_t k = c.i[axis];
        int64_t sb = 0, lb = 0, wb = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) { sb += c.i[d] * sst.s[d]; lb += c.i[d] * ll_st.s[d]; wb += c.i[d] * wst.s[d]; }
        if (mask != nullptr && !mask[c.i[1]]) {
            w[wb + (2 * k) * wst.s[axis]] = src[sb + (2 * k) * sst.s[axis]];
            w[wb + (2 * k + 1) * wst.s[axis]] = src[sb + (2 * k + 1) * sst.s[axis]];
            continue;
        }
        const bool use_ll = (ll != nullptr) && in_low_corner(c, axis, lo);
        T s = use_ll ? ll[lb + k * ll_st.s[axis]] : src[sb + k * sst.s[axis]];
        T dd = src[sb + (nx + k) * sst.s[axis]];
        w[wb + k * wst.s[axis]] = n1 * s;
        w[wb + (nx + k) * wst.s[axis]] = n2 * dd;
    }
}

// merge: Util.merge! (util_main.jl:216-278): dst[2k] = w[k], dst[2k+1] = w[nx+k]
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_lift_merge(const T *__restrict__ w, Strides3 wst, T *__restrict__ dst, Strides3 dst_st, Extent3 n, int axis,
                     const uint8_t *__restrict__ mask)
{
    const int64_t nx = n.n[axis] >> 1;
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = nx;
    const int64_t total = e[0] * e[1] * e[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, e[0], e[1]);
        const int64_t k = c.i[axis];
        int64_t db = 0, wb = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d != axis) { db += c.i[d] * dst_st.s[d]; wb += c.i[d] * wst.s[d]; }
        if (mask != nullptr && !mask[c.i[1]]) {
            dst[db + (2 * k) * dst_st.s[axis]] = w[wb + (2 * k) * wst.s[axis]];
            dst[db + (2 * k + 1) * dst_st.s[axis]] = w[wb + (2 * k + 1) * wst.s[axis]];
            continue;
        }
        dst[db + (2 * k) * dst_st.s[axis]] = w[wb + k * wst.s[axis]];
        dst[db + (2 * k + 1) * dst_st.s[axis]] = w[wb + (nx + k) * wst.s[axis]];
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
k_generic_copy_box(const T *__restrict__ src, Strides3 sst, T *__restrict__ dst, Strides3 dst_st, Extent3 n)
{
    const int64_t total = n.n[0] * n.n[1] * n.n[2];
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBlock) {
        Idx3 c = unflatten(t, n.n[0], n.n[1]);
        int64_t so = 0, dof = 0;
#pragma unroll
        for (int d = 0; d < 3; ++d) { so += c.i[d] * sst.s[d]; dof += c.i[d] * dst_st.s[d]; }
        dst[dof] = src[so];
    }
}

// the same copy for boxes whose lines are contiguous, 16-byte aligned and a multiple of 16 bytes long (the staging copies of the
// in-place transforms: 512 MiB for an in-place batched inverse): one 16-byte access per thread and step, lines over
// blockIdx.y / z with grid-stride loops, no index division (the element-wise kernel above moves 1.3 TB/s)
template <typename T>
__global__ void __launch_bounds__(kBlock)
k_copy_lines(const T *__restrict__ src, Strides3 sst, T *__restrict__ dst, Strides3 dst_st, Extent3 n)
{
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    const int64_t nv = n.n[0] / V;
    for (int64_t i2 = blockIdx.z; i2 < n.n[2]; i2 += gridDim.z)
        for (int64_t i1 = blockIdx.y; i1 < n.n[1]; i1 += gridDim.y) {
            const VT *sp = reinterpret_cast<const VT *>(src + i1 * sst.s[1] + i2 * sst.s[2]);
            VT *dp = reinterpret_cast<VT *>(dst + i1 * dst_st.s[1] + i2 * dst_st.s[2]);
            for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < nv; i0 += (int64_t)gridDim.x * kBlock) dp[i0] = sp[i0];
        }
}

// --------------------------------------------------------------------------------------
// launchers
template <typename T>
hipError_t generic_fwd_filter_pass(hipStream_t st, const Taps<T> &taps, const T *src, Strides3 sst,
                                   T *dst, Strides3 dst_st, T *ll, Strides3 ll_st,
                                   Extent3 n, int axis, Extent3 lo, const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2] / 2;
    hipLaunchKernelGGL(k_generic_fwd_filter<T>, grid_for(total), dim3(kBlock), 0, st,
                       taps, src, sst, dst, dst_st, ll, ll_st, n, axis, lo, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_inv_filter_pass(hipStream_t st, const Taps<T> &taps, const T *src, Strides3 sst,
                                   const T *ll, Strides3 ll_st, T *dst, Strides3 dst_st,
                                   Extent3 n, int axis, Extent3 lo, const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2];
    hipLaunchKernelGGL(k_generic_inv_filter<T>, grid_for(total), dim3(kBlock), 0, st,
                       taps, src, sst, ll, ll_st, dst, dst_st, n, axis, lo, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_lift_split(hipStream_t st, const T *src, Strides3 sst, T *w, Strides3 wst, Extent3 n, int axis,
                              const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2] / 2;
    hipLaunchKernelGGL(k_generic_lift_split<T>, grid_for(total), dim3(kBlock), 0, st, src, sst, w, wst, n, axis, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_lift_step(hipStream_t st, const LiftStep<T> &step, T *w, Strides3 wst, Extent3 n, int axis,
                             const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2] / 2;
    hipLaunchKernelGGL(k_generic_lift_step<T>, grid_for(total), dim3(kBlock), 0, st, step, w, wst, n, axis, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_lift_finish_fwd(hipStream_t st, T n1, T n2, const T *w, Strides3 wst, T *dst, Strides3 dst_st,
                                   T *ll, Strides3 ll_st, Extent3 n, int axis, Extent3 lo, const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2] / 2;
    hipLaunchKernelGGL(k_generic_lift_finish_fwd<T>, grid_for(total), dim3(kBlock), 0, st,
                       n1, n2, w, wst, dst, dst_st, ll, ll_st, n, axis, lo, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_lift_norm_inv(hipStream_t st, T n1, T n2, const T *src, Strides3 sst, const T *ll, Strides3 ll_st,
                                 T *w, Strides3 wst, Extent3 n, int axis, Extent3 lo, const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2] / 2;
    hipLaunchKernelGGL(k_generic_lift_norm_inv<T>, grid_for(total), dim3(kBlock), 0, st,
                       n1, n2, src, sst, ll, ll_st, w, wst, n, axis, lo, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_lift_merge(hipStream_t st, const T *w, Strides3 wst, T *dst, Strides3 dst_st, Extent3 n, int axis,
                              const uint8_t *mask)
{
    int64_t total = n.n[0] * n.n[1] * n.n[2] / 2;
    hipLaunchKernelGGL(k_generic_lift_merge<T>, grid_for(total), dim3(kBlock), 0, st, w, wst, dst, dst_st, n, axis, mask);
    return hipGetLastError();
}
template <typename T>
hipError_t generic_copy_box(hipStream_t st, const T *src, Strides3 sst, T *dst, Strides3 dst_st, Extent3 n)
{
    constexpr int V = 16 / sizeof(T);
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (sst.s[0] == 1 && dst_st.s[0] == 1 && (n.n[0] % V) == 0 && (sst.s[1] % V) == 0 && (sst.s[2] % V) == 0 && (dst_st.s[1] % V) == 0 &&
        (dst_st.s[2] % V) == 0 && al(src) && al(dst) && n.n[0] >= V) {
        int64_t gx = (n.n[0] / V + kBlock - 1) / kBlock, gy = n.n[1], gz = n.n[2];
        if (gy > 65535) gy = 65535;
        if (gz > 65535) gz = 65535;
        while (gx * gy * gz > 16384) {
            if (gz > 1 && gz >= gy) gz = (gz + 1) / 2;
            else if (gy > 1) gy = (gy + 1) / 2;
            else gx = (gx + 1) / 2;
        }
        hipLaunchKernelGGL(k_copy_lines<T>, dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(kBlock), 0, st, src, sst, dst, dst_st, n);
        return hipGetLastError();
    }
    int64_t total = n.n[0] * n.n[1] * n.n[2];
    hipLaunchKernelGGL(k_generic_copy_box<T>, grid_for(total), dim3(kBlock), 0, st, src, sst, dst, dst_st, n);
    return hipGetLastError();
}

#define WL_INSTANTIATE(T)                                                                                              \
    template hipError_t generic_fwd_filter_pass<T>(hipStream_t, const Taps<T> &, const T *, Strides3, T *, Strides3,  \
                                                   T *, Strides3, Extent3, int, Extent3, const uint8_t *);             \
    template hipError_t generic_inv_filter_pass<T>(hipStream_t, const Taps<T> &, const T *, Strides3, const T *,       \
                                                   Strides3, T *, Strides3, Extent3, int, Extent3, const uint8_t *);   \
    template hipError_t generic_lift_split<T>(hipStream_t, const T *, Strides3, T *, Strides3, Extent3, int,           \
                                              const uint8_t *);                                                        \
    template hipError_t generic_lift_step<T>(hipStream_t, const LiftStep<T> &, T *, Strides3, Extent3, int,            \
                                             const uint8_t *);                                                         \
    template hipError_t generic_lift_finish_fwd<T>(hipStream_t, T, T, const T *, Strides3, T *, Strides3, T *,         \
                                                   Strides3, Extent3, int, Extent3, const uint8_t *);                  \
    template hipError_t generic_lift_norm_inv<T>(hipStream_t, T, T, const T *, Strides3, const T *, Strides3, T *,     \
                                                 Strides3, Extent3, int, Extent3, const uint8_t *);                    \
    template hipError_t generic_lift_merge<T>(hipStream_t, const T *, Strides3, T *, Strides3, Extent3, int,           \
                                              const uint8_t *);                                                        \
    template hipError_t generic_copy_box<T>(hipStream_t, const T *, Strides3, T *, Strides3, Extent3);
WL_INSTANTIATE(float)
WL_INSTANTIATE(double)

}  // namespace wl
