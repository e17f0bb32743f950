// wl_anyaxis.hip -- one filter-bank pass along ANY axis of a box of ANY even extent, even F <= 24 (compile-time taps): the pass the
// remaining odd-sized cases use -- 3-D volumes whose sides are not powers of two (100^3, 240 x 240 x 160), batched lines of
// lengths that are not multiples of 8 (44100-sample columns), 2-D blocks with an odd stride -- instead of the
// one-thread-per-output kernels of wl_generic.hip (run-time taps, two emulated 64-bit divisions per element, one dependent
// load per tap).  A thread produces FOUR consecutive coefficient pairs (forward) / eight consecutive samples (inverse) along
// the axis from one register window (2F + 4 loads instead of 8F), the index space is mapped without integer division
// (threads along dim 0, blocks over the other dims, grid-stride loops), the periodic wrap is a compare-and-reset on a
// 32-bit running index.  Same closed forms and summation order as everywhere else: bit-identical to the generic kernels.
#include "wl_fast.h"
#include "wl_dev.h"

namespace wl {

struct AnyLaunch { dim3 grid, block; };
static inline AnyLaunch any_launch(int64_t e0, int64_t e1, int64_t e2)
{
    int bx = 256;
    while (bx > 1 && (bx >> 1) >= e0) bx >>= 1;
    const int by = 256 / bx;
    int64_t gx = (e0 + bx - 1) / bx, gy = (e1 + by - 1) / by, gz = e2;
    if (gx < 1) gx = 1;
    if (gy < 1) gy = 1;
    if (gz < 1) gz = 1;
    if (gy > 65535) gy = 65535;
    if (gz > 65535) gz = 65535;
    while (gx * gy * gz > 8192) {                 // a few workgroups per CU; the grid-stride loops cover the rest
        if (gz > 1 && gz >= gy && gz >= gx) gz = (gz + 1) / 2;
        else if (gy > 1 && gy >= gx) gy = (gy + 1) / 2;
        else gx = (gx + 1) / 2;
    }
    AnyLaunch l;
    l.grid = dim3((unsigned)gx, (unsigned)gy, (unsigned)gz);
    l.block = dim3((unsigned)bx, (unsigned)by, 1);
    return l;
}
#define WL_ANY_LOOP(E0, E1, E2)                                                                                                 \
    for (int i2 = blockIdx.z; i2 < (E2); i2 += gridDim.z)                                                                       \
        for (int i1 = blockIdx.y * blockDim.y + threadIdx.y; i1 < (E1); i1 += gridDim.y * blockDim.y)                           \
            for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < (E0); i0 += gridDim.x * blockDim.x)

template <typename T, int F>
struct AnyArgs {
    const T *src; Strides3 sst;
    T *dst; Strides3 dst_st;            // fw: coefficient array (level box at its origin)        inv: result box
    T *ll; Strides3 ll_st;              // fw: approximation destination for the low corner, or nullptr
                                        // inv: approximation source for the low corner, or nullptr
    int n[3];                           // level box
    int lo[3];                          // low corner extents (n/2 along the transformed dims)
    int axis;
    TapsF<T, F> tp;
};

constexpr int kAnyPG = 4;               // coefficient pairs per thread

// forward: pairs k0 .. k0+3 of the line through (c0, c1, c2) along `axis`
template <typename T, int F>
__global__ void __launch_bounds__(256) k_fwd_any(AnyArgs<T, F> a)
{
    constexpr int PG = kAnyPG, NW = 2 * PG + 2 * F - 4;
    const int axis = a.axis;
    const int nax = a.n[axis], nx = nax >> 1, ng = (nx + PG - 1) / PG;
    int e[3] = {a.n[0], a.n[1], a.n[2]};
    e[axis] = ng;
    const int64_t sa = a.sst.s[axis], da = a.dst_st.s[axis];
    WL_ANY_LOOP(e[0], e[1], e[2]) {
        int c[3] = {i0, i1, i2};
        const int k0 = c[axis] * PG;
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
        const T *p = a.src + base;
        int idx = 2 * k0 - (F - 2);
        while (idx < 0) idx += nax;
        while (idx >= nax) idx -= nax;
        T W[NW > 0 ? NW : 2];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            W[w] = p[(int64_t)idx * sa];
            if (++idx == nax) idx = 0;
        }
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            if (k0 + q < nx) {
                T s = a.tp.h[0] * W[2 * q + F - 2];
#pragma unroll
                for (int m = 1; m < F; ++m) s = s + a.tp.h[m] * W[2 * q + F - 2 + m];
                T d = a.tp.g[F - 1] * W[2 * q];
#pragma unroll
                for (int m = F - 2; m >= 0; --m) d = d + a.tp.g[m] * W[2 * q + F - 1 - m];
                const int k = k0 + q;
                if (low) a.ll[lbase + (int64_t)k * a.ll_st.s[axis]] = s;
                else a.dst[dbase + (int64_t)k * da] = s;
                a.dst[dbase + (int64_t)(nx + k) * da] = d;
            }
        }
    }
}

// inverse: samples 2 p0 .. 2 p0 + 7 of the line through (c0, c1, c2) along `axis`
template <typename T, int F>
__global__ void __launch_bounds__(256) k_inv_any(AnyArgs<T, F> a)
{
    constexpr int PG = kAnyPG, SH = (F - 2) / 2, NS = PG + SH;
    const int axis = a.axis;
    const int nax = a.n[axis], nx = nax >> 1, ng = (nx + PG - 1) / PG;
    int e[3] = {a.n[0], a.n[1], a.n[2]};
    e[axis] = ng;
    const int64_t sa = a.sst.s[axis], da = a.dst_st.s[axis];
    WL_ANY_LOOP(e[0], e[1], e[2]) {
        int c[3] = {i0, i1, i2};
        const int p0 = c[axis] * PG;
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
        const T *ps = low ? (a.ll + lbase) : (a.src + base);
        const int64_t ss = low ? a.ll_st.s[axis] : sa;
        const T *pd = a.src + base + (int64_t)nx * sa;
        T sw[NS], dw[NS];
        int is = p0 - SH;
        while (is < 0) is += nx;
        while (is >= nx) is -= nx;
        int id = p0;
        while (id >= nx) id -= nx;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            sw[j] = ps[(int64_t)is * ss];
            dw[j] = pd[(int64_t)id * sa];
            if (++is == nx) is = 0;
            if (++id == nx) id = 0;
        }
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            if (p0 + q < nx) {
                T s1[SH + 1], d1[SH + 1];
#pragma unroll
                for (int j = 0; j <= SH; ++j) { s1[j] = sw[q + j]; d1[j] = dw[q + j]; }
                T xe, xo;
                window_inv<T, F>(s1, d1, a.tp, xe, xo);
                a.dst[dbase + (int64_t)(2 * (p0 + q)) * da] = xe;
                a.dst[dbase + (int64_t)(2 * (p0 + q) + 1) * da] = xo;
            }
        }
    }
}

bool any_axis_ok(int F, const Extent3 &n, int axis)
{
    if (F < 2 || F > 24 || (F & 1) || F == 22) return false;
    for (int d = 0; d < 3; ++d)
        if (n.n[d] < 1 || n.n[d] >= ((int64_t)1 << 30)) return false;
    return n.n[axis] >= 2 && (n.n[axis] % 2) == 0;
}

template <typename T, int F, int FW>
static hipError_t launch_any_f(hipStream_t st, const Taps<T> &taps, const T *src, Strides3 sst, T *dst, Strides3 dst_st, T *ll, Strides3 ll_st,
                               Extent3 n, int axis, Extent3 lo)
{
    AnyArgs<T, F> a;
    a.src = src; a.sst = sst; a.dst = dst; a.dst_st = dst_st; a.ll = ll; a.ll_st = ll_st; a.axis = axis;
    for (int d = 0; d < 3; ++d) { a.n[d] = (int)n.n[d]; a.lo[d] = (int)lo.n[d]; }
    a.tp = shrink<T, F>(taps);
    int64_t e[3] = {n.n[0], n.n[1], n.n[2]};
    e[axis] = ((n.n[axis] >> 1) + kAnyPG - 1) / kAnyPG;
    const AnyLaunch l = any_launch(e[0], e[1], e[2]);
    if (FW) hipLaunchKernelGGL((k_fwd_any<T, F>), l.grid, l.block, 0, st, a);
    else hipLaunchKernelGGL((k_inv_any<T, F>), l.grid, l.block, 0, st, a);
    return hipGetLastError();
}

template <typename T>
hipError_t any_axis_pass(hipStream_t st, const Taps<T> &taps, int fw, const T *src, Strides3 sst, T *dst, Strides3 dst_st, T *ll,
                         Strides3 ll_st, Extent3 n, int axis, Extent3 lo)
{
#define WL_ANY(FF_)                                                                                                    \
    case FF_: return fw ? launch_any_f<T, FF_, 1>(st, taps, src, sst, dst, dst_st, ll, ll_st, n, axis, lo)             \
                        : launch_any_f<T, FF_, 0>(st, taps, src, sst, dst, dst_st, ll, ll_st, n, axis, lo);
    switch (taps.F) {
        WL_ANY(2) WL_ANY(4) WL_ANY(6) WL_ANY(8) WL_ANY(10) WL_ANY(12) WL_ANY(14) WL_ANY(16) WL_ANY(18) WL_ANY(20) WL_ANY(24)
    default: return hipErrorInvalidValue;
    }
#undef WL_ANY
}
template hipError_t any_axis_pass<float>(hipStream_t, const Taps<float> &, int, const float *, Strides3, float *, Strides3, float *, Strides3, Extent3,
                                         int, Extent3);
template hipError_t any_axis_pass<double>(hipStream_t, const Taps<double> &, int, const double *, Strides3, double *, Strides3, double *, Strides3,
                                          Extent3, int, Extent3);

}  // namespace wl
