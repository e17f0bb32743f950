// wl_fast.h -- entry points of the fast-path translation units (wl_fwd.hip, wl_inv.hip, wl_lift.hip, wl_axis.hip) used by
// the ABI layer (wl_api.hip) and by each other.  Every fast kernel must produce results bit-identical to the generic
// kernels (wl_generic.hip).
#pragma once
#include "wl_internal.h"

namespace wl {

// Enqueue all L forward levels of a filter-bank transform.  path: 0 = pick the best kernel
// per level (streaming kernels for large levels, one LDS-resident kernel for the tail,
// generic otherwise), 1 = generic kernels only.  *kernel_name = dominant kernel used.
template <typename T>
int filter_fwd_levels(void *ws, bool ws_gen, int cu_count, int path, hipStream_t st, const BoxSpec &b,
                      T *y, const T *x, const Taps<T> &taps, int L,
                      const char **kernel_name, int *hip_err);

// Inverse filter-bank transform: all L levels (streaming kernels for large levels, generic otherwise).
template <typename T>
int filter_inv_levels(void *ws, bool ws_gen, int cu_count, int path, hipStream_t st, const BoxSpec &b,
                      T *y, const T *x, const Taps<T> &taps, int L,
                      const char **kernel_name, int *hip_err);

// One level of `nlines` independent lines (segment = line) with the streaming kernels; false = not eligible.
template <typename T>
bool fast_lines_fwd_level(hipStream_t st, const Taps<T> &taps, const T *src, int64_t src_ls, T *sdst, int64_t s_ls,
                          T *ddst, int64_t d_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err);
template <typename T>
bool fast_lines_inv_level(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls,
                          T *dst, int64_t o_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err);

// Lifting transform (forward or inverse) of `nlines` independent lines of length n (line stride ld):
// 1-D vectors and batched columns.  Sets *handled = 1 when everything was enqueued by the fused
// kernels of wl_lift.hip; otherwise enqueues nothing.
template <typename T>
int lifting_lines_fast(void *ws, int cu_count, hipStream_t st, int64_t n, int64_t nlines, int64_t ld,
                       T *y, const T *x, const LiftScheme<T> &sc, int L, int fw,
                       int *handled, const char **kernel_name, int *hip_err);

// 2-D lifting transform of a square n0 x n0 array (leading dimension ldy) through the fused line
// kernels + tiled transposes.  *handled = 1 when it was enqueued.
template <typename T>
int lifting_2d_fast(void *ws, int cu_count, hipStream_t st, int64_t n0, int64_t ldy, T *y, const T *x,
                    const LiftScheme<T> &sc, int L, int fw, int *handled, const char **kernel_name, int *hip_err);

// Level-1 source view of a batch of planes (translation-invariant denoise): plane p of the batch is NOT materialised; it is
// read from row-shifted copy (spin0 + p) % mod of the image with its columns rotated by (spin0 + p) / mod -- a circular shift
// along dim 2 is an offset of the column index, only the shift along the contiguous dim 1 needs a copy (mod copies instead of
// mod * nsp1 planes).  Set by the caller around filter_fwd_levels; consumed (and `used` set) by the plane-batched level-1 launch.
// th >= 0 additionally makes that launch threshold the level-1 DETAIL coefficients as it stores them (threshold!(xt, th, sigma * t),
// denoising.jl:58; sigma = sigma_host, or *mad_dev / 0.6745 when sigma_host < 0): the caller then thresholds only the
// approximation quadrant, which the deeper levels fill.
struct SrcView { int mod; int64_t spin0; int used; int th; double t_unit, sigma_host; const double *mad_dev;
                 int64_t corner0, corner1; };      // th >= 0: the low corner [0, corner0) x [0, corner1) of every plane is what the fused thresholds left
extern thread_local SrcView tl_srcview;

// LDS-exchange streaming kernel (wl_fwd2d.hip): one (nlev = 1) or two (nlev = 2) fused forward 2-D levels, Float32,
// even F <= 10 (F <= 8 for nlev = 2).  fwd2d_lds_ok = shape eligibility.
bool fwd2d_lds_ok(int F, int nlev, int64_t ms, int64_t ns);
hipError_t fwd2d_lds_launch(hipStream_t st, const Taps<float> &taps, int nlev, bool lvl1, const float *src, int64_t lds,
                            float *y, int64_t ldy, float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                            int64_t nbatch = 1, int64_t bs_src = 0, int64_t bs_y = 0, int64_t bs_ll = 0, int nll = 1, int src_mod = 0,
                            int64_t spin0 = 0, const SrcView *thresh = nullptr);

// 12..20 taps in one pass per level (wl_fwd2d_long.hip): the LDS-exchange kernel with a 24-slot ring and a wider window.
bool fwd2d_long_ok(int F, int64_t ms, int64_t ns);
hipError_t fwd2d_long_launch(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                             float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count);

// The Float64 instance of the LDS-exchange kernel (wl_fwd2d64.hip): two rows per lane, exact tiling only.
bool fwd2d_lds64_ok(int F, int64_t ms, int64_t ns);
hipError_t fwd2d_lds64_launch(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds,
                              double *y, int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count,
                              int64_t nbatch = 1, int64_t bs_src = 0, int64_t bs_y = 0, int64_t bs_ll = 0, int nll = 1);

// ... and of the fused pair (wl_pair2d64.hip)
bool fwd2d_pair64_ok(int F, int64_t ms, int64_t ns);
hipError_t fwd2d_pair64_launch(hipStream_t st, const Taps<double> &taps, bool lvl1, const double *src, int64_t lds, double *y,
                               int64_t ldy, double *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count);

// Two fused forward 2-D levels per launch with a dedicated level-2 wave per workgroup (wl_pair2d.hip), Float32, even F <= 10,
// blocks whose rows tile into strips of 1024 / 512.
bool fwd2d_pair_ok(int F, int64_t ms, int64_t ns);
hipError_t fwd2d_pair_launch(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                             float *ll, int64_t ldll, int64_t ms, int64_t ns, int cu_count, int64_t nbatch = 1, int64_t bs_src = 0,
                             int64_t bs_y = 0, int64_t bs_ll = 0, int src_mod = 0, int64_t spin0 = 0, const SrcView *thresh = nullptr);

// Tile kernel for the cache-resident 2-D levels (wl_tile.hip): NL = 1..3 fused forward levels per launch, Float32, even F <= 10.
bool fwd2d_tile_ok(int F, int NL, int64_t M, int64_t N);
// four levels per launch: the fused pair + the 64 x 64 two-level tiles of its approximation behind in-launch hand-over flags (wl_pair2d.hip)
bool fwd2d_pair_tile_ok(int F, int64_t ms, int64_t ns, int cu_count);
hipError_t fwd2d_pair_tile_launch(hipStream_t st, const Taps<float> &taps, bool lvl1, const float *src, int64_t lds, float *y, int64_t ldy,
                                  float *ll2, int64_t ldll2, float *ll4, int64_t ldll4, int64_t ms, int64_t ns, int cu_count, unsigned *prog);
template <typename T>
hipError_t fwd2d_tile_launch(hipStream_t st, const Taps<T> &taps, int NL, const T *src, int64_t lds, T *y, int64_t ldy, T *ll,
                             int64_t ldll, int M, int N);

// ... and its variant without the staging buffer (two levels, Float32): four workgroups per CU for blocks of 2048 rows
hipError_t fwd2d_tileB_launch(hipStream_t st, const Taps<float> &taps, const float *src, int64_t lds, float *y, int64_t ldy, float *ll,
                              int64_t ldll, int M, int N);

// Deep tail of a forward transform (wl_tail.hip): every remaining level of a small power-of-two block / line in one launch.
// one 2-D lifting level of a square block of 128 ... 2048 rows (a multiple of 64) in one launch of 64 x 64 tiles (wl_lift_tile.hip);
// id = the scheme shape (even: forward, odd: inverse); arguments as the level kernels of wl_lift.hip
bool lift2d_tile_ok(int id, int64_t n);
bool lift2d_tile2_ok(int id, int64_t n);
bool lift2d_tile2_inv_ok(int id, int64_t n);
template <typename T>
hipError_t lift2d_tile2_inv_launch(int id, hipStream_t st, const LiftScheme<T> &sc, const T *x, int64_t ldx, T *out, int64_t ldo, const T *ll, int64_t ldl,
                                   int64_t n);
template <typename T>
hipError_t lift2d_tile2_fwd_launch(int id, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy, T *ll, int64_t ldl,
                                   int64_t n);
template <typename T>
hipError_t lift2d_tile_launch(int id, int fw, hipStream_t st, const LiftScheme<T> &sc, const T *src, int64_t lds, T *y, int64_t ldy, T *ll,
                              int64_t ldl, int64_t n);

// one lifting pass along any axis of a box of any even extent, known scheme shapes (wl_lift.hip)
template <typename T>
bool lift_any_pass(hipStream_t st, const LiftScheme<T> &sc, int fw, const T *src, Strides3 sst, T *dst, Strides3 dst_st, T *ll,
                   Strides3 ll_st, Extent3 n, int axis, Extent3 lo, hipError_t *err);
// one filter-bank pass along any axis of a box of any even extent, F <= 10 (wl_anyaxis.hip)
bool any_axis_ok(int F, const Extent3 &n, int axis);
template <typename T>
hipError_t any_axis_pass(hipStream_t st, const Taps<T> &taps, int fw, const T *src, Strides3 sst, T *dst, Strides3 dst_st, T *ll,
                         Strides3 ll_st, Extent3 n, int axis, Extent3 lo);
// one 2-D level of any even extents (wl_gtile.hip)
bool gtile_ok(int F, int64_t M, int64_t N);
template <typename T>
hipError_t gtile_launch(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t lds, T *y, int64_t ldy, T *ll, int64_t ldll,
                        int M, int N);
// two inverse 2-D levels of a cache-resident block per launch (wl_tile.hip)
template <typename T>
bool inv2d_tile2_ok(int F, int64_t M, int64_t N);
template <typename T>
hipError_t inv2d_tile2_launch(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl, T *dst, int64_t ldd,
                              int M, int N);
// 3-D boxes of <= 4096 elements: all remaining forward levels / the deepest inverse levels in one workgroup (wl_tail.hip)
template <typename T>
bool tail3_ok(int F, int64_t n0, int64_t n1, int64_t n2, int nlev);
template <typename T>
hipError_t launch_tail3(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t s1, int64_t s2, T *y, int64_t y1, int64_t y2,
                        int n0, int n1, int n2, int nlev);
template <typename T>
bool tail2_inv_ok(int F, int nt, int64_t n0, int64_t n1, int nlev, const T *out, int64_t out_item);
template <typename T>
hipError_t launch_tail2_inv(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, int64_t x_item, T *out, int64_t ldo,
                            int64_t out_item, int nitems, int n0, int n1, int nt, int nlev);
template <typename T>
bool tail2_ok(int F, int nt, int64_t m0, int64_t m1, int nlev, int fmax = 10);
template <typename T>
hipError_t launch_tail2(hipStream_t st, const Taps<T> &taps, const T *src, int64_t s1, T *y, int64_t ldy,
                        int64_t src_item, int64_t y_item, int nitems, int m0, int m1, int nt, int nlev);

// Long filters (12, 14, 16, 18, 20, 24 taps): one level of contiguous lines / of the strided axis of a matrix (wl_axis.hip).
bool long_filter_ok(int F);
bool long_shape2d_ok(int F, int64_t n0, int64_t n1);
// ---- odd-length / up to 64-tap filters (Battle 23, 41, 59): wl_vlong.hip ----
bool vlong_filter_ok(int F);
bool vlong_only(int F);
template <typename T>
hipError_t vl_lines_fwd(hipStream_t st, const Taps<T> &taps, const T *src, int64_t src_ls, T *sdst, int64_t s_ls, T *ddst,
                        int64_t d_ls, int64_t n, int64_t nlines);
template <typename T>
hipError_t vl_lines_inv(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls, T *dst,
                        int64_t o_ls, int64_t n, int64_t nlines);
template <typename T>
hipError_t vl_axis(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t lds, T *dst, int64_t ldd, int64_t R, int64_t C,
                   int cu_count);
template <typename T>
bool long_lines_fwd_level(hipStream_t st, const Taps<T> &taps, const T *src, int64_t src_ls, T *sdst, int64_t s_ls,
                          T *ddst, int64_t d_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err);
template <typename T>
bool long_lines_inv_level(hipStream_t st, const Taps<T> &taps, const T *ssrc, int64_t s_ls, const T *dsrc, int64_t d_ls,
                          T *dst, int64_t o_ls, int64_t n, int64_t nlines, int cu_count, hipError_t *err);
template <typename T>
bool long_axis_level(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t lds, T *dst, int64_t ldd,
                     int64_t R, int64_t C, int cu_count, hipError_t *err);

// The fused 2-D level kernels on a batch of planes (two of the three passes of a 3-D level in one launch).
template <typename T>
bool fwd2d_planes(hipStream_t st, const Taps<T> &taps, const T *src, T *y, int64_t y1, int64_t y2, T *ll, int64_t n0, int64_t n1,
                  int64_t nplanes, int nll, int cu_count, hipError_t *err);
template <typename T>
bool inv2d_planes(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *ll, T *dst, int64_t n0, int64_t n1,
                  int64_t nplanes, int nll, int cu_count, hipError_t *err, const char **kernel = nullptr, int64_t dst_plane_stride = 0);

// 3-D lifting transform of a cube (2^k <= 512 per side) through the axis-streaming and short-line kernels.
template <typename T>
int lifting_3d_fast(void *ws, int cu_count, hipStream_t st, int64_t n0, T *y, const T *x,
                    const LiftScheme<T> &sc, int L, int fw, int *handled, const char **kernel_name, int *hip_err);

// One forward 3-D level in one pass over HBM (wl_fwd3d.hip): even F <= 8, lines of 128 ... 1024, both element types.
template <typename T>
bool fwd3d_one_ok(int F, const T *cur, int64_t c1, int64_t c2, const T *y, int64_t y1, int64_t y2, const T *ll, const int64_t n[3], bool any_tier = false);
template <typename T>
hipError_t fwd3d_one_launch(hipStream_t st, const Taps<T> &taps, const T *cur, int64_t c1, int64_t c2, T *y, int64_t y1, int64_t y2,
                            T *ll, const int64_t n[3], int cu_count);

// One inverse 3-D level in one pass over HBM (wl_inv3d.hip): even F <= 8, lines of 32 ... 1024, both element types.
template <typename T>
bool inv3d_one_ok(int F, const T *x, int64_t x1, int64_t x2, const T *ll, const T *out, int64_t o1, int64_t o2, const int64_t n[3], bool any_tier = false);
template <typename T>
hipError_t inv3d_one_launch(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *ll, T *out, int64_t o1, int64_t o2,
                            const int64_t n[3], int cu_count);

// One small 3-D level (4096 < elements <= 2^18) in one launch, forward or inverse (wl_level3.hip): LDS blocks of 4^3 / 8^3 pairs.
template <typename T>
bool level3_lds_ok(int F, const int64_t n[3], bool any_tier = false);
template <typename T>
hipError_t level3_lds_launch(hipStream_t st, const Taps<T> &taps, int fw, const T *src, int64_t s1, int64_t s2, T *dst, int64_t d1, int64_t d2,
                             const T *llr, T *llw, const int64_t n[3]);

// One 3-D filter-bank level assembled from single-axis streaming passes (wl_axis.hip); false = not eligible.
template <typename T>
bool fast3d_fwd_level(hipStream_t st, const Taps<T> &taps, const T *cur, int64_t c1, int64_t c2, T *y, int64_t y1, int64_t y2,
                      T *ll, const int64_t n[3], T *T0, T *T1, int cu_count, hipError_t *err, const char **kname = nullptr);
template <typename T>
bool fast3d_inv_level(hipStream_t st, const Taps<T> &taps, const T *x, int64_t x1, int64_t x2, const T *llsrc,
                      T *out, int64_t o1, int64_t o2, const int64_t n[3], T *T0, T *T1, int cu_count, hipError_t *err, const char **kname = nullptr);

// one fused 2-D inverse level through an LDS exchange, 8..20 taps (wl_inv2d_long.hip: which filter lengths per element type); ll = deeper reconstruction or nullptr
bool inv2d_long_ok(int F, int64_t n0, int64_t n1, int esize);
// (a batch of independent blocks: nplanes over blockIdx.y, element strides, the first nll planes take their approximation from ll)
struct InvLongBatch { int64_t nplanes, bs_x, bs_ll, bs_dst; int nll; };
template <typename T>
hipError_t inv2d_long_launch(hipStream_t st, const Taps<T> &taps, const T *x, int64_t ldx, const T *ll, int64_t ldl,
                             T *dst, int64_t ldd, int64_t n0, int64_t n1, int cu_count, const InvLongBatch &bt = InvLongBatch{1, 0, 0, 0, 1});

// ---- fully split depths of the packet transform (wl_wpt.hip) ----
template <typename T> int wpt_tile_samples();
template <typename T> bool wpt_fwd_multi_ok(int F, int64_t n, int64_t nj, int NL);
template <typename T> hipError_t wpt_fwd_multi_launch(hipStream_t st, const Taps<T> &taps, const T *src, T *dst, int64_t n, int64_t nj, int NL, const uint8_t *mask = nullptr);
template <typename T> bool wpt_inv_multi_ok(int F, int64_t n, int64_t nj, int NL);
template <typename T> hipError_t wpt_inv_multi_launch(hipStream_t st, const Taps<T> &taps, const T *src, T *dst, int64_t n, int64_t nj, int NL, const uint8_t *mask = nullptr);
template <typename T> bool wpt_tail_ok(int F, int64_t n, int64_t nj, int ndepth);
template <typename T> hipError_t wpt_tail_launch(hipStream_t st, const Taps<T> &taps, int fw, const T *src, T *dst, int64_t n, int64_t nj, int ndepth, const uint8_t *mask = nullptr);

}  // namespace wl
