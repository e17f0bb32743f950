// wl_api.hip -- the C ABI of libwavelets_mi355x.so (include/wavelets_mi355x.h):
// argument contract of the reference's _dwt!/_wpt! methods, workspace management, and
// the host-side level loops that sequence the HIP kernels on the caller's stream.
#include "wl_internal.h"
#include "wl_fast.h"

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

using namespace wl;

#include "wl_ctx.h"

thread_local const wl::Opts *wl::tl_opts = nullptr;
thread_local unsigned *wl::tl_sync = nullptr;

int wl_ensure_ws(wl_ctx *ctx, size_t bytes, hipStream_t st, bool ordered)
{
    if (bytes <= ctx->ws_bytes) return WL_OK;
    // Grow-only.  The old block may still be in use by kernels queued on the caller's stream.
    //  * ordered (growth inside a transform call, round 4): STREAM-ORDERED -- the old block is released and the new one allocated
    //    in the order of the call's stream (hipFreeAsync / hipMallocAsync), nothing waits on the host, the device is not
    //    synchronised.  A context belongs to one stream (header), so every user of the old block is ahead of the release.
    //  * otherwise (wl_ctx_reserve, which takes no stream): device synchronisation, then the same pool on the null stream,
    //    synchronised before returning so that the block is usable from any stream.
    const size_t want = (bytes + 255) & ~(size_t)255;
    const bool pool = wl::opt("WL_WS_SYNC_ALLOC", 0) == 0;
    void *p = nullptr;
    if (ordered) {
        // A growth while the call's stream is being captured into a hipGraph would record alloc / free nodes and leave the context
        // pointing at graph-owned memory that later eager calls use: refuse (reserve wl_workspace_bytes_full before capturing).
        // A query that itself fails (e.g. hipErrorStreamCaptureImplicit: the legacy stream while another stream captures) is treated as
        // "capturing", and its error is cleared so that a later launch's hipGetLastError() cannot report it.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const hipError_t qe = hipStreamIsCapturing(st, &cs);
        if (qe != hipSuccess) (void)hipGetLastError();
        if (qe != hipSuccess || cs != hipStreamCaptureStatusNone) { ctx->last_hip = (int)hipErrorStreamCaptureUnsupported; return WL_EHIP; }
    } else {
        // wl_ctx_reserve (no stream argument): it synchronises the device, which is illegal while ANY stream of this thread's capture
        // mode is capturing -- refuse before touching anything (the legacy-stream query reports an ongoing capture as an error)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const hipError_t qe = hipStreamIsCapturing(nullptr, &cs);
        if (qe != hipSuccess) (void)hipGetLastError();
        if (qe != hipSuccess || cs != hipStreamCaptureStatusNone) { ctx->last_hip = (int)hipErrorStreamCaptureUnsupported; return WL_EHIP; }
    }
    if (ordered && pool && (ctx->ws == nullptr || ctx->ws_pooled)) {
        if (ctx->ws) { WL_HIP(ctx, hipFreeAsync(ctx->ws, st)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
        hipError_t e = hipMallocAsync(&p, want, st);
        if (e == hipSuccess) {
            ctx->ws = p; ctx->ws_bytes = want; ctx->ws_pooled = true;
            return WL_OK;
        }
        // The released block cannot be reused by the pool before the stream reaches the release, so old + new had to fit.  Fall
        // through to the synchronising path (device sync, pool trimmed, retry) before reporting WL_ENOMEM.
        (void)hipGetLastError();
        hipMemPool_t mp = nullptr;
        int dev = 0;
        if (hipDeviceSynchronize() == hipSuccess && hipGetDevice(&dev) == hipSuccess && hipDeviceGetDefaultMemPool(&mp, dev) == hipSuccess && mp)
            (void)hipMemPoolTrimTo(mp, 0);
        (void)hipGetLastError();
    }
    WL_HIP(ctx, hipDeviceSynchronize());
    if (ctx->ws) {
        if (ctx->ws_pooled) { WL_HIP(ctx, hipFreeAsync(ctx->ws, nullptr)); WL_HIP(ctx, hipStreamSynchronize(nullptr)); }
        else WL_HIP(ctx, hipFree(ctx->ws));
        ctx->ws = nullptr; ctx->ws_bytes = 0;
    }
    hipError_t e;
    if (pool) {
        e = hipMallocAsync(&p, want, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    } else {
        e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); ctx->last_hip = (int)e; return WL_ENOMEM; }
    ctx->ws = p;
    ctx->ws_bytes = want;
    ctx->ws_pooled = pool;
    return WL_OK;
}

int wl_stage_to_device(wl_ctx *ctx, void *dst, const void *host, size_t bytes, hipStream_t st)
{
    const int k = ctx->stage_next;
    ctx->stage_next = (k + 1) % wl_ctx::kStage;
    if (ctx->stage_ev[k]) WL_HIP(ctx, hipEventSynchronize(ctx->stage_ev[k]));      // the copy that last used this slot
    else WL_HIP(ctx, hipEventCreateWithFlags(&ctx->stage_ev[k], hipEventDisableTiming));
    if (ctx->stage_bytes[k] < bytes) {
        if (ctx->stage[k]) { WL_HIP(ctx, hipHostFree(ctx->stage[k])); ctx->stage[k] = nullptr; ctx->stage_bytes[k] = 0; }
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        if (hipHostMalloc(&ctx->stage[k], want, hipHostMallocDefault) != hipSuccess) { ctx->stage[k] = nullptr; return WL_ENOMEM; }
        ctx->stage_bytes[k] = want;
    }
    std::memcpy(ctx->stage[k], host, bytes);
    WL_HIP(ctx, hipMemcpyAsync(dst, ctx->stage[k], bytes, hipMemcpyHostToDevice, st));
    WL_HIP(ctx, hipEventRecord(ctx->stage_ev[k], st));
    return WL_OK;
}

namespace {

inline bool sufficientpoweroftwo(int64_t n, int L) { return L < 62 && (n % ((int64_t)1 << L)) == 0; }

inline int ensure_ws(wl_ctx *ctx, size_t bytes) { return wl_ensure_ws(ctx, bytes); }
inline int ensure_ws(wl_ctx *ctx, size_t bytes, hipStream_t st) { return wl_ensure_ws(ctx, bytes, st, true); }

// makescheme (transforms_lifting.jl:13-25)
template <typename T>
int make_scheme(int nsteps, const int32_t *is_update, const int32_t *ncoef, const int32_t *shift,
                const double *coefs, double norm1, double norm2, int fw, LiftScheme<T> &sc)
{
    if (nsteps < 0 || nsteps > WL_MAX_STEPS) return WL_EINVAL_SCHEME;
    if (nsteps > 0 && (!is_update || !ncoef || !shift || !coefs)) return WL_EINVAL_ARG;
    int off[WL_MAX_STEPS];
    int o = 0;
    for (int i = 0; i < nsteps; ++i) {
        if (ncoef[i] < 1 || ncoef[i] > WL_MAX_NCOEF) return WL_EINVAL_SCHEME;
        off[i] = o;
        o += ncoef[i];
    }
    sc.nsteps = nsteps;
    for (int i = 0; i < nsteps; ++i) {
        int j = fw ? i : nsteps - 1 - i;
        LiftStep<T> &st = sc.step[i];
        st.is_update = is_update[j] ? 1 : 0;
        st.nc = ncoef[j];
        st.shift = shift[j];
        for (int k = 0; k < WL_MAX_NCOEF; ++k) st.c[k] = (T)0;
        for (int k = 0; k < st.nc; ++k) st.c[k] = (T)(coefs[off[j] + k] * (fw ? -1.0 : 1.0));
    }
    sc.norm1 = (T)(fw ? norm1 : 1.0 / norm1);
    sc.norm2 = (T)(fw ? norm2 : 1.0 / norm2);
    return WL_OK;
}

// ---- generic filter level loops ----------------------------------------------------------
// ---- generic lifting level loops -----------------------------------------------------------
template <typename T>
int generic_lifting_fwd(wl_ctx *ctx, hipStream_t st, const BoxSpec &b, T *y, const T *x,
                        const LiftScheme<T> &sc, int L)
{
    int64_t N = b.dims[0] * b.dims[1] * b.dims[2];
    Work<T> w = carve<T>(ctx->ws, N);
    const T *cur = x;
    Strides3 cur_st = b.full;
    int pp = 0;
    for (int l = 1; l <= L; ++l) {
        int64_t n[3];
        level_box(b, l, n);
        Extent3 ext = {{n[0], n[1], n[2]}};
        Extent3 lo = low_corner(b, n);
        const bool last = (l == L);
        T *llbuf = pp ? w.B : w.A;
        int64_t hn[3] = {lo.n[0], lo.n[1], lo.n[2]};
        Strides3 ll_st = dense_strides(hn);
        Strides3 box_st = dense_strides(n);
        const T *in = cur;
        Strides3 in_st = cur_st;
        int tog = 0;
        for (int a = b.nt - 1; a >= 0; --a) {
            // known scheme shapes: the whole pass (split, steps, normalize) in one launch (k_lift_any, wl_lift.hip) unless the
            // pass would read and write the same array (in-place 1-D level 1)
            {
                T *out = (a != 0) ? (tog ? w.T1 : w.T0) : y;
                hipError_t ea = hipSuccess;
                if (ctx->path == 0 && (const T *)out != in &&
                    lift_any_pass<T>(st, sc, 1, in, in_st, out, a != 0 ? box_st : b.full, (a != 0 || last) ? (T *)nullptr : llbuf, ll_st, ext, a,
                                     lo, &ea)) {
                    WL_HIP(ctx, ea);
                    ctx->last_kernel = "k_lift_any";
                    if (a != 0) { in = out; in_st = box_st; tog ^= 1; }
                    continue;
                }
            }
            WL_HIP(ctx, generic_lift_split<T>(st, in, in_st, w.W, box_st, ext, a));
            for (int s = 0; s < sc.nsteps; ++s)
                WL_HIP(ctx, generic_lift_step<T>(st, sc.step[s], w.W, box_st, ext, a));
            if (a != 0) {
                T *out = tog ? w.T1 : w.T0;
                WL_HIP(ctx, generic_lift_finish_fwd<T>(st, sc.norm1, sc.norm2, w.W, box_st, out, box_st,
                                                       (T *)nullptr, box_st, ext, a, lo));
                in = out; in_st = box_st; tog ^= 1;
            } else {
                WL_HIP(ctx, generic_lift_finish_fwd<T>(st, sc.norm1, sc.norm2, w.W, box_st, y, b.full,
                                                       last ? (T *)nullptr : llbuf, ll_st, ext, a, lo));
            }
        }
        cur = llbuf; cur_st = ll_st; pp ^= 1;
    }
    return WL_OK;
}

template <typename T>
int generic_lifting_inv(wl_ctx *ctx, hipStream_t st, const BoxSpec &b, T *y, const T *x,
                        const LiftScheme<T> &sc, int L)
{
    int64_t N = b.dims[0] * b.dims[1] * b.dims[2];
    Work<T> w = carve<T>(ctx->ws, N);
    int pp = 0;
    const T *llsrc = nullptr;
    Strides3 llsrc_st = {{0, 0, 0}};
    for (int l = L; l >= 1; --l) {
        int64_t n[3];
        level_box(b, l, n);
        Extent3 ext = {{n[0], n[1], n[2]}};
        Extent3 lo = low_corner(b, n);
        Strides3 box_st = dense_strides(n);
        const T *in = x;
        Strides3 in_st = b.full;
        int tog = 0;
        T *res = nullptr;
        for (int a = 0; a < b.nt; ++a) {
            const bool firstp = (a == 0), lastp = (a == b.nt - 1);
            T *out; Strides3 out_st;
            if (lastp) {
                if (l == 1) { out = y; out_st = b.full; }
                else { out = pp ? w.B : w.A; out_st = box_st; }
                res = out;
            } else { out = tog ? w.T1 : w.T0; out_st = box_st; tog ^= 1; }
            hipError_t ea = hipSuccess;
            if (ctx->path == 0 && (const T *)out != in &&
                lift_any_pass<T>(st, sc, 0, in, in_st, out, out_st, firstp ? const_cast<T *>(llsrc) : (T *)nullptr, llsrc_st, ext, a, lo, &ea)) {
                WL_HIP(ctx, ea);                         // normalize, steps and merge of the pass in one launch (k_lift_any)
                ctx->last_kernel = "k_lift_any";
            } else {
                WL_HIP(ctx, generic_lift_norm_inv<T>(st, sc.norm1, sc.norm2, in, in_st,
                                                     firstp ? llsrc : (const T *)nullptr, llsrc_st, w.W, box_st, ext, a, lo));
                for (int s = 0; s < sc.nsteps; ++s)
                    WL_HIP(ctx, generic_lift_step<T>(st, sc.step[s], w.W, box_st, ext, a));
                WL_HIP(ctx, generic_lift_merge<T>(st, w.W, box_st, out, out_st, ext, a));
            }
            in = out; in_st = out_st;
        }
        llsrc = res; llsrc_st = box_st; pp ^= 1;
    }
    return WL_OK;
}

// ---- argument contract (transforms_filter.jl:24-38, transforms_lifting.jl:33-43,131-143) ----
int check_box(int ndims, const int64_t *dims, int L, BoxSpec &b)
{
    if (!dims) return WL_EINVAL_ARG;
    if (ndims < 1 || ndims > 3) return WL_EDIMS;
    b.nd = ndims; b.nt = ndims;
    for (int d = 0; d < 3; ++d) b.dims[d] = (d < ndims) ? dims[d] : 1;
    for (int d = 0; d < ndims; ++d)
        if (b.dims[d] < 1) return WL_EDIMS;
    if (L < 0) return WL_EINVAL_L;
    for (int d = 0; d < ndims; ++d)
        if (!sufficientpoweroftwo(b.dims[d], L)) return WL_EINVAL_SIZE;
    b.full = dense_strides(b.dims);
    return WL_OK;
}

template <typename T>
int dwt_filter_impl(wl_ctx *ctx, hipStream_t st, const BoxSpec &b, T *y, const T *x,
                    const double *qmf, int flen, int L, int fw)
{
    const int64_t N = b.dims[0] * b.dims[1] * b.dims[2];
    if (L == 0) {
        Extent3 ext = {{b.dims[0], b.dims[1], b.dims[2]}};
        WL_HIP(ctx, generic_copy_box<T>(st, x, b.full, y, b.full, ext));
        ctx->last_kernel = "copy";
        return WL_OK;
    }
    // The fast paths need the approximation ping-pong only (2 * (N >> nt) elements); the generic / long-filter / 3-D families
    // also want three N-element buffers.  Start with whichever the context already holds (at least the ping-pong); a level
    // that finds the big buffers missing reports WL_RETRY_GEN, the workspace grows once, and the call is repeated -- x is
    // never modified by a filter transform, so repeating is harmless.
    const size_t ab_bytes = ws_ab_elems(N, b.nt) * sizeof(T), full_bytes = ws_elems(N, b.nt) * sizeof(T);
    const bool want_gen = (ctx->path != 0) || (b.nd == 3) || (flen % 2 != 0) || (flen > 10 && b.nt > 1);
    int rc = ensure_ws(ctx, want_gen ? full_bytes : ab_bytes, st);
    if (rc) return rc;
    Taps<T> taps;
    make_taps<T>(qmf, flen, taps);
    for (int attempt = 0; attempt < 2; ++attempt) {
        const bool have_gen = ctx->ws_bytes >= full_bytes;
        rc = fw ? filter_fwd_levels<T>(ctx->ws, have_gen, ctx->cu_count, ctx->path, st, b, y, x, taps, L, &ctx->last_kernel, &ctx->last_hip)
                : filter_inv_levels<T>(ctx->ws, have_gen, ctx->cu_count, ctx->path, st, b, y, x, taps, L, &ctx->last_kernel, &ctx->last_hip);
        if (rc != WL_RETRY_GEN) return rc;
        rc = ensure_ws(ctx, full_bytes, st);
        if (rc) return rc;
    }
    return WL_EINVAL_ARG;
}

template <typename T>
int dwt_lifting_impl(wl_ctx *ctx, hipStream_t st, const BoxSpec &b, T *y, const T *x,
                     int nsteps, const int32_t *is_update, const int32_t *ncoef, const int32_t *shift,
                     const double *coefs, double norm1, double norm2, int L, int fw)
{
    LiftScheme<T> sc;
    int rc = make_scheme<T>(nsteps, is_update, ncoef, shift, coefs, norm1, norm2, fw, sc);
    if (rc) return rc;
    return wl_lifting_box<T>(ctx, st, b, y, x, sc, L, fw);
}

}  // namespace

template <typename T>
int wl_make_scheme(int nsteps, const int32_t *is_update, const int32_t *ncoef, const int32_t *shift, const double *coefs, double norm1,
                   double norm2, int fw, wl::LiftScheme<T> &sc)
{
    return make_scheme<T>(nsteps, is_update, ncoef, shift, coefs, norm1, norm2, fw, sc);
}
template int wl_make_scheme<float>(int, const int32_t *, const int32_t *, const int32_t *, const double *, double, double, int, wl::LiftScheme<float> &);
template int wl_make_scheme<double>(int, const int32_t *, const int32_t *, const int32_t *, const double *, double, double, int, wl::LiftScheme<double> &);

// the lifting transform of a box with a direction-adjusted scheme (shared with wl_ext.hip: the translation-invariant denoise)
template <typename T>
int wl_lifting_box(wl_ctx *ctx, hipStream_t st, const BoxSpec &b, T *y, const T *x, const LiftScheme<T> &sc, int L, int fw)
{
    const int64_t N = b.dims[0] * b.dims[1] * b.dims[2];
    int rc = WL_OK;
    if (L == 0) {
        if (y != x) {
            Extent3 ext = {{b.dims[0], b.dims[1], b.dims[2]}};
            WL_HIP(ctx, generic_copy_box<T>(st, x, b.full, y, b.full, ext));
        }
        ctx->last_kernel = "copy";
        return WL_OK;
    }
    rc = ensure_ws(ctx, ws_elems(N) * sizeof(T), st);
    if (rc) return rc;
    if (ctx->path == 0 && b.nt == 1 && b.full.s[0] == 1) {
        int handled = 0;
        rc = lifting_lines_fast<T>(ctx->ws, ctx->cu_count, st, b.dims[0], b.nd == 1 ? 1 : b.dims[1], b.full.s[1],
                                   y, x, sc, L, fw, &handled, &ctx->last_kernel, &ctx->last_hip);
        if (rc) return rc;
        if (handled) return WL_OK;
    }
    if (ctx->path == 0 && b.nd == 2 && b.nt == 2 && b.full.s[0] == 1) {
        int handled = 0;
        rc = lifting_2d_fast<T>(ctx->ws, ctx->cu_count, st, b.dims[0], b.full.s[1], y, x, sc, L, fw, &handled,
                                &ctx->last_kernel, &ctx->last_hip);
        if (rc) return rc;
        if (handled) return WL_OK;
    }
    if (ctx->path == 0 && b.nd == 3 && b.nt == 3 && b.full.s[0] == 1 && b.dims[0] == b.dims[1] && b.dims[1] == b.dims[2] &&
        b.full.s[1] == b.dims[0] && b.full.s[2] == b.dims[0] * b.dims[1]) {
        int handled = 0;
        rc = lifting_3d_fast<T>(ctx->ws, ctx->cu_count, st, b.dims[0], y, x, sc, L, fw, &handled, &ctx->last_kernel, &ctx->last_hip);
        if (rc) return rc;
        if (handled) return WL_OK;
    }
    ctx->last_kernel = fw ? "k_generic_lift_fwd" : "k_generic_lift_inv";
    return fw ? generic_lifting_fwd<T>(ctx, st, b, y, x, sc, L) : generic_lifting_inv<T>(ctx, st, b, y, x, sc, L);
}
template int wl_lifting_box<float>(wl_ctx *, hipStream_t, const BoxSpec &, float *, const float *, const LiftScheme<float> &, int, int);
template int wl_lifting_box<double>(wl_ctx *, hipStream_t, const BoxSpec &, double *, const double *, const LiftScheme<double> &, int, int);

// ==========================================================================================
extern "C" {

int wl_version(void) { return WL_VERSION; }

const char *wl_strerror(int status)
{
    switch (status) {
    case WL_OK: return "ok";
    case WL_EINVAL_SIZE: return "size must have a sufficient power of 2 factor";
    case WL_EINVAL_L: return "L must be positive";
    case WL_EALIAS: return "in array is out array";
    case WL_EDIMS: return "in and out array size must match / bad dimensions";
    case WL_EINVAL_CUBE: return "array must be square/cube";
    case WL_EINVAL_TREE: return "invalid tree";
    case WL_EINVAL_SCHEME: return "invalid lifting scheme";
    case WL_EINVAL_DTYPE: return "unsupported element type";
    case WL_EINVAL_FILTER: return "unsupported filter length";
    case WL_EINVAL_ARG: return "invalid argument";
    case WL_ENOMEM: return "device workspace allocation failed";
    case WL_EHIP: return "HIP runtime error";
    case WL_ENODEVICE: return "no gfx950 HIP device";
    default: return "unknown status";
    }
}

int wl_maxtransformlevels(int64_t n)
{
    if (n <= 1) return 0;
    int tl = 0;
    while (sufficientpoweroftwo(n, tl)) tl += 1;
    return tl - 1;
}

int wl_ctx_create(int device, wl_ctx **out)
{
    if (!out) return WL_EINVAL_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return WL_ENODEVICE;
    if (device < 0 || device >= count) return WL_ENODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return WL_ENODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return WL_ENODEVICE;
    wl_ctx *c = new (std::nothrow) wl_ctx();
    if (!c) return WL_ENOMEM;
    c->device = device;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    {
        // the hand-over block of the launches whose workgroups signal one another: zeroed once here, left zero by every launch
        int prev = -1;
        (void)hipGetDevice(&prev);
        hipError_t e = (prev == device) ? hipSuccess : hipSetDevice(device);
        if (e == hipSuccess) e = hipMalloc(&c->sync, wl::kSyncWords * sizeof(unsigned));
        if (e == hipSuccess) e = hipMemset(c->sync, 0, wl::kSyncWords * sizeof(unsigned));
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        if (e != hipSuccess) {
            if (c->sync) (void)hipFree(c->sync);
            delete c;
            (void)hipGetLastError();
            return WL_ENOMEM;
        }
    }
    *out = c;
    return WL_OK;
}

int wl_ctx_destroy(wl_ctx *ctx)
{
    if (!ctx) return WL_EINVAL_ARG;
    {
        CallScope scope(ctx);                  // free on the context's device, leave the caller's device current
        (void)hipDeviceSynchronize();
        if (ctx->ws) { if (ctx->ws_pooled) { (void)hipFreeAsync(ctx->ws, nullptr); (void)hipStreamSynchronize(nullptr); } else (void)hipFree(ctx->ws); }
        if (ctx->aux) (void)hipFree(ctx->aux);
        if (ctx->sync) (void)hipFree(ctx->sync);
        for (int k = 0; k < wl_ctx::kStage; ++k) {
            if (ctx->stage_ev[k]) (void)hipEventDestroy(ctx->stage_ev[k]);
            if (ctx->stage[k]) (void)hipHostFree(ctx->stage[k]);
        }
    }
    delete ctx;
    return WL_OK;
}

int wl_shard_range(int64_t nunits, int rank, int world, int64_t *lo, int64_t *hi)
{
    if (!lo || !hi || nunits < 0 || world < 1 || rank < 0 || rank >= world) return WL_EINVAL_ARG;
    const int64_t base = nunits / world, rem = nunits % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return WL_OK;
}

size_t wl_workspace_bytes(int dtype, int ndims, const int64_t *dims, int L)
{
    (void)L;
    if (!dims || ndims < 1 || ndims > 3) return 0;
    int64_t N = 1;
    for (int d = 0; d < ndims; ++d) N *= dims[d];
    // what the fast filter-bank paths of dwt / idwt on this box use: the approximation ping-pong, 2 * (N / 2^ndims) elements.
    // (dwtc: pass ndims = 1 with dims[0] = len * nsignals.)  Lifting, long / odd filters, 3-D boxes and the generic
    // family use up to 4 N elements more; the context grows to that on their first call.
    return ws_ab_elems(N, ndims) * (dtype == WL_F64 ? 8 : 4);
}
size_t wl_workspace_bytes_full(int dtype, int ndims, const int64_t *dims, int L)
{
    (void)L;
    if (!dims || ndims < 1 || ndims > 3) return 0;
    int64_t N = 1;
    for (int d = 0; d < ndims; ++d) N *= dims[d];
    // ping-pong + T0 / T1 / W of the general families (ws_elems); the packet transforms add a depth table of N bytes
    return ws_elems(N, 1) * (dtype == WL_F64 ? 8 : 4) + (size_t)N + 256;
}
size_t wl_ctx_workspace_held(const wl_ctx *ctx) { return ctx ? ctx->ws_bytes : 0; }

int wl_ctx_reserve(wl_ctx *ctx, size_t bytes)
{
    if (!ctx) return WL_EINVAL_ARG;
    WL_SCOPE(ctx);
    return ensure_ws(ctx, bytes);
}

int wl_stream_sync(wl_ctx *ctx, void *stream)
{
    if (!ctx) return WL_EINVAL_ARG;
    WL_SCOPE(ctx);
    WL_HIP(ctx, hipStreamSynchronize((hipStream_t)stream));
    return WL_OK;
}

int wl_last_hip_error(const wl_ctx *ctx) { return ctx ? ctx->last_hip : 0; }
int wl_ctx_set_path(wl_ctx *ctx, int path)
{
    if (!ctx || path < 0 || path > 1) return WL_EINVAL_ARG;
    ctx->path = path;
    return WL_OK;
}
const char *wl_last_kernel(const wl_ctx *ctx) { return ctx ? ctx->last_kernel : "none"; }

int wl_ctx_set_option(wl_ctx *ctx, const char *key, int64_t value)
{
    if (!ctx || !key || !*key || strlen(key) >= (size_t)Opts::kKeyLen) return WL_EINVAL_ARG;
    Opts &o = ctx->opts;
    for (int i = 0; i < o.n; ++i)
        if (strcmp(o.key[i], key) == 0) { o.val[i] = (long long)value; return WL_OK; }
    if (o.n >= Opts::kMax) return WL_EINVAL_ARG;
    strcpy(o.key[o.n], key);
    o.val[o.n] = (long long)value;
    ++o.n;
    return WL_OK;
}
int wl_ctx_clear_options(wl_ctx *ctx)
{
    if (!ctx) return WL_EINVAL_ARG;
    ctx->opts.n = 0;
    return WL_OK;
}

int wl_dwt_filter(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims,
                  const double *qmf, int flen, int L, int fw, void *stream)
{
    if (!ctx || !y || !x || !qmf) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (flen < 2 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    BoxSpec b;
    int rc = check_box(ndims, dims, L, b);
    if (rc) return rc;
    if (y == x) return WL_EALIAS;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32 ? dwt_filter_impl<float>(ctx, st, b, (float *)y, (const float *)x, qmf, flen, L, fw)
                           : dwt_filter_impl<double>(ctx, st, b, (double *)y, (const double *)x, qmf, flen, L, fw);
}

static int lifting_common(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims,
                          int nsteps, const int32_t *is_update, const int32_t *ncoef, const int32_t *shift,
                          const double *coefs, double norm1, double norm2, int L, int fw, void *stream)
{
    if (!ctx || !y || !x) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    BoxSpec b;
    // iscube check comes first in the reference (transforms_lifting.jl:131-136)
    if (dims && ndims >= 2 && ndims <= 3)
        for (int d = 1; d < ndims; ++d)
            if (dims[d] != dims[0]) return WL_EINVAL_CUBE;
    int rc = check_box(ndims, dims, L, b);
    if (rc) return rc;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32
               ? dwt_lifting_impl<float>(ctx, st, b, (float *)y, (const float *)x, nsteps, is_update, ncoef, shift, coefs, norm1, norm2, L, fw)
               : dwt_lifting_impl<double>(ctx, st, b, (double *)y, (const double *)x, nsteps, is_update, ncoef, shift, coefs, norm1, norm2, L, fw);
}

int wl_dwt_lifting(wl_ctx *ctx, int dtype, void *y, int ndims, const int64_t *dims,
                   int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                   const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                   int L, int fw, void *stream)
{
    return lifting_common(ctx, dtype, y, y, ndims, dims, nsteps, step_is_update, step_ncoef, step_shift,
                          coefs_flat, norm1, norm2, L, fw, stream);
}

int wl_dwt_lifting_oop(wl_ctx *ctx, int dtype, void *y, const void *x, int ndims, const int64_t *dims,
                       int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                       const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                       int L, int fw, void *stream)
{
    return lifting_common(ctx, dtype, y, x, ndims, dims, nsteps, step_is_update, step_ncoef, step_shift,
                          coefs_flat, norm1, norm2, L, fw, stream);
}

// ---- batched column-wise --------------------------------------------------------------------
static int check_dwtc(int64_t len, int64_t nsignals, int64_t ld, int L, BoxSpec &b)
{
    if (len < 1 || nsignals < 1 || ld < len) return WL_EDIMS;
    if (L < 0) return WL_EINVAL_L;
    if (!sufficientpoweroftwo(len, L)) return WL_EINVAL_SIZE;
    b.nd = 2; b.nt = 1;
    b.dims[0] = len; b.dims[1] = nsignals; b.dims[2] = 1;
    b.full.s[0] = 1; b.full.s[1] = ld; b.full.s[2] = ld * nsignals;
    return WL_OK;
}

int wl_dwtc_filter(wl_ctx *ctx, int dtype, void *y, const void *x, int64_t len, int64_t nsignals, int64_t ld,
                   const double *qmf, int flen, int L, int fw, void *stream)
{
    if (!ctx || !y || !x || !qmf) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (flen < 2 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    BoxSpec b;
    int rc = check_dwtc(len, nsignals, ld, L, b);
    if (rc) return rc;
    if (y == x) return WL_EALIAS;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32 ? dwt_filter_impl<float>(ctx, st, b, (float *)y, (const float *)x, qmf, flen, L, fw)
                           : dwt_filter_impl<double>(ctx, st, b, (double *)y, (const double *)x, qmf, flen, L, fw);
}

int wl_dwtc_lifting_oop(wl_ctx *ctx, int dtype, void *y, const void *x, int64_t len, int64_t nsignals, int64_t ld,
                        int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                        const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                        int L, int fw, void *stream)
{
    if (!ctx || !y || !x) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    BoxSpec b;
    int rc = check_dwtc(len, nsignals, ld, L, b);
    if (rc) return rc;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32
               ? dwt_lifting_impl<float>(ctx, st, b, (float *)y, (const float *)x, nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, L, fw)
               : dwt_lifting_impl<double>(ctx, st, b, (double *)y, (const double *)x, nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, L, fw);
}

int wl_dwtc_lifting(wl_ctx *ctx, int dtype, void *y, int64_t len, int64_t nsignals, int64_t ld,
                    int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                    const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                    int L, int fw, void *stream)
{
    if (!ctx || !y) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    BoxSpec b;
    int rc = check_dwtc(len, nsignals, ld, L, b);
    if (rc) return rc;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    return dtype == WL_F32
               ? dwt_lifting_impl<float>(ctx, st, b, (float *)y, (const float *)y, nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, L, fw)
               : dwt_lifting_impl<double>(ctx, st, b, (double *)y, (const double *)y, nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, L, fw);
}

}  // extern "C"

// ---- wavelet packet transforms (1-D) -------------------------------------------------------------
// The reference walks the tree one depth at a time (transforms_filter.jl:325-356,
// transforms_lifting.jl:297-316): at depth d the vector is 2^d segments of length n/2^d, and
// every segment whose node bit is set gets one [s ; d] level.  Here one depth = one launch over
// the box (n/2^d, 2^d) with the node bits as a per-segment mask (unset segments are copied
// through), ping-ponging between y and a work buffer so that the last launch lands in y.
// index of the last set node (-1: none); the tail of a tree vector is normally one long run of zeros
static int64_t last_set_node(const uint8_t *b, int64_t nb)
{
    int64_t i = nb;
    while (i > 0 && (reinterpret_cast<uintptr_t>(b + i) & 7) != 0) { if (b[i - 1]) return i - 1; --i; }
    while (i >= 64) {                                   // 64 bytes per iteration, independent loads
        uint64_t w[8];
        std::memcpy(w, b + i - 64, 64);
        if ((w[0] | w[1] | w[2] | w[3]) | (w[4] | w[5] | w[6] | w[7])) break;
        i -= 64;
    }
    while (i > 0) { if (b[i - 1]) return i - 1; --i; }
    return -1;
}
// isvalidtree (util_main.jl:301-314): the length matches and no node is set below an unset node.  Equivalent
// statement checked here: every set node other than the root has its parent set -- O(last set node), not O(2^ns).
static bool isvalidtree(int64_t n, const uint8_t *b, int64_t nb, int64_t *last_set)
{
    int ns = wl_maxtransformlevels(n);
    *last_set = -1;
    if (nb != ((int64_t)1 << ns) - 1) return false;
    if (ns == 0) return true;
    const int64_t hi = last_set_node(b, nb);
    *last_set = hi;
    for (int64_t j = 1; j <= hi; ++j)
        if (b[j] && !b[(j - 1) >> 1]) return false;
    return true;
}

template <typename T>
static int wpt_impl(wl_ctx *ctx, hipStream_t st, T *y, const T *x, int64_t n,
                    const Taps<T> *taps, const LiftScheme<T> *sc,
                    const uint8_t *tree, int64_t ntree, int64_t last_set, int fw, int full_depth = -1)
{
    // full_depth >= 0: the full tree of that depth, no tree vector (tree == nullptr)
    const bool lifting = (sc != nullptr);
    Extent3 full = {{n, 1, 1}};
    Strides3 fst = {{1, n, n}};
    if (full_depth >= 0 ? full_depth == 0 : (ntree == 0 || !tree[0])) {
        if (y != x) WL_HIP(ctx, generic_copy_box<T>(st, x, fst, y, fst, full));
        ctx->last_kernel = "copy";
        return WL_OK;
    }
    const int Lmax = wl_maxtransformlevels(n);
    // depths in processing order, skipping depths where no node is set (pure copy-through); kind 2: every segment splits
    std::vector<int> depths, kind;
    bool any_partial = false;
    for (int L = Lmax; L > 0; --L) {
        int d = fw ? Lmax - L : L - 1;
        if (full_depth >= 0) {
            if (d < full_depth) { depths.push_back(d); kind.push_back(2); }
            continue;
        }
        int64_t first = ((int64_t)1 << d) - 1, cnt = (int64_t)1 << d;
        int64_t nset = 0;
        if (first <= last_set)
            for (int64_t k = 0; k < cnt; ++k) nset += tree[first + k] != 0;
        if (nset) { depths.push_back(d); kind.push_back(nset == cnt ? 2 : 1); any_partial = any_partial || nset != cnt; }
    }
    const int K = (int)depths.size();
    // only the nodes up to the last set one are ever looked at on the device (whole depths: round up to 2^(d+1) - 1), and only
    // when some depth is PARTIALLY split: the kernels of fully split depths take no mask.  The bits travel through the
    // context's pinned staging ring -- the caller's (pageable) buffer is not referenced after this call returns and the stream
    // is not synchronised (round 3 did hipStreamSynchronize here, against the header's contract).
    int64_t ncopy = 1;
    while (ncopy - 1 <= last_set && ncopy - 1 < ntree) ncopy <<= 1;
    ncopy = (ncopy - 1 < ntree) ? ncopy - 1 : ntree;
    if (full_depth >= 0) ncopy = 0;
    int rc = ensure_ws(ctx, ws_elems(n) * sizeof(T) + (size_t)(any_partial ? ncopy : 0) + 256, st);
    if (rc) return rc;
    Work<T> w = carve<T>(ctx->ws, n);
    uint8_t *dtree = (uint8_t *)ctx->ws + ws_elems(n) * sizeof(T);
    if (any_partial) {
        rc = wl_stage_to_device(ctx, dtree, tree, (size_t)ncopy, st);
        if (rc) return rc;
    }
    // the packet kernels (k_wpt_fwd_multi / _inv_multi / _tail) move 16-byte vectors straight on x, y and the work buffer: a view that
    // starts 4 or 8 bytes off the grid (buf[1:1+n]) takes the per-depth kernels, which gate on alignment themselves
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool fast_any = (ctx->path == 0) && opt("WL_WPT_FAST", 1) != 0;          // the lifting line kernels check alignment themselves
    const bool fast = fast_any && al16(x) && al16(y) && al16(w.T0);              // the filter-bank packet kernels

    if (lifting) {
        // wpt!(y, scheme, ...) is in place for the caller.  A fused lifting level cannot run in place (its [s ; d] outputs land
        // where other waves still read interleaved input; lifting_lines_fast would stage and copy back: three launches), so fully
        // split depths ping-pong between y and a work buffer -- one launch per depth -- and an odd count ends with one copy.
        const char *name = "k_generic_lift_wpt";
        T *cur = y;
        for (int i = 0; i < K; ++i) {
            const int d = depths[i];
            const int64_t nj = n >> d, nseg = (int64_t)1 << d;
            // fully split depth: one lifting level (all steps fused) of nseg lines of length nj
            if (kind[i] == 2 && fast_any) {
                int handled = 0, herr = 0;
                const char *kn = nullptr;
                T *out = (cur == y) ? w.T0 : y;
                rc = lifting_lines_fast<T>(ctx->ws, ctx->cu_count, st, nj, nseg, nj, out, cur, *sc, 1, fw, &handled, &kn, &herr);
                if (rc) { ctx->last_hip = herr; return rc; }
                if (handled) { name = kn ? kn : "k_lift1d_stream"; cur = out; continue; }
            }
            if (cur != y) { WL_HIP(ctx, generic_copy_box<T>(st, cur, fst, y, fst, full)); cur = y; }
            Extent3 ext = {{nj, nseg, 1}};
            Strides3 bst = {{1, nj, n}};
            Extent3 lo = {{nj >> 1, nseg, 1}};
            const uint8_t *mask = (kind[i] == 2) ? nullptr : dtree + (((int64_t)1 << d) - 1);
            // reads of y all happen in the first kernel, so in place is safe
            if (fw) {
                WL_HIP(ctx, generic_lift_split<T>(st, y, bst, w.W, bst, ext, 0, mask));
                for (int s = 0; s < sc->nsteps; ++s)
                    WL_HIP(ctx, generic_lift_step<T>(st, sc->step[s], w.W, bst, ext, 0, mask));
                WL_HIP(ctx, generic_lift_finish_fwd<T>(st, sc->norm1, sc->norm2, w.W, bst, y, bst, (T *)nullptr, bst, ext, 0, lo, mask));
            } else {
                WL_HIP(ctx, generic_lift_norm_inv<T>(st, sc->norm1, sc->norm2, y, bst, (const T *)nullptr, bst, w.W, bst, ext, 0, lo, mask));
                for (int s = 0; s < sc->nsteps; ++s)
                    WL_HIP(ctx, generic_lift_step<T>(st, sc->step[s], w.W, bst, ext, 0, mask));
                WL_HIP(ctx, generic_lift_merge<T>(st, w.W, bst, y, bst, ext, 0, mask));
            }
        }
        if (cur != y) WL_HIP(ctx, generic_copy_box<T>(st, cur, fst, y, fst, full));
        ctx->last_kernel = name;
        return WL_OK;
    }

    // ---- filter bank: plan the launches first (the output ping-pongs between y and a work buffer and must end in y) ----
    struct Step { int kindk; int d; int nd; };          // 0 generic / line kernel (one depth), 1 / 3 forward / inverse multi (nd fused depths from depth d), 2 tail (nd depths)
    std::vector<Step> plan;
    const int F = taps->F;
    const int TSw = wpt_tile_samples<T>();
    // round 5: the packet kernels take the node bits (per-segment split mask), so partially split depths (a dwt-shaped tree, a best-basis
    // tree) ride the same fused passes -- leaves are passed through inside the launch -- instead of one generic launch per depth
    const uint8_t *pmask = any_partial ? dtree : nullptr;
    for (int i = 0; i < K;) {
        const int d = depths[i];
        int run = 0;                                     // consecutive depths d, d +- 1, ... in processing order (fully or partially split)
        while (fast && i + run < K && depths[i + run] == (fw ? d + run : d - run)) ++run;
        bool run_partial = false;
        for (int k = 0; k < run; ++k) run_partial = run_partial || kind[i + k] != 2;
        if (run >= 1 && fw) {
            const int64_t nj = n >> d;
            if (wpt_tail_ok<T>(F, n, nj, run)) { plan.push_back({2, d, run}); i += run; continue; }
            int to_tail = 0;                             // depths until the segments fit a workgroup
            while ((nj >> to_tail) > TSw) ++to_tail;
            int lim = run < to_tail ? run : to_tail;
            if (lim < 1) lim = 1;
            const int stages = (lim + 2) / 3;
            int NL = (lim + stages - 1) / stages;
            while (NL >= 1 && !wpt_fwd_multi_ok<T>(F, n, nj, NL)) --NL;
            if (NL >= 1 && (NL > 1 || run_partial || opt("WL_WPT_MULTI1", 0))) { plan.push_back({1, d, NL}); i += NL; continue; }
        } else if (run >= 1 && !fw) {
            // deepest first: take every depth of the run whose segments still fit a workgroup
            int nd = 0;
            while (nd < run && wpt_tail_ok<T>(F, n, n >> (d - nd), nd + 1)) ++nd;
            if (nd >= 1) { plan.push_back({2, d - nd + 1, nd}); i += nd; continue; }
            // big segments: up to three depths per pass (k_wpt_inv_multi), the run split evenly over the fewest launches
            const int stages = (run + 2) / 3;
            int NL = (run + stages - 1) / stages;
            const int nlmin = run_partial ? 1 : 2;
            while (NL >= nlmin && !wpt_inv_multi_ok<T>(F, n, n >> (d - NL + 1), NL)) --NL;
            if (NL >= nlmin) { plan.push_back({3, d - NL + 1, NL}); i += NL; continue; }
        }
        plan.push_back({0, d, 1});
        ++i;
    }
    const int P = (int)plan.size();
    const T *cur = x;
    const char *name = "k_generic_filter_wpt";
    for (int i = 0; i < P; ++i) {
        const Step &sp = plan[i];
        const int d = sp.d;
        T *out = ((P - 1 - i) % 2 == 0) ? y : w.T0;
        if (sp.kindk == 1) {
            WL_HIP(ctx, wpt_fwd_multi_launch<T>(st, *taps, cur, out, n, n >> d, sp.nd, pmask));
            name = "k_wpt_fwd_multi";
        } else if (sp.kindk == 3) {
            WL_HIP(ctx, wpt_inv_multi_launch<T>(st, *taps, cur, out, n, n >> d, sp.nd, pmask));
            name = "k_wpt_inv_multi";
        } else if (sp.kindk == 2) {
            WL_HIP(ctx, wpt_tail_launch<T>(st, *taps, fw, cur, out, n, n >> d, sp.nd, pmask));
            if (std::strncmp(name, "k_wpt", 5) != 0) name = fw ? "k_wpt_fwd_tail" : "k_wpt_inv_tail";
        } else {
            const int64_t nj = n >> d, nseg = (int64_t)1 << d;
            Extent3 ext = {{nj, nseg, 1}};
            Strides3 bst = {{1, nj, n}};
            Extent3 lo = {{nj >> 1, nseg, 1}};
            int ki = 0;
            while (depths[ki] != d) ++ki;
            const bool all_set = kind[ki] == 2;
            bool done = false;
            if (all_set && ctx->path == 0) {             // fully split depth: every segment is a line of the streaming kernels
                hipError_t he = hipSuccess;
                done = fw ? fast_lines_fwd_level<T>(st, *taps, cur, nj, out, nj, out + (nj >> 1), nj, nj, nseg, ctx->cu_count, &he)
                          : fast_lines_inv_level<T>(st, *taps, cur, nj, cur + (nj >> 1), nj, out, nj, nj, nseg, ctx->cu_count, &he);
                if (he != hipSuccess) return hip_fail(ctx, he);
                if (done && std::strncmp(name, "k_wpt", 5) != 0) name = fw ? "k_fwd1d_stream" : "k_inv1d_stream";
            }
            if (!done) {
                const uint8_t *mask = all_set ? nullptr : dtree + (((int64_t)1 << d) - 1);
                if (fw)
                    WL_HIP(ctx, generic_fwd_filter_pass<T>(st, *taps, cur, bst, out, bst, (T *)nullptr, bst, ext, 0, lo, mask));
                else
                    WL_HIP(ctx, generic_inv_filter_pass<T>(st, *taps, cur, bst, (const T *)nullptr, bst, out, bst, ext, 0, lo, mask));
            }
        }
        cur = out;
    }
    ctx->last_kernel = name;
    return WL_OK;
}

extern "C" {

int wl_wpt_filter(wl_ctx *ctx, int dtype, void *y, const void *x, int64_t n, const double *qmf, int flen,
                  const uint8_t *tree, int64_t ntree, int fw, void *stream)
{
    if (!ctx || !y || !x || !qmf || (!tree && ntree > 0)) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (flen < 2 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    if (n < 1) return WL_EDIMS;
    if (y == x) return WL_EALIAS;
    int64_t last_set = -1;
    if (!isvalidtree(n, tree, ntree, &last_set)) return WL_EINVAL_TREE;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WL_F32) {
        Taps<float> t; make_taps<float>(qmf, flen, t);
        return wpt_impl<float>(ctx, st, (float *)y, (const float *)x, n, &t, nullptr, tree, ntree, last_set, fw);
    }
    Taps<double> t; make_taps<double>(qmf, flen, t);
    return wpt_impl<double>(ctx, st, (double *)y, (const double *)x, n, &t, nullptr, tree, ntree, last_set, fw);
}

int wl_wpt_lifting(wl_ctx *ctx, int dtype, void *y, int64_t n,
                   int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                   const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                   const uint8_t *tree, int64_t ntree, int fw, void *stream)
{
    if (!ctx || !y || (!tree && ntree > 0)) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (n < 1) return WL_EDIMS;
    int64_t last_set = -1;
    if (!isvalidtree(n, tree, ntree, &last_set)) return WL_EINVAL_TREE;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WL_F32) {
        LiftScheme<float> sc;
        int rc = make_scheme<float>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, fw, sc);
        if (rc) return rc;
        return wpt_impl<float>(ctx, st, (float *)y, (const float *)y, n, nullptr, &sc, tree, ntree, last_set, fw);
    }
    LiftScheme<double> sc;
    int rc = make_scheme<double>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, fw, sc);
    if (rc) return rc;
    return wpt_impl<double>(ctx, st, (double *)y, (const double *)y, n, nullptr, &sc, tree, ntree, last_set, fw);
}

int wl_dwt_filter_batch(wl_ctx *ctx, int dtype, void *y, const void *x, const int64_t *dims, int64_t nimages, int64_t image_stride,
                        const double *qmf, int flen, int L, int fw, void *stream)
{
    if (!ctx || !y || !x || !dims || !qmf) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (flen < 2 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    if (dims[0] < 1 || dims[1] < 1 || nimages < 1 || image_stride < dims[0] * dims[1]) return WL_EDIMS;
    if (L < 0) return WL_EINVAL_L;
    if (!sufficientpoweroftwo(dims[0], L) || !sufficientpoweroftwo(dims[1], L)) return WL_EINVAL_SIZE;
    if (y == x) return WL_EALIAS;
    BoxSpec b;
    b.nd = 3; b.nt = 2;
    b.dims[0] = dims[0]; b.dims[1] = dims[1]; b.dims[2] = nimages;
    b.full.s[0] = 1; b.full.s[1] = dims[0]; b.full.s[2] = image_stride;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    // images in groups of at most 65535 (one grid row / plane per image in the batched kernels)
    const size_t es = dtype == WL_F32 ? 4 : 8;
    for (int64_t i0 = 0; i0 < nimages; i0 += 65535) {
        b.dims[2] = (nimages - i0 < 65535) ? (nimages - i0) : 65535;
        char *yy = (char *)y + (size_t)i0 * image_stride * es;
        const char *xx = (const char *)x + (size_t)i0 * image_stride * es;
        int rc = dtype == WL_F32 ? dwt_filter_impl<float>(ctx, st, b, (float *)yy, (const float *)xx, qmf, flen, L, fw)
                                 : dwt_filter_impl<double>(ctx, st, b, (double *)yy, (const double *)xx, qmf, flen, L, fw);
        if (rc) return rc;
    }
    return WL_OK;
}

int wl_wpt_filter_full(wl_ctx *ctx, int dtype, void *y, const void *x, int64_t n, const double *qmf, int flen, int L, int fw, void *stream)
{
    if (!ctx || !y || !x || !qmf) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (flen < 2 || flen > WL_MAX_FLEN) return WL_EINVAL_FILTER;
    if (n < 1) return WL_EDIMS;
    if (y == x) return WL_EALIAS;
    if (L < 0 || L > wl_maxtransformlevels(n)) return WL_EINVAL_L;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WL_F32) {
        Taps<float> t; make_taps<float>(qmf, flen, t);
        return wpt_impl<float>(ctx, st, (float *)y, (const float *)x, n, &t, nullptr, nullptr, 0, -1, fw, L);
    }
    Taps<double> t; make_taps<double>(qmf, flen, t);
    return wpt_impl<double>(ctx, st, (double *)y, (const double *)x, n, &t, nullptr, nullptr, 0, -1, fw, L);
}

int wl_wpt_lifting_full(wl_ctx *ctx, int dtype, void *y, int64_t n,
                        int nsteps, const int32_t *step_is_update, const int32_t *step_ncoef,
                        const int32_t *step_shift, const double *coefs_flat, double norm1, double norm2,
                        int L, int fw, void *stream)
{
    if (!ctx || !y) return WL_EINVAL_ARG;
    if (dtype != WL_F32 && dtype != WL_F64) return WL_EINVAL_DTYPE;
    if (n < 1) return WL_EDIMS;
    if (L < 0 || L > wl_maxtransformlevels(n)) return WL_EINVAL_L;
    WL_SCOPE(ctx);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == WL_F32) {
        LiftScheme<float> sc;
        int rc = make_scheme<float>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, fw, sc);
        if (rc) return rc;
        return wpt_impl<float>(ctx, st, (float *)y, (const float *)y, n, nullptr, &sc, nullptr, 0, -1, fw, L);
    }
    LiftScheme<double> sc;
    int rc = make_scheme<double>(nsteps, step_is_update, step_ncoef, step_shift, coefs_flat, norm1, norm2, fw, sc);
    if (rc) return rc;
    return wpt_impl<double>(ctx, st, (double *)y, (const double *)y, n, nullptr, &sc, nullptr, 0, -1, fw, L);
}

}  // extern "C"
