// wl_ctx.h -- the context object behind the C ABI (one per device + stream) and the helpers shared by
// the translation units that implement entry points (wl_api.hip, wl_ext.hip).
#pragma once
#include "wl_internal.h"

struct wl_ctx {
    int device = 0;
    void *ws = nullptr;                 // grow-only transform workspace
    size_t ws_bytes = 0;
    bool ws_pooled = false;             // allocated from the stream-ordered pool (hipMallocAsync): released with hipFreeAsync
    void *aux = nullptr;                // small persistent block: order-statistic selection state (wl_ext.hip)
    void *sync = nullptr;               // wl::kSyncWords zeroed words: in-launch hand-over flags (wl::tl_sync)
    int last_hip = 0;
    int path = 0;                       // 0 auto, 1 generic only
    const char *last_kernel = "none";
    int cu_count = 256;
    wl::Opts opts;                      // wl_ctx_set_option
    // pinned host staging for small host arguments that a kernel reads from device memory (the node bits of a partially split
    // packet tree): the caller's buffer is copied here synchronously, the copy to the device is asynchronous.  A ring of slots,
    // each guarded by an event, so that a call only ever waits for the copy issued kStage calls earlier.
    enum { kStage = 4 };
    void *stage[kStage] = {nullptr, nullptr, nullptr, nullptr};
    size_t stage_bytes[kStage] = {0, 0, 0, 0};
    hipEvent_t stage_ev[kStage] = {nullptr, nullptr, nullptr, nullptr};
    int stage_next = 0;
};
// copy `bytes` host bytes to device memory `dst` on stream st without synchronising the stream (wl_api.hip)
int wl_stage_to_device(wl_ctx *ctx, void *dst, const void *host, size_t bytes, hipStream_t st);

// Installed by every ABI entry point for the duration of the call: makes the context's device current (and restores
// the caller's device on exit, so a call never changes the process's current device) and publishes the context's
// option table to the launchers.
struct CallScope {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    const wl::Opts *prev_opts;
    unsigned *prev_sync;
    explicit CallScope(wl_ctx *ctx) : prev_opts(wl::tl_opts), prev_sync(wl::tl_sync)
    {
        wl::tl_opts = &ctx->opts;
        wl::tl_sync = static_cast<unsigned *>(ctx->sync);
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != ctx->device) {
            err = hipSetDevice(ctx->device);
            switched = (err == hipSuccess);
        }
    }
    ~CallScope()
    {
        wl::tl_opts = prev_opts;
        wl::tl_sync = prev_sync;
        if (switched) (void)hipSetDevice(prev);
    }
    CallScope(const CallScope &) = delete;
    CallScope &operator=(const CallScope &) = delete;
};
#define WL_SCOPE(ctx)                                   \
    CallScope scope__(ctx);                             \
    if (scope__.err != hipSuccess) return hip_fail((ctx), scope__.err)

inline int hip_fail(wl_ctx *ctx, hipError_t e)
{
    if (ctx) ctx->last_hip = (int)e;
    return WL_EHIP;
}
#define WL_HIP(ctx, expr)                                  \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return hip_fail((ctx), e__); \
    } while (0)

// grow the workspace to at least `bytes`: stream-ordered on `st` (no synchronisation) when `ordered`, otherwise with a device
// synchronisation (wl_ctx_reserve)
int wl_ensure_ws(wl_ctx *ctx, size_t bytes, hipStream_t st = nullptr, bool ordered = false);

// lifting transform of a box with a direction-adjusted scheme / makescheme (wl_api.hip; used by wl_ext.hip)
template <typename T>
int wl_lifting_box(wl_ctx *ctx, hipStream_t st, const wl::BoxSpec &b, T *y, const T *x, const wl::LiftScheme<T> &sc, int L, int fw);
template <typename T>
int wl_make_scheme(int nsteps, const int32_t *is_update, const int32_t *ncoef, const int32_t *shift, const double *coefs, double norm1,
                   double norm2, int fw, wl::LiftScheme<T> &sc);
