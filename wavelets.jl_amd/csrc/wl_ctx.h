// wl_ctx.h -- the context object behind the C ABI (one per device + stream) and the helpers shared by
// the translation units that implement entry points (wl_api.hip, wl_ext.hip).
#pragma once
#include "wl_internal.h"

struct wl_ctx {
    int device = 0;
    void *ws = nullptr;                 // grow-only transform workspace
    size_t ws_bytes = 0;
    void *aux = nullptr;                // small persistent block: order-statistic selection state (wl_ext.hip)
    int last_hip = 0;
    int path = 0;                       // 0 auto, 1 generic only
    const char *last_kernel = "none";
    int cu_count = 256;
};

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

// grow the workspace to at least `bytes` (synchronises the device when it has to reallocate)
int wl_ensure_ws(wl_ctx *ctx, size_t bytes);
