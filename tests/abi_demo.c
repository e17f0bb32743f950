/* tests/abi_demo.c -- the boundary is plain C: this file is compiled with gcc (no HIP, no C++) against
 * include/wavelets_mi355x.h and linked with libwavelets_mi355x.so by tests/test_abi.py.  Without a device it only
 * exercises the host-side entry points and checks that context creation fails loudly (no CPU fallback). */
#include <stdio.h>
#include <string.h>
#include "wavelets_mi355x.h"

int main(void)
{
    wl_ctx *ctx = NULL;
    int64_t dims[3] = {8192, 8192, 1};
    if (wl_version() != WL_VERSION) return 1;
    if (wl_maxtransformlevels(8192) != 13 || wl_maxtransformlevels(40) != 3 || wl_maxmodwttransformlevels(129) != 7) return 2;
    if (strcmp(wl_strerror(WL_EALIAS), "in array is out array") != 0) return 3;
    {   /* fast filter-bank path of the 8192 x 8192 configuration: two approximation buffers of N/4 elements */
        size_t wsb = wl_workspace_bytes(WL_F32, 2, dims, 13);
        if (wsb < (size_t)2 * (8192 * 8192 / 4) * 4 || wsb > ((size_t)129 << 20)) return 4;
    }
    int rc = wl_ctx_create(0, &ctx);
    if (rc == WL_OK) {                       /* a gfx950 device is present: the context works, then goes away */
        if (!ctx || wl_ctx_set_path(ctx, 0) != WL_OK || strcmp(wl_last_kernel(ctx), "none") != 0) return 5;
        if (wl_ctx_set_option(ctx, "WL_TJ", 128) != WL_OK || wl_ctx_clear_options(ctx) != WL_OK || wl_ctx_workspace_held(ctx) != 0) return 9;
        if (wl_dwt_filter(ctx, WL_F32, NULL, NULL, 1, dims, NULL, 8, 1, 1, NULL) != WL_EINVAL_ARG) return 6;
        if (wl_ctx_destroy(ctx) != WL_OK) return 7;
        printf("abi_demo: device context ok\n");
    } else {
        if (rc != WL_ENODEVICE || ctx != NULL) return 8;
        printf("abi_demo: no gfx950 device -> WL_ENODEVICE (no CPU fallback)\n");
    }
    return 0;
}
