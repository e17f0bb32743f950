#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    int lane = threadIdx.x;
    int v = lane * 10;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false); // wave_shl:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false); // wave_shr:1
    int c = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false); // wave_rol:1
    int d = __builtin_amdgcn_update_dpp(-1, v, 0x13c, 0xf, 0xf, false); // wave_ror:1
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c; out[192 + lane] = d;
}
int main() {
    int *d; hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    int h[256]; hipMemcpy(h, d, 256 * 4, hipMemcpyDeviceToHost);
    const char *nm[4] = {"wave_shl1", "wave_shr1", "wave_rol1", "wave_ror1"};
    for (int r = 0; r < 4; ++r) { printf("%s:", nm[r]); for (int i = 0; i < 64; ++i) printf(" %d", h[r * 64 + i]); printf("\n"); }
    return 0;
}
