// Scratch: the floor of a chain of dependent tiny kernels on one stream (what fusing small levels could save),
// plus the cost of a device-wide barrier inside one persistent kernel (atomic counter in global memory).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 999) *p = 1; }
__global__ void k_small(float *x, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) x[i] = x[i] * 1.0001f + 1.0f; }
__global__ void k_gridbar(unsigned *cnt, int phases, float *x, int n)
{
    const unsigned G = gridDim.x;
    for (int ph = 1; ph <= phases; ++ph) {
        int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(cnt, 1u);
            unsigned spins = 0;
            while (atomicAdd(cnt, 0u) < (unsigned)ph * G && ++spins < (1u << 22)) { __builtin_amdgcn_s_sleep(1); }
        }
        __syncthreads();
    }
}
int main()
{
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float *x; hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20);
    unsigned *cnt; hipMalloc(&cnt, 4);
    float ms;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, st);
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int *)nullptr);
        hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("empty kernel chain (1 WG): %.2f us per launch\n", ms);
        hipEventRecord(e0, st);
        for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(k_small, dim3(256), dim3(256), 0, st, x, 65536);
        hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("small kernel chain (256 WGs x 256 thr, 256 KiB RMW): %.2f us per launch\n", ms);
        for (int G : {64, 256}) {
            hipMemsetAsync(cnt, 0, 4, st);
            hipEventRecord(e0, st);
            hipLaunchKernelGGL(k_gridbar, dim3(G), dim3(256), 0, st, cnt, 100, x, G * 256);
            hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            unsigned c = 0; hipMemcpy(&c, cnt, 4, hipMemcpyDeviceToHost);
            printf("persistent kernel, %d WGs, 100 phases with grid barrier: %.2f us per phase (counter %u, expected %u)\n", G, ms * 10.0f, c, 100u * G);
        }
    }
    return 0;
}
