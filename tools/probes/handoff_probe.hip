// Scratch microbenchmark (not product code): what does a chain of DEPENDENT small phases cost
//   (a) as separate kernel launches on one stream (what the library did up to round 3),
//   (b) as ONE launch in which the last workgroup to finish a group of phase-p tiles continues with the phase-(p+1)
//       tile that depends on them ("last arriver continues": device-scope counter, release before / acquire after; no
//       workgroup ever waits for another one),
//   (c) as ONE persistent launch with a device-wide barrier between phases?
// The phases have the data-movement skeleton of the tail of an 8192^2 2-D transform (2048^2 block -> ... -> 1):
//   phase 0: 1024 tiles, each reads 16 KiB, writes 16 KiB of "details" + 1 KiB of "approximation"   (2048^2 -> 512^2)
//   phase 1:   64 tiles, the same on the 1 MiB approximation of phase 0                              (512^2 -> 128^2)
//   phase 2:    1 tile: reads the 64 KiB approximation of phase 1, NP dependent LDS passes, writes 64 KiB  (128^2 -> 1)
// No arithmetic to speak of: the numbers are floors for the launch / hand-over structure, not kernel timings.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d (%s) at line %d\n", e, hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct P {
    const float4 *src;      // 16 MiB
    float4 *y;              // 16 MiB + 1 MiB + 64 KiB
    float4 *b0;             // 1 MiB
    float4 *b1;             // 64 KiB
    unsigned *cnt;          // [0..63]: arrivals per phase-1 tile, [64]: arrivals at phase 2
    int np;                 // LDS passes of phase 2
    int acq_all;            // every wave runs the acquire fence (1) or thread 0 only (0)
};

// ---- (d): the hand-over data (the 1/16 "approximation") travels with sc1 stores / loads (write-through to, and read from,
// the memory side: visible to the other XCDs' L2s without a release fence = without writing back a whole L2) ----
typedef float F4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1(float4 *p, float4 v)
{
    F4v w = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ float4 ld_sc1(const float4 *p)
{
    F4v w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
    return make_float4(w.x, w.y, w.z, w.w);
}
template <bool IN_SC1>
__device__ __forceinline__ void tile_phase_sc1(const float4 *in, float4 *yout, float4 *bout, int tid)
{
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = IN_SC1 ? ld_sc1(in + tid + 256 * r) : in[tid + 256 * r];
    float4 s;
    s.x = v[0].x + v[1].x + v[2].x + v[3].x; s.y = v[0].y + v[1].y + v[2].y + v[3].y;
    s.z = v[0].z + v[1].z + v[2].z + v[3].z; s.w = v[0].w + v[1].w + v[2].w + v[3].w;
#pragma unroll
    for (int r = 0; r < 4; ++r) yout[tid + 256 * r] = v[r];
    if (tid < 64) st_sc1(bout + tid, s);
}
__device__ __forceinline__ bool arrive_sc1(unsigned *c, unsigned expected, int tid)
{
    __shared__ unsigned last1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's sc1 stores have reached the memory side
    __syncthreads();
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last1 = (old == expected - 1) ? 1u : 0u;
    }
    __syncthreads();
    return last1 != 0;
}

__device__ __forceinline__ void tile_phase(const float4 *in, float4 *yout, float4 *bout, int tid)
{
    float4 v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = in[tid + 256 * r];
    float4 s;
    s.x = v[0].x + v[1].x + v[2].x + v[3].x; s.y = v[0].y + v[1].y + v[2].y + v[3].y;
    s.z = v[0].z + v[1].z + v[2].z + v[3].z; s.w = v[0].w + v[1].w + v[2].w + v[3].w;
#pragma unroll
    for (int r = 0; r < 4; ++r) yout[tid + 256 * r] = v[r];
    if (tid < 64) bout[tid] = s;
}

__device__ __forceinline__ void tail_phase(const float4 *in, float4 *yout, int np, int tid, float *lds)
{
    // 64 KiB = 4096 float4 -> 16 per thread
    float4 v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = in[tid + 256 * r];
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
    for (int p = 0; p < np; ++p) {
        lds[tid] = acc;
        __syncthreads();
        acc = lds[(tid + 1) & 255] * 0.5f + lds[(tid + 3) & 255] * 0.25f;
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[r].x += acc; yout[tid + 256 * r] = v[r]; }
}

__global__ void __launch_bounds__(256) k_p0(P a) { tile_phase(a.src + (size_t)blockIdx.x * 1024, a.y + (size_t)blockIdx.x * 1024, a.b0 + (size_t)blockIdx.x * 64, threadIdx.x); }
__global__ void __launch_bounds__(256) k_p1(P a) { tile_phase(a.b0 + (size_t)blockIdx.x * 1024, a.y + (size_t)(1024 + blockIdx.x) * 1024, a.b1 + (size_t)blockIdx.x * 64, threadIdx.x); }
__global__ void __launch_bounds__(256) k_p2(P a) { __shared__ float lds[256]; tail_phase(a.b1, a.y + (size_t)(1024 + 64) * 1024, a.np, threadIdx.x, lds); }

// release: make this workgroup's stores visible device-wide, then count; returns true in every thread of the LAST arriver
__device__ __forceinline__ bool arrive(unsigned *c, unsigned expected, int tid, int acq_all)
{
    __shared__ unsigned last;
    __syncthreads();                                   // every wave's stores have left the CU (vmcnt(0) + barrier)
    if (tid == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);       // agent scope: write back L2
        const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (old == expected - 1) ? 1u : 0u;
        if (old == expected - 1 && !acq_all) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    }
    __syncthreads();
    const bool l = last != 0;
    if (l && acq_all) __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return l;
}

__global__ void __launch_bounds__(256) k_chain(P a)
{
    __shared__ float lds[256];
    const int tid = threadIdx.x;
    const unsigned t0 = blockIdx.x;
    tile_phase(a.src + (size_t)t0 * 1024, a.y + (size_t)t0 * 1024, a.b0 + (size_t)t0 * 64, tid);
    const unsigned t1 = t0 >> 4;
    if (!arrive(a.cnt + t1, 16, tid, a.acq_all)) return;
    tile_phase(a.b0 + (size_t)t1 * 1024, a.y + (size_t)(1024 + t1) * 1024, a.b1 + (size_t)t1 * 64, tid);
    if (!arrive(a.cnt + 64, 64, tid, a.acq_all)) return;
    tail_phase(a.b1, a.y + (size_t)(1024 + 64) * 1024, a.np, tid, lds);
    if (tid <= 64) a.cnt[tid] = 0;                     // self-cleaning: the next launch starts from zero
}

__global__ void __launch_bounds__(256) k_chain_sc1(P a)
{
    __shared__ float lds[256];
    const int tid = threadIdx.x;
    const unsigned t0 = blockIdx.x;
    tile_phase_sc1<false>(a.src + (size_t)t0 * 1024, a.y + (size_t)t0 * 1024, a.b0 + (size_t)t0 * 64, tid);
    const unsigned t1 = t0 >> 4;
    if (!arrive_sc1(a.cnt + t1, 16, tid)) return;
    tile_phase_sc1<true>(a.b0 + (size_t)t1 * 1024, a.y + (size_t)(1024 + t1) * 1024, a.b1 + (size_t)t1 * 64, tid);
    if (!arrive_sc1(a.cnt + 64, 64, tid)) return;
    {
        float4 v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ld_sc1(a.b1 + tid + 256 * r);
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += v[r].x + v[r].y + v[r].z + v[r].w;
        for (int p = 0; p < a.np; ++p) {
            lds[tid] = acc;
            __syncthreads();
            acc = lds[(tid + 1) & 255] * 0.5f + lds[(tid + 3) & 255] * 0.25f;
            __syncthreads();
        }
        float4 *yout = a.y + (size_t)(1024 + 64) * 1024;
#pragma unroll
        for (int r = 0; r < 16; ++r) { v[r].x += acc; yout[tid + 256 * r] = v[r]; }
    }
    if (tid <= 64) __hip_atomic_store(a.cnt + tid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// persistent variant: G workgroups, device-wide barrier between phases (sense = phase number * G)
__device__ __forceinline__ void grid_barrier(unsigned *c, unsigned target, int tid)
{
    __syncthreads();
    if (tid == 0) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
}
__global__ void __launch_bounds__(256) k_persist(P a, unsigned base)
{
    __shared__ float lds[256];
    const int tid = threadIdx.x;
    const unsigned G = gridDim.x;
    for (unsigned t0 = blockIdx.x; t0 < 1024; t0 += G) tile_phase(a.src + (size_t)t0 * 1024, a.y + (size_t)t0 * 1024, a.b0 + (size_t)t0 * 64, tid);
    grid_barrier(a.cnt + 65, base + G, tid);
    for (unsigned t1 = blockIdx.x; t1 < 64; t1 += G) tile_phase(a.b0 + (size_t)t1 * 1024, a.y + (size_t)(1024 + t1) * 1024, a.b1 + (size_t)t1 * 64, tid);
    grid_barrier(a.cnt + 65, base + 2 * G, tid);
    if (blockIdx.x == 0) tail_phase(a.b1, a.y + (size_t)(1024 + 64) * 1024, a.np, tid, lds);
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 300;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    P a;
    float4 *src, *y, *b0, *b1; unsigned *cnt;
    CK(hipMalloc(&src, 16 << 20)); CK(hipMalloc(&y, (16 << 20) + (1 << 20) + (64 << 10))); CK(hipMalloc(&b0, 1 << 20)); CK(hipMalloc(&b1, 64 << 10));
    CK(hipMalloc(&cnt, 1024)); CK(hipMemset(cnt, 0, 1024)); CK(hipMemset(src, 0, 16 << 20));
    a.src = src; a.y = y; a.b0 = b0; a.b1 = b1; a.cnt = cnt; a.np = 12; a.acq_all = 1;
    float ms;
    for (int pass = 0; pass < 2; ++pass) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) {
            hipLaunchKernelGGL(k_p0, dim3(1024), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_p1, dim3(64), dim3(256), 0, st, a);
            hipLaunchKernelGGL(k_p2, dim3(1), dim3(256), 0, st, a);
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) three launches:              %.2f us per chain\n", ms * 1000.f / reps);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_p0, dim3(1024), dim3(256), 0, st, a);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("    phase 0 alone:               %.2f us\n", ms * 1000.f / reps);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_p1, dim3(64), dim3(256), 0, st, a);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("    phase 1 alone:               %.2f us\n", ms * 1000.f / reps);
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_p2, dim3(1), dim3(256), 0, st, a);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("    phase 2 alone:               %.2f us\n", ms * 1000.f / reps);
        for (int acq = 0; acq < 2; ++acq) {
            a.acq_all = acq;
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_chain, dim3(1024), dim3(256), 0, st, a);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned c[65]; CK(hipMemcpy(c, cnt, sizeof(c), hipMemcpyDeviceToHost));
            unsigned bad = 0; for (int i = 0; i < 65; ++i) bad += c[i];
            printf("(b) one launch, last arriver (acquire by %s): %.2f us per chain (counters left: %u)\n", acq ? "every wave" : "thread 0", ms * 1000.f / reps, bad);
        }
        {
            CK(hipMemset(cnt, 0, 1024));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_chain_sc1, dim3(1024), dim3(256), 0, st, a);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned c[65]; CK(hipMemcpy(c, cnt, sizeof(c), hipMemcpyDeviceToHost));
            unsigned bad = 0; for (int i = 0; i < 65; ++i) bad += c[i];
            printf("(d) one launch, last arriver, sc1 hand-over data, no fences: %.2f us per chain (counters left: %u)\n", ms * 1000.f / reps, bad);
        }
        for (int G : {256, 1024}) {
            CK(hipMemsetAsync(cnt + 65, 0, 4, st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, st, a, (unsigned)(i * 2 * G + 0));
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("(c) one launch, %4d WGs, grid barriers: %.2f us per chain\n", G, ms * 1000.f / reps);
        }
    }
    return 0;
}
