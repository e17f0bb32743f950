// Scratch microbenchmark (not product code): what HBM copy rates does this MI355X reach for the
// access shapes the 2-D DWT kernel uses?  256 MiB in, 256 MiB out per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

__global__ void copy_f4(const float4 *in, float4 *out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void copy_f4_unroll4(const float4 *in, float4 *out, size_t n4) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        float4 a = in[i], b = in[i + stride], c = in[i + 2 * stride], d = in[i + 3 * stride];
        out[i] = a; out[i + stride] = b; out[i + 2 * stride] = c; out[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) out[i] = in[i];
}
__global__ void copy_f4_st2(const float4 *in, float2 *out, size_t n4) {   // 16-B loads, 8-B stores
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = in[i];
        // two 8-byte stores into two different halves of the output (like the quadrant streams)
        out[i] = make_float2(v.x, v.y);
        out[n4 + i] = make_float2(v.z, v.w);
    }
}
// column-marching waves: wave w owns rows [240w-8, 240w+248) of an 8192x8192 column-major matrix and
// walks TJ columns; per column one float4 load per lane, per two columns four float2 stores (lanes 2..61)
__global__ void __launch_bounds__(64) march(const float *in, float *out, int n, int TJ, int nstrips, int st8, int doload, int vrows) {
    int lane = threadIdx.x;
    unsigned b = blockIdx.x, nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
    unsigned logical = xcd * q8 + (xcd < r8 ? xcd : r8) + (b >> 3);
    int strip = logical % nstrips, chunk = logical / nstrips;
    long gi = (long)strip * vrows + (lane - 2) * 4;
    long row = gi < 0 ? gi + n : (gi >= n ? gi - n : gi);
    bool valid = lane >= 2 && lane < 2 + vrows / 4 && gi < n;
    long ko = gi >> 1, hm = n / 2, nxj = n / 2;
    long j0 = (long)chunk * TJ;
    for (int t = 0; t < TJ / 2; ++t) {
        float4 a = make_float4(1.f, 2.f, 3.f, (float)t), c = make_float4(5.f, 6.f, 7.f, (float)t);
        if (doload) {
            a = *(const float4 *)(in + (j0 + 2 * t) * n + row);
            c = *(const float4 *)(in + (j0 + 2 * t + 1) * n + row);
        }
        if (valid) {
            long k = j0 / 2 + t;
            if (st8 == 2) {   // even lanes: rows ko..ko+3 of streams 0,1 ; odd lanes: rows ko-2..ko+1 of streams 2,3
                bool odd = lane & 1;
                float *p0 = odd ? (out + (nxj + k) * n + ko - 2) : (out + k * n + ko);
                float *p1 = odd ? (out + (nxj + k) * n + hm + ko - 2) : (out + k * n + hm + ko);
                *(float4 *)p0 = a;
                *(float4 *)p1 = c;
            } else if (st8) {
                *(float2 *)(out + k * n + ko) = make_float2(a.x, a.y);
                *(float2 *)(out + k * n + hm + ko) = make_float2(a.z, a.w);
                *(float2 *)(out + (nxj + k) * n + ko) = make_float2(c.x, c.y);
                *(float2 *)(out + (nxj + k) * n + hm + ko) = make_float2(c.z, c.w);
            } else {   // same bytes, two 16-byte stores into two streams
                *(float4 *)(out + (2 * k) * n + 2 * ko) = a;
                *(float4 *)(out + (2 * k + 1) * n + 2 * ko) = c;
            }
        }
    }
}

__global__ void write_f4(float4 *out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void read_f4(const float4 *in, float *out, size_t n4) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
template <typename F> float timeit(F f, int reps = 30) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    std::vector<float> v;
    for (int i = 0; i < reps; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); v.push_back(ms); }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2] * 1e3f;
}
int main() {
    const int n = 8192; const size_t N = (size_t)n * n;
    float *in, *out; CK(hipMalloc(&in, N * 4)); CK(hipMalloc(&out, N * 4));
    CK(hipMemset(in, 1, N * 4)); CK(hipMemset(out, 0, N * 4));
    const double bytes = 2.0 * N * 4;
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        float us = timeit([&] { copy_f4<<<g, 256>>>((const float4 *)in, (float4 *)out, N / 4); });
        printf("copy_f4          grid %5d: %7.1f us  %6.0f GB/s\n", g, us, bytes / us / 1e3);
    }
    for (int g : {512, 1024, 2048, 4096}) {
        float us = timeit([&] { copy_f4_unroll4<<<g, 256>>>((const float4 *)in, (float4 *)out, N / 4); });
        printf("copy_f4_unroll4  grid %5d: %7.1f us  %6.0f GB/s\n", g, us, bytes / us / 1e3);
    }
    for (int g : {2048, 8192}) {
        float us = timeit([&] { copy_f4_st2<<<g, 256>>>((const float4 *)in, (float2 *)out, N / 4); });
        printf("copy_f4_st2      grid %5d: %7.1f us  %6.0f GB/s\n", g, us, bytes / us / 1e3);
    }
    for (int g : {2048, 8192, 32768}) {
        float us = timeit([&] { write_f4<<<g, 256>>>((float4 *)out, N / 4); });
        printf("write_f4         grid %5d: %7.1f us  %6.0f GB/s (256 MiB written)\n", g, us, bytes / 2 / us / 1e3);
        us = timeit([&] { read_f4<<<g, 256>>>((const float4 *)in, out, N / 4); });
        printf("read_f4          grid %5d: %7.1f us  %6.0f GB/s (256 MiB read)\n", g, us, bytes / 2 / us / 1e3);
    }
    for (int TJ : {128})
      for (int vrows : {240, 192})
        for (int doload : {1, 0})
          for (int st8 : {1, 0, 2}) {
            int nstrips = (n + vrows - 1) / vrows, nchunks = n / TJ;
            float us = timeit([&] { march<<<nstrips * nchunks, 64>>>(in, out, n, TJ, nstrips, st8, doload, vrows); });
            printf("march TJ %3d vrows %d load %d st %s waves %5d: %7.1f us\n", TJ, vrows, doload, st8 == 1 ? "4x8B " : (st8 == 0 ? "2x16B" : "eo16B"), nstrips * nchunks, us);
          }
    return 0;
}
