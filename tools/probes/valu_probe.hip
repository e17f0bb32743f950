// Scratch microbenchmark (not product code): issue cost of the VALU forms the 2-D DWT kernels can be written in.
//   scalar  v_mul_f32 + v_add_f32            (one sample per lane per instruction)
//   packed  v_pk_mul_f32 + v_pk_add_f32      (two samples per lane per instruction)
//   dpp     v_mul_f32_dpp (wave_shr:1 folded into the multiply) + v_add_f32
//   movdpp  v_mov_b32_dpp + v_mul_f32 + v_add_f32   (what the packed kernels have to do for neighbour rows)
//   fma     v_fma_f32                         (for scale)
// Every variant runs the same number of multiply-add pairs per lane; waves per SIMD = 1, 2, 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %d at %d\n", e, __LINE__); return 1; } } while (0)

#define REP8(X) X X X X X X X X

template <int MODE>
__global__ void __launch_bounds__(64) k_valu(float *out, int iters, float t0)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float x0 = 1.f + threadIdx.x * 1e-3f, x1 = x0 * 1.1f, x2 = x0 * 1.2f, x3 = x0 * 1.3f, x4 = x0 * 1.4f, x5 = x0 * 1.5f, x6 = x0 * 1.6f, x7 = x0 * 1.7f;
    float p0, p1, p2, p3, p4, p5, p6, p7;
    float t = t0;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 A0 = {a0, a1}, A1 = {a2, a3}, A2 = {a4, a5}, A3 = {a6, a7}, X0 = {x0, x1}, X1 = {x2, x3}, X2 = {x4, x5}, X3 = {x6, x7}, T = {t, t}, P0, P1, P2, P3;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {          // scalar: 8 mul + 8 add per block, x8
            REP8(asm volatile(
                "v_mul_f32 %0, %16, %8\n v_mul_f32 %1, %16, %9\n v_mul_f32 %2, %16, %10\n v_mul_f32 %3, %16, %11\n"
                "v_mul_f32 %4, %16, %12\n v_mul_f32 %5, %16, %13\n v_mul_f32 %6, %16, %14\n v_mul_f32 %7, %16, %15\n"
                "v_add_f32 %8, %8, %0\n v_add_f32 %9, %9, %1\n v_add_f32 %10, %10, %2\n v_add_f32 %11, %11, %3\n"
                "v_add_f32 %12, %12, %4\n v_add_f32 %13, %13, %5\n v_add_f32 %14, %14, %6\n v_add_f32 %15, %15, %7\n"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                  "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(t));)
        } else if constexpr (MODE == 1) {   // packed: 4 pk_mul + 4 pk_add per block (same 8 mul-add pairs per lane), x8
            REP8(asm volatile(
                "v_pk_mul_f32 %0, %8, %4\n v_pk_mul_f32 %1, %8, %5\n v_pk_mul_f32 %2, %8, %6\n v_pk_mul_f32 %3, %8, %7\n"
                "v_pk_add_f32 %4, %4, %0\n v_pk_add_f32 %5, %5, %1\n v_pk_add_f32 %6, %6, %2\n v_pk_add_f32 %7, %7, %3\n"
                : "=&v"(P0), "=&v"(P1), "=&v"(P2), "=&v"(P3), "+v"(X0), "+v"(X1), "+v"(X2), "+v"(X3) : "v"(T));)
        } else if constexpr (MODE == 2) {   // dpp folded into the multiply
            REP8(asm volatile(
                "v_mul_f32_dpp %0, %8, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %9, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %2, %10, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %11, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %4, %12, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %13, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %6, %14, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %15, %16 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32 %8, %8, %0\n v_add_f32 %9, %9, %1\n v_add_f32 %10, %10, %2\n v_add_f32 %11, %11, %3\n"
                "v_add_f32 %12, %12, %4\n v_add_f32 %13, %13, %5\n v_add_f32 %14, %14, %6\n v_add_f32 %15, %15, %7\n"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                  "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(t));)
        } else if constexpr (MODE == 3) {   // separate dpp move, then scalar mul + add
            REP8(asm volatile(
                "v_mov_b32_dpp %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %9 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %2, %10 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %11 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %4, %12 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %13 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mov_b32_dpp %6, %14 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %15 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32 %0, %16, %0\n v_mul_f32 %1, %16, %1\n v_mul_f32 %2, %16, %2\n v_mul_f32 %3, %16, %3\n"
                "v_mul_f32 %4, %16, %4\n v_mul_f32 %5, %16, %5\n v_mul_f32 %6, %16, %6\n v_mul_f32 %7, %16, %7\n"
                "v_add_f32 %8, %8, %0\n v_add_f32 %9, %9, %1\n v_add_f32 %10, %10, %2\n v_add_f32 %11, %11, %3\n"
                "v_add_f32 %12, %12, %4\n v_add_f32 %13, %13, %5\n v_add_f32 %14, %14, %6\n v_add_f32 %15, %15, %7\n"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                  "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(t));)
        } else if constexpr (MODE == 4) {   // fma
            REP8(asm volatile(
                "v_fma_f32 %0, %8, %0, %0\n v_fma_f32 %1, %8, %1, %1\n v_fma_f32 %2, %8, %2, %2\n v_fma_f32 %3, %8, %3, %3\n"
                "v_fma_f32 %4, %8, %4, %4\n v_fma_f32 %5, %8, %5, %5\n v_fma_f32 %6, %8, %6, %6\n v_fma_f32 %7, %8, %7, %7\n"
                : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(t));)
        } else if constexpr (MODE == 5) {   // packed fma
            REP8(asm volatile(
                "v_pk_fma_f32 %0, %4, %0, %0\n v_pk_fma_f32 %1, %4, %1, %1\n v_pk_fma_f32 %2, %4, %2, %2\n v_pk_fma_f32 %3, %4, %3, %3\n"
                : "+v"(X0), "+v"(X1), "+v"(X2), "+v"(X3) : "v"(T));)
        } else if constexpr (MODE == 6) {   // scalar with the tap in an SGPR
            REP8(asm volatile(
                "v_mul_f32 %0, %16, %8\n v_mul_f32 %1, %16, %9\n v_mul_f32 %2, %16, %10\n v_mul_f32 %3, %16, %11\n"
                "v_mul_f32 %4, %16, %12\n v_mul_f32 %5, %16, %13\n v_mul_f32 %6, %16, %14\n v_mul_f32 %7, %16, %15\n"
                "v_add_f32 %8, %8, %0\n v_add_f32 %9, %9, %1\n v_add_f32 %10, %10, %2\n v_add_f32 %11, %11, %3\n"
                "v_add_f32 %12, %12, %4\n v_add_f32 %13, %13, %5\n v_add_f32 %14, %14, %6\n v_add_f32 %15, %15, %7\n"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                  "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "s"(t0));)
        } else if constexpr (MODE == 7) {   // row_shr:1 instead of wave_shr:1
            REP8(asm volatile(
                "v_mul_f32_dpp %0, %8, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %9, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %2, %10, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %11, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %4, %12, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %13, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_mul_f32_dpp %6, %14, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %15, %16 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                "v_add_f32 %8, %8, %0\n v_add_f32 %9, %9, %1\n v_add_f32 %10, %10, %2\n v_add_f32 %11, %11, %3\n"
                "v_add_f32 %12, %12, %4\n v_add_f32 %13, %13, %5\n v_add_f32 %14, %14, %6\n v_add_f32 %15, %15, %7\n"
                : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                  "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(t));)
        }
    }
    float r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + X0.x + X0.y + X1.x + X1.y + X2.x + X2.y + X3.x + X3.y + A0.x + A1.x + A2.x + A3.x;
    if (r == 123.456f) out[threadIdx.x] = r;
}

template <typename F> float timeit(F f, int reps = 10) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) f();
    std::vector<float> v;
    for (int i = 0; i < reps; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); v.push_back(ms); }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2] * 1e3f;
}

int main() {
    float *out; CK(hipMalloc(&out, 4096));
    const int iters = 4000;
    const char *names[] = {"scalar mul+add (128 instr/iter, 64 madd pairs)", "packed pk_mul+pk_add (64 instr/iter, 64 madd pairs)",
                           "dpp-folded mul + add (128 instr/iter)", "mov_dpp + mul + add (192 instr/iter)", "fma (64 instr/iter, 64 fma)",
                           "pk_fma (32 instr/iter, 64 fma)", "scalar mul+add, tap in SGPR (128 instr/iter)", "row_shr dpp-folded mul + add (128 instr/iter)"};
    const int ninstr[] = {128, 64, 128, 192, 64, 32, 128, 128};
    for (int wps : {1, 2, 4}) {
        const int grid = 256 * 4 * wps;
        for (int m = 0; m < 8; ++m) {
            float us = 0;
            switch (m) {
            case 0: us = timeit([&] { k_valu<0><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 1: us = timeit([&] { k_valu<1><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 2: us = timeit([&] { k_valu<2><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 3: us = timeit([&] { k_valu<3><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 4: us = timeit([&] { k_valu<4><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 5: us = timeit([&] { k_valu<5><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 6: us = timeit([&] { k_valu<6><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            case 7: us = timeit([&] { k_valu<7><<<grid, 64>>>(out, iters, 1.0001f); }); break;
            }
            // cycles per instruction per SIMD at a nominal 2.4 GHz (the real clock under this load may be lower)
            const double cyc = us * 1e-6 * 2.4e9 / ((double)iters * ninstr[m] * wps);
            printf("waves/SIMD %d  %-55s %8.1f us  %5.2f cyc/instr  %6.2f cyc per madd pair\n", wps, names[m], us, cyc, cyc * ninstr[m] / 64.0);
        }
    }
    return 0;
}
