// Microbenchmark: do v_mfma_f32_16x16x4_f32 and plain VALU (v_fma_f32) overlap on one SIMD?
// One workgroup per CU; waves 0..3 land on SIMDs 0..3, waves 4..7 pair up with them.
//   mode 0: every wave runs MFMA only        mode 1: every wave runs VALU only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU   (needs 512 threads)
// Prints cycles per instruction per SIMD derived from wall time and an assumed clock read back
// from s_memtime / s_memrealtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#ifndef USE_BF16
#define USE_BF16 0
#endif

template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* clk) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = (MODE == 0) || (MODE == 2 && wave < 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x * 1e-3f + 1.f, b = 1.0001f;
#if USE_BF16
    if (do_mfma) {
        bf16x8 av, bv;
        for (int j = 0; j < 8; ++j) { av[j] = (short)(0x3f80 + threadIdx.x + j); bv[j] = (short)(0x3f80 + j); }
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c7, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    } else
#endif
    if (do_mfma) {
        f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c5, 0, 0, 0);
            c6 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c6, 0, 0, 0);
            c7 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c7, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
    } else {
        float x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
        for (int i = 0; i < iters * 8; ++i) {           // 64 FMAs per outer MFMA-iteration equivalent
            asm volatile("v_fma_f32 %0, %9, %8, %0\n v_fma_f32 %1, %9, %8, %1\n v_fma_f32 %2, %9, %8, %2\n v_fma_f32 %3, %9, %8, %3\n"
                         "v_fma_f32 %4, %9, %8, %4\n v_fma_f32 %5, %9, %8, %5\n v_fma_f32 %6, %9, %8, %6\n v_fma_f32 %7, %9, %8, %7\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int MODE>
void run(const char* name, int threads, int iters, float* out, unsigned long long* clk) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double ghz = (double)h[0] / (double)h[1] * 0.1;      // memrealtime ticks at 100 MHz
    double cyc = ms * 1e-3 * ghz * 1e9;
    int wps = threads / 256;                              // waves per SIMD
    printf("%-34s threads=%4d waves/SIMD=%d  %.3f ms  clk=%.2f GHz  cycles=%.3g", name, threads, wps, ms, ghz, cyc);
    if (MODE == 0) printf("  -> %.1f cyc per MFMA per SIMD\n", cyc / (8.0 * iters * wps));
    if (MODE == 1) printf("  -> %.2f cyc per v_fma per SIMD\n", cyc / (64.0 * iters * wps));
    if (MODE == 2) printf("  -> per SIMD: %d MFMA + %d v_fma; if serial expect sum of the two single-mode times\n", 8 * iters, 64 * iters);
}

int main() {
    float* out; unsigned long long* clk;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&clk, 16);
    const int iters = 20000;
    run<0>(USE_BF16 ? "MFMA bf16 16x16x32 only" : "MFMA f32 16x16x4 only", 256, iters, out, clk);
    run<0>(USE_BF16 ? "MFMA bf16 16x16x32 only" : "MFMA f32 16x16x4 only", 512, iters, out, clk);
    run<1>("VALU v_fma_f32 only", 256, iters, out, clk);
    run<1>("VALU v_fma_f32 only", 512, iters, out, clk);
    run<1>("VALU v_fma_f32 only", 1024, iters, out, clk);
    run<2>("mixed: 1 MFMA wave + 1 VALU wave/SIMD", 512, iters, out, clk);
    return 0;
}
