// Microbenchmark: the exact 3-way bf16 split of an fp32 value, x = h + m + l (h, m = top 16 bits by truncation),
// with the remainders formed (a) by v_and + v_sub (what K1 did in round 1) and (b) by v_dot2c_f32_bf16 with a
// (-1, 0) / (0, -1) constant on the packed high halves:  r0 = x0 - H.lo,  r1 = x1 - H.hi  in ONE instruction each.
//   1. exactness: both forms must give bit-identical remainders for random floats over many binades
//      (incl. tiny values: a dot unit that flushed denormals or truncated the addend would show here);
//   2. issue rate: cycles per instruction per SIMD for v_dot2c_f32_bf16 against v_sub_f32, 1 / 2 / 3 waves.
// Build: hipcc -O3 --offload-arch=gfx950 split_dot2c.hip -o split_dot2c ; run: ./split_dot2c
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int pack_hi16(float hi_src, float lo_src) {
    return __builtin_amdgcn_perm(__float_as_int(hi_src), __float_as_int(lo_src), 0x07060302);
}
__device__ __forceinline__ float sub_lo(int packed, float x) {    // x - bf16(packed.lo)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, 0x0000BF80), x, false);
}
__device__ __forceinline__ float sub_hi(int packed, float x) {    // x - bf16(packed.hi)
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, packed), __builtin_bit_cast(bf16x2, (int)0xBF800000), x, false);
}

__global__ void k_exact(const float* __restrict__ x, int n, unsigned long long* bad) {
    const int i = 2 * (blockIdx.x * blockDim.x + threadIdx.x);
    if (i + 1 >= n) return;
    const float x0 = x[i], x1 = x[i + 1];
    // reference form
    const float h0 = __int_as_float(__float_as_int(x0) & 0xFFFF0000), h1 = __int_as_float(__float_as_int(x1) & 0xFFFF0000);
    const float r0 = x0 - h0, r1 = x1 - h1;
    const float m0 = __int_as_float(__float_as_int(r0) & 0xFFFF0000), m1 = __int_as_float(__float_as_int(r1) & 0xFFFF0000);
    const float l0 = r0 - m0, l1 = r1 - m1;
    // dot2c form
    const int H = pack_hi16(x1, x0);
    const float s0 = sub_lo(H, x0), s1 = sub_hi(H, x1);
    const int M = pack_hi16(s1, s0);
    const float t0 = sub_lo(M, s0), t1 = sub_hi(M, s1);
    const bool ok = __float_as_int(s0) == __float_as_int(r0) && __float_as_int(s1) == __float_as_int(r1) &&
                    __float_as_int(t0) == __float_as_int(l0) && __float_as_int(t1) == __float_as_int(l1) &&
                    M == pack_hi16(r1, r0);
    if (!ok) atomicAdd(bad, 1ull);
}

template <int MODE>
__global__ void k_rate(float* out, int iters) {
    float x0 = threadIdx.x * 1e-3f + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    int p = __float_as_int(x0 * 1e-4f);
    float c = 1e-7f;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0)
            asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n"
                         "v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(c));
        else
            asm volatile("v_dot2c_f32_bf16 %0, -1.0, %8\n v_dot2c_f32_bf16 %1, -1.0, %8\n v_dot2c_f32_bf16 %2, -1.0, %8\n v_dot2c_f32_bf16 %3, -1.0, %8\n"
                         "v_dot2c_f32_bf16 %4, -1.0, %8\n v_dot2c_f32_bf16 %5, -1.0, %8\n v_dot2c_f32_bf16 %6, -1.0, %8\n v_dot2c_f32_bf16 %7, -1.0, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(p));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
    const int n = 1 << 24;
    std::vector<float> h(n);
    srand(7);
    for (int i = 0; i < n; ++i) {      // random mantissas over ~80 binades, both signs, some exact bf16 values and zeros
        const unsigned man = ((unsigned)rand() << 9) ^ (unsigned)rand();
        const unsigned ex = 127 - 60 + rand() % 80;
        unsigned bits = ((unsigned)(rand() & 1) << 31) | (ex << 23) | (man & 0x7FFFFF);
        if (i % 97 == 0) bits &= 0xFFFF0000u;
        if (i % 1009 == 0) bits = 0;
        if (i % 5003 == 0) bits = (bits & 0x80000000u) | (1u << 23) | (man & 0x7FFFFF);      // smallest binade: remainders are denormal
        memcpy(&h[i], &bits, 4);
    }
    float* dx; unsigned long long* dbad; float* dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dout, 1 << 22);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(k_exact, dim3(n / 2 / 256), dim3(256), 0, 0, dx, n, dbad);
    unsigned long long bad = 0;
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost);
    printf("exactness: %llu of %d pairs differ between and/sub and dot2c remainders\n", bad, n / 2);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    const double mhz = prop.clockRate / 1e3;
    for (int waves = 1; waves <= 3; ++waves)
        for (int mode = 0; mode < 2; ++mode) {
            const int iters = 200000, threads = 256 * waves;          // `waves` waves per SIMD, one workgroup per CU
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(ncu), dim3(threads), 0, 0, dout, iters);
                else hipLaunchKernelGGL(k_rate<1>, dim3(ncu), dim3(threads), 0, 0, dout, iters);
                hipEventRecord(b); hipEventSynchronize(b);
            }
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s, %d wave(s)/SIMD: %.2f cycles per instruction per SIMD (at %.0f MHz nominal)\n",
                   mode ? "v_dot2c_f32_bf16" : "v_sub_f32       ", waves, ms * 1e-3 * mhz * 1e6 / (8.0 * iters * waves), mhz);
        }
    return bad != 0;
}
