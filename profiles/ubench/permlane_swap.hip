// What v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) do to the four 16-lane rows of a wave, and what a
// dependent chain of them costs next to ds_bpermute.   hipcc -O3 --offload-arch=gfx950 permlane_swap.hip -o permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_sem(unsigned* out) {
    const unsigned lane = threadIdx.x;
    const unsigned x = 100 * (lane >> 4) + (lane & 15);           // row id * 100 + position
    const unsigned y = 1000 + x;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    auto s = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1]; out[128 + lane] = s[0]; out[192 + lane] = s[1];
}
__global__ void k_time(float* out, int n, int mode) {
    float v = out[threadIdx.x];
    const int addr = (threadIdx.x & 15) << 2;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        if (mode == 0) {
            v = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))) + 1.0f;
        } else {
            unsigned u = __float_as_uint(v);
            auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            auto s = __builtin_amdgcn_permlane32_swap(r[0], r[0], false, false);
            v = __uint_as_float(s[0]) + 1.0f;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = v;
    if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / n;
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 4);
    k_sem<<<1, 64>>>(d);
    unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane16_swap(x,y)[0]", "permlane16_swap(x,y)[1]", "permlane32_swap(x,y)[0]", "permlane32_swap(x,y)[1]"};
    for (int a = 0; a < 4; ++a) {
        printf("%s rows:", names[a]);
        for (int r = 0; r < 4; ++r) printf(" %u..%u", h[a * 64 + 16 * r], h[a * 64 + 16 * r + 15]);
        printf("\n");
    }
    float* f; hipMalloc(&f, 128 * 4); hipMemset(f, 0, 128 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        k_time<<<1, 64>>>(f, 4096, mode);
        float o[65]; hipMemcpy(o, f, sizeof(o), hipMemcpyDeviceToHost);
        printf("%s: %.1f cycles per dependent replicate + add\n", mode ? "permlane16_swap + permlane32_swap" : "ds_bpermute", o[64]);
    }
    return 0;
}
