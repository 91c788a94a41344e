// K0 / K7 / K7': Z composition and predictions.
//
//   als_compose_z     Z = V + X W                 (reference scripts/als.py:262-281)
//   als_predict_at    U_u.Z_i + mu + b_u + b_i at (u,i) pairs
//                     (how callers read predict(): scripts/tune_params.py:165-166)
//   als_predict_dense U Z^T + mu + b_u + b_i      (scripts/als.py:574)
#include "als_device.hpp"
#include "als_hip.h"

namespace {

__global__ __launch_bounds__(256)
void k_compose_z(int64_t n, int ld, int D, const float* __restrict__ V, const float* __restrict__ X,
                 const float* __restrict__ W, float* __restrict__ Z) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n * ld) return;
    const int64_t i = e / ld;
    const int c = (int)(e - i * ld);
    float acc = V[e];
    for (int a = 0; a < D; ++a) acc = fmaf(X[i * D + a], W[(int64_t)a * ld + c], acc);
    Z[e] = acc;
}

template <int KB>
__global__ __launch_bounds__(256)
void k_predict_at(int ld, int64_t npairs, const int32_t* __restrict__ us, const int32_t* __restrict__ is,
                  const float* __restrict__ U, const float* __restrict__ Z, const float* __restrict__ b_u,
                  const float* __restrict__ b_i, const double* __restrict__ mu_p, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int c = lane & 15, q = lane >> 4;
    const float mu = (float)(*mu_p);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; base < npairs; base += nwaves * 4) {
        const int64_t t = base + q;
        const bool ok = t < npairs;
        const int u = ok ? us[t] : 0, i = ok ? is[t] : 0;
        float a[KB], z[KB];
        load_frow<KB>(U + (size_t)u * ld + KB * c, a);
        load_frow<KB>(Z + (size_t)i * ld + KB * c, z);
        float dot = 0.f;
#pragma unroll
        for (int b = 0; b < KB; ++b) dot = fmaf(a[b], z[b], dot);
        dot += __shfl_xor(dot, 1, 64);
        dot += __shfl_xor(dot, 2, 64);
        dot += __shfl_xor(dot, 4, 64);
        dot += __shfl_xor(dot, 8, 64);
        if (ok && c == 0) out[t] = dot + mu + b_u[u] + b_i[i];
    }
}

// One wave: 16 users x (NJB * 16) items with v_mfma_f32_16x16x4_f32.  Lane (c,q)
// keeps the 4*KB contiguous floats [4KB*q, 4KB*q + 4KB) of user row c; step e
// contracts k = 4KB*q + e, the same k on both operands.
template <int KB>
__global__ __launch_bounds__(256)
void k_predict_dense(int ld, int64_t m, int64_t n, const float* __restrict__ U, const float* __restrict__ Z,
                     const float* __restrict__ b_u, const float* __restrict__ b_i,
                     const double* __restrict__ mu_p, float* __restrict__ out) {
    constexpr int NJB = 4;
    constexpr int E = 4 * KB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const float mu = (float)(*mu_p);
    const int64_t row0 = (int64_t)blockIdx.y * 16;
    const int64_t ur = min(row0 + c, m - 1);
    float ua[E];
    load_frow<E>(U + (size_t)ur * ld + E * q, ua);
    float bu[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bu[r] = b_u[min(row0 + 4 * q + r, m - 1)];
    for (int jb = 0; jb < NJB; ++jb) {
        const int64_t col0 = (((int64_t)blockIdx.x * 4 + wave) * NJB + jb) * 16;
        if (col0 >= n) break;
        const int64_t ir = min(col0 + c, n - 1);
        float za[E];
        load_frow<E>(Z + (size_t)ir * ld + E * q, za);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[e], za[e], acc, 0, 0, 0);
        const float bi = b_i[ir];
        if (col0 + c < n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t rr = row0 + 4 * q + r;
                if (rr < m) out[rr * n + col0 + c] = acc[r] + mu + bu[r] + bi;
            }
        }
    }
}

}  // namespace

extern "C" int als_compose_z(int64_t n, int ld, int D, const float* V, const float* X,
                             const float* W, float* Z, void* stream) {
    if (n < 0 || ld <= 0 || D < 0 || !V || !Z || (D > 0 && (!X || !W))) return ALS_E_BADARG;
    if (n == 0) return 0;
    const int64_t nblk = (n * ld + 255) / 256;
    hipLaunchKernelGGL(k_compose_z, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, n, ld, D, V, X, W, Z);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

extern "C" int als_predict_at(int k, int ld, int64_t npairs, const int32_t* us, const int32_t* is,
                              const float* U, const float* Z, const float* b_u, const float* b_i,
                              const double* mu, float* out, void* stream) {
    const int kp = als_padded_k(k);
    if (kp < 0) return ALS_E_BADK;
    if (ld != kp || npairs < 0 || !U || !Z || !b_u || !b_i || !mu) return ALS_E_BADARG;
    if (npairs == 0) return 0;
    if (!us || !is || !out) return ALS_E_BADARG;
    int64_t nblk = (npairs + 15) / 16;
    if (nblk > 8192) nblk = 8192;
    hipStream_t st = (hipStream_t)stream;
#define ALS_PA_CASE(KB) \
    case KB: hipLaunchKernelGGL(k_predict_at<KB>, dim3((unsigned)nblk), dim3(256), 0, st, ld, npairs, us, is, U, Z, b_u, b_i, mu, out); break;
    switch (ld / 16) {
        ALS_PA_CASE(1) ALS_PA_CASE(2) ALS_PA_CASE(3) ALS_PA_CASE(4) ALS_PA_CASE(5)
        ALS_PA_CASE(6) ALS_PA_CASE(7) ALS_PA_CASE(8) ALS_PA_CASE(9) ALS_PA_CASE(10)
        default: return ALS_E_BADK;
    }
#undef ALS_PA_CASE
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

extern "C" int als_predict_dense(int k, int ld, int64_t m, int64_t n, const float* U, const float* Z,
                                 const float* b_u, const float* b_i, const double* mu, float* out,
                                 void* stream) {
    const int kp = als_padded_k(k);
    if (kp < 0) return ALS_E_BADK;
    if (ld != kp || m <= 0 || n <= 0 || !U || !Z || !b_u || !b_i || !mu || !out) return ALS_E_BADARG;
    const dim3 grid((unsigned)((n + 16 * 16 - 1) / (16 * 16)), (unsigned)((m + 15) / 16));
    if (grid.y > 65535u * 32u) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
#define ALS_PD_CASE(KB) \
    case KB: hipLaunchKernelGGL(k_predict_dense<KB>, grid, dim3(256), 0, st, ld, m, n, U, Z, b_u, b_i, mu, out); break;
    switch (ld / 16) {
        ALS_PD_CASE(1) ALS_PD_CASE(2) ALS_PD_CASE(3) ALS_PD_CASE(4) ALS_PD_CASE(5)
        ALS_PD_CASE(6) ALS_PD_CASE(7) ALS_PD_CASE(8) ALS_PD_CASE(9) ALS_PD_CASE(10)
        default: return ALS_E_BADK;
    }
#undef ALS_PD_CASE
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
