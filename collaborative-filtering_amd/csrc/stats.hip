// K6: mu / train-RMSE / norm reductions with fp64 accumulation.
//
// Replaces reference scripts/als.py:503-517.  One pass over the CSR ratings
// computes d = r - (U_u.Z_i + b_u + b_i + mu_old) per rating and accumulates
// sum(d), sum(d^2) in double.  The host then forms
//     mu_new = mu_old + sum(d)/N,  rmse = sqrt(sum(d^2)/N - (sum(d)/N)^2)
// which equals the reference's two-step mean / RMSE ("Stats").
// Reductions are two-stage and atomic-free, so results are run-to-run
// reproducible (the early-stop rule compares RMSE differences of 1e-4).
#include "als_device.hpp"
#include "als_hip.h"

namespace {

constexpr int STAT_WPW = 4;
constexpr int SUMSQ_BLOCKS = 1024;

template <int KB>
__global__ __launch_bounds__(64 * STAT_WPW)
void k_residual_stats(int ld, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                      const float* __restrict__ vals, const float* __restrict__ U,
                      const float* __restrict__ Z, const float* __restrict__ b_u,
                      const float* __restrict__ b_i, const double* __restrict__ mu_p,
                      const als_task* __restrict__ tasks, int64_t ntasks,
                      double* __restrict__ partials) {
    __shared__ double red[2 * STAT_WPW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, q = lane >> 4;
    const int64_t tid = (int64_t)blockIdx.x * STAT_WPW + wave;
    double sd = 0.0, sd2 = 0.0;
    if (tid < ntasks) {
        const als_task t = tasks[tid];
        const int row = t.row;
        const int64_t beg = indptr[row] + (int64_t)t.seg * ALS_SPLIT_CHUNK;
        const int len = (int)min((int64_t)ALS_SPLIT_CHUNK, indptr[row + 1] - beg);
        const float mu = (float)(*mu_p);
        const float bu = b_u[row];
        float u[KB];
        load_frow<KB>(U + (size_t)row * ld + KB * c, u);
        const int32_t* idxp = indices + beg;
        const float* valp = vals + beg;
        for (int base = 0; base < len; base += 64) {
            const int tl = base + lane;
            const bool ok = tl < len;
            const int idx_l = ok ? idxp[tl] : 0;
            const float rb_l = ok ? (valp[tl] - mu - bu - b_i[idx_l]) : 0.f;
            const int nsteps = (min(64, len - base) + 3) >> 2;
            for (int s = 0; s < nsteps; ++s) {
                const int src = 4 * s + q;
                const int idx_t = bperm_i(idx_l, src);
                const float rb_t = bperm_f(rb_l, src);
                float z[KB];
                load_frow<KB>(Z + (size_t)idx_t * ld + KB * c, z);
                float dot = 0.f;
#pragma unroll
                for (int b = 0; b < KB; ++b) dot = fmaf(u[b], z[b], dot);
                dot += __shfl_xor(dot, 1, 64);
                dot += __shfl_xor(dot, 2, 64);
                dot += __shfl_xor(dot, 4, 64);
                dot += __shfl_xor(dot, 8, 64);
                if (c == 0 && base + src < len) {
                    const double d = (double)(rb_t - dot);
                    sd += d;
                    sd2 += d * d;
                }
            }
        }
    }
    sd = wave_sum_d(sd);
    sd2 = wave_sum_d(sd2);
    if (lane == 0) { red[2 * wave] = sd; red[2 * wave + 1] = sd2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < STAT_WPW; ++w) { a += red[2 * w]; b += red[2 * w + 1]; }
        partials[2 * (int64_t)blockIdx.x] = a;
        partials[2 * (int64_t)blockIdx.x + 1] = b;
    }
}

// out[j] = sum_i partials[i*stride + j], j < stride <= 2; fixed summation order
__global__ __launch_bounds__(256)
void k_reduce_final(const double* __restrict__ partials, int64_t n, int stride, double* __restrict__ out) {
    __shared__ double red[256 * 2];
    double acc[2] = {0.0, 0.0};
    for (int64_t i = threadIdx.x; i < n; i += 256)
        for (int j = 0; j < stride; ++j) acc[j] += partials[i * stride + j];
    red[threadIdx.x] = acc[0];
    red[256 + threadIdx.x] = acc[1];
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[threadIdx.x] += red[threadIdx.x + o];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) for (int j = 0; j < stride; ++j) out[j] = red[256 * j];
}

__global__ __launch_bounds__(256)
void k_sumsq_partial(const float* __restrict__ x, int64_t n, double* __restrict__ partials) {
    __shared__ double red[256];
    double acc = 0.0;
    const int64_t n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = x4[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const double v = x[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256)
void k_sum_pairs_partial(const float* __restrict__ x, int64_t npairs, double* __restrict__ partials) {
    __shared__ double red[512];
    double a = 0.0, b = 0.0;
    const f32x2* x2 = reinterpret_cast<const f32x2*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * 256) {
        const f32x2 v = x2[i];
        a += (double)v.x;
        b += (double)v.y;
    }
    red[threadIdx.x] = a;
    red[256 + threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) {
            red[threadIdx.x] += red[threadIdx.x + o];
            red[256 + threadIdx.x] += red[256 + threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = red[0]; partials[2 * blockIdx.x + 1] = red[256]; }
}

// The history row of an iteration in two launches (scripts/als.py:505-517): sums of squares of U, V, b_u, b_i
// (blockIdx.y picks the array, same per-array partition as als_sumsq) ...
struct Norm4 { const float* x[4]; int64_t n[4]; int nblk[4]; };

__global__ __launch_bounds__(256)
void k_sumsq4_partial(const Norm4 a, double* __restrict__ partials) {
    __shared__ double red[256];
    const int j = blockIdx.y;
    if ((int)blockIdx.x >= a.nblk[j]) return;
    const float* x = a.x[j];
    const int64_t n = a.n[j], n4 = n >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)a.nblk[j] * 256) {
        const f32x4 v = x4[i];
        acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const double v = x[(n4 << 2) + threadIdx.x];
        acc += v * v;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[(int64_t)j * SUMSQ_BLOCKS + blockIdx.x] = red[0];
}

// ... then the final sums and the scalar arithmetic of the row: mu += mean(d), RMSE = sqrt(mean(d^2) - mean(d)^2),
// the four norms.  fp64, no contraction (the same operations the host would do one by one).
__global__ __launch_bounds__(256)
void k_history_final(const Norm4 a, const double* __restrict__ partials, const double* __restrict__ stats, double nnz,
                     double* __restrict__ mu, double* __restrict__ row) {
#pragma clang fp contract(off)
    __shared__ double red[4][256];
    for (int j = 0; j < 4; ++j) {
        double acc = 0.0;
        for (int i = threadIdx.x; i < a.nblk[j]; i += 256) acc += partials[(int64_t)j * SUMSQ_BLOCKS + i];
        red[j][threadIdx.x] = acc;
    }
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int j = 0; j < 4; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mean_d = stats[0] / nnz;
        const double m = *mu + mean_d;
        const double var = stats[1] / nnz - mean_d * mean_d;
        *mu = m;
        row[0] = sqrt(var < 0.0 ? 0.0 : var);          // (a NaN - no ratings at all - stays a NaN, as numpy's mean of nothing)
        for (int j = 0; j < 4; ++j) row[1 + j] = sqrt(red[j][0]);
        row[5] = m;
    }
}

template <int KB>
int launch_stats(int ld, const int64_t* indptr, const int32_t* indices, const float* vals,
                 const float* U, const float* Z, const float* b_u, const float* b_i,
                 const double* mu, const als_task* tasks, int64_t ntasks, double* partials,
                 double* out, hipStream_t st) {
    const int64_t nblk = (ntasks + STAT_WPW - 1) / STAT_WPW;
    hipLaunchKernelGGL(k_residual_stats<KB>, dim3((unsigned)nblk), dim3(64 * STAT_WPW), 0, st, ld, indptr,
                       indices, vals, U, Z, b_u, b_i, mu, tasks, ntasks, partials);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, partials, nblk, 2, out);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

extern "C" int als_residual_stats(int k, int ld, const int64_t* indptr, const int32_t* indices,
                                  const float* vals, const float* U, const float* Z,
                                  const float* b_u, const float* b_i, const double* mu,
                                  const als_task* tasks, int64_t ntasks, double* partials,
                                  double* out, void* stream) {
    const int kp = als_padded_k(k);
    if (kp < 0) return ALS_E_BADK;
    if (ld != kp || !indptr || !indices || !vals || !U || !Z || !b_u || !b_i || !mu || !tasks ||
        ntasks <= 0 || !partials || !out)
        return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
#define ALS_STATS_CASE(KB) \
    case KB: return launch_stats<KB>(ld, indptr, indices, vals, U, Z, b_u, b_i, mu, tasks, ntasks, partials, out, st);
    switch (ld / 16) {
        ALS_STATS_CASE(1) ALS_STATS_CASE(2) ALS_STATS_CASE(3) ALS_STATS_CASE(4) ALS_STATS_CASE(5)
        ALS_STATS_CASE(6) ALS_STATS_CASE(7) ALS_STATS_CASE(8) ALS_STATS_CASE(9) ALS_STATS_CASE(10)
    }
#undef ALS_STATS_CASE
    return ALS_E_BADK;
}

extern "C" int als_sum_pairs(const float* x, int64_t npairs, double* partials, double* out, void* stream) {
    if (!x || npairs < 0 || !partials || !out || ((uintptr_t)x & 7) != 0) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t nblk = (npairs + 255) / 256;
    if (nblk < 1) nblk = 1;
    if (nblk > SUMSQ_BLOCKS) nblk = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(k_sum_pairs_partial, dim3((unsigned)nblk), dim3(256), 0, st, x, npairs, partials);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, partials, nblk, 2, out);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

extern "C" int als_sumsq_partials(void) { return SUMSQ_BLOCKS; }

extern "C" int als_sumsq(const float* x, int64_t n, double* partials, double* out, void* stream) {
    if (!x || n < 0 || !partials || !out) return ALS_E_BADARG;
    if (((uintptr_t)x & 15) != 0) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    int64_t nblk = (n / 4 + 255) / 256;
    if (nblk < 1) nblk = 1;
    if (nblk > SUMSQ_BLOCKS) nblk = SUMSQ_BLOCKS;
    hipLaunchKernelGGL(k_sumsq_partial, dim3((unsigned)nblk), dim3(256), 0, st, x, n, partials);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, st, partials, nblk, 1, out);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

extern "C" int als_history_row(const float* U, int64_t nU, const float* V, int64_t nV, const float* b_u, int64_t nbu,
                               const float* b_i, int64_t nbi, const double* stats, int64_t nnz, double* mu,
                               double* partials, double* row, void* stream) {
    if (!U || !V || !b_u || !b_i || !stats || !mu || !partials || !row || nnz < 0 || nU < 0 || nV < 0 || nbu < 0 || nbi < 0)
        return ALS_E_BADARG;
    Norm4 a;
    const float* xs[4] = {U, V, b_u, b_i};
    const int64_t ns[4] = {nU, nV, nbu, nbi};
    int maxblk = 1;
    for (int j = 0; j < 4; ++j) {
        if (((uintptr_t)xs[j] & 15) != 0) return ALS_E_BADARG;
        int64_t nblk = (ns[j] / 4 + 255) / 256;
        if (nblk < 1) nblk = 1;
        if (nblk > SUMSQ_BLOCKS) nblk = SUMSQ_BLOCKS;
        a.x[j] = xs[j]; a.n[j] = ns[j]; a.nblk[j] = (int)nblk;
        if (nblk > maxblk) maxblk = (int)nblk;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sumsq4_partial, dim3((unsigned)maxblk, 4), dim3(256), 0, st, a, partials);
    hipLaunchKernelGGL(k_history_final, dim3(1), dim3(256), 0, st, a, partials, stats, (double)nnz, mu, row);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
