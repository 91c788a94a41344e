// Item-feature normalisation on the device: the arithmetic of the reference's
// scripts/prepare_features.py (`_row_l1` / `_row_l2` :95-106, `_col_zscore` :109-116, `_col_minmax` :119-124,
// `normalize_feature` :131-201) on a float64 [n][d] matrix in HBM, result float32 [n][d].
//
// The reference computes in numpy float64 and casts once; the kernels reproduce numpy's summation ORDER so that
// the float32 outputs are bitwise those of the reference (tests/test_gpu_features.py, fixtures written by the
// unmodified reference):
//   * a sum along the contiguous axis (row sizes; column statistics when d == 1) is numpy's pairwise sum: fewer
//     than 8 terms one after the other, up to 128 terms eight interleaved partial sums combined as
//     ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus the tail, longer runs split in halves (multiples of 8) recursively;
//   * a sum along the other axis (column statistics, d >= 2) adds the rows one after the other;
//   * mean = sum / n, var = sum((x - mean)^2) / n, products and sums rounded separately (no fused multiply-add).
// O(n d) once per data set - nowhere near a roofline; one thread per row / per column keeps the order exact.
#include "als_device.hpp"
#include "als_hip.h"

namespace {

enum { M_NONE = 0, M_ROW_L1 = 1, M_ROW_L2 = 2, M_COL_ZSCORE = 3, M_COL_MINMAX = 4 };

// the term of element x in the sum: T = 0 plain, 1 |x|, 2 x^2, 3 (x - c)^2
template <int T>
__device__ __forceinline__ double term(double x, double c) {
    if (T == 1) return fabs(x);
    if (T == 2) return __dmul_rn(x, x);
    if (T == 3) { const double t = __dsub_rn(x, c); return __dmul_rn(t, t); }
    return x;
}

// numpy's pairwise sum of n terms a[0], a[stride], ...
template <int T>
__device__ double np_pairwise(const double* a, int64_t n, int64_t stride, double c) {
    if (n < 8) {
        double res = -0.0;
        for (int64_t i = 0; i < n; ++i) res = __dadd_rn(res, term<T>(a[i * stride], c));
        return res;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = term<T>(a[j * stride], c);
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __dadd_rn(r[j], term<T>(a[(i + j) * stride], c));
        }
        double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                               __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __dadd_rn(res, term<T>(a[i * stride], c));
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    const double lo = np_pairwise<T>(a, n2, stride, c);
    const double hi = np_pairwise<T>(a + n2 * stride, n - n2, stride, c);
    return __dadd_rn(lo, hi);
}

// sum over the rows of column j: numpy adds row after row when d >= 2 and sums pairwise when the column is the
// whole (contiguous) array
template <int T>
__device__ double col_sum(const double* X, int64_t n, int d, int j, double c) {
    if (d == 1) return __dadd_rn(0.0, np_pairwise<T>(X, n, 1, c));
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s = __dadd_rn(s, term<T>(X[i * d + j], c));
    return s;
}

__global__ __launch_bounds__(256)
void k_feat_finite(int64_t total, const double* __restrict__ X, int32_t* __restrict__ status) {
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const double v = X[e];
        bad = bad || !(fabs(v) <= 1.7976931348623157e308);      // NaN or +-inf
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(status, 1);
}

__global__ __launch_bounds__(256)
void k_feat_cast(int64_t total, const double* __restrict__ X, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < total) out[e] = (float)X[e];
}

// rows to unit L1 / L2 size, sizes below eps clamped (prepare_features.py:95-106)
template <int T>
__global__ __launch_bounds__(64)
void k_feat_rows(int64_t n, int d, const double* __restrict__ X, double eps, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const double* row = X + i * d;
    double size = __dadd_rn(0.0, np_pairwise<T>(row, d, 1, 0.0));
    if (T == 2) size = __dsqrt_rn(size);
    const double den = fmax(size, eps);
    for (int j = 0; j < d; ++j) out[i * d + j] = (float)__ddiv_rn(row[j], den);
}

// stats[j] = shift, stats[d + j] = divisor of column j
__global__ __launch_bounds__(64)
void k_feat_col_stats(int64_t n, int d, const double* __restrict__ X, int method, double eps,
                      double* __restrict__ stats) {
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= d) return;
    if (method == M_COL_ZSCORE) {       // (x - mean) / std, a (near-)constant column keeps std 1 (:109-116)
        const double mean = __ddiv_rn(col_sum<0>(X, n, d, j, 0.0), (double)n);
        const double var = __ddiv_rn(col_sum<3>(X, n, d, j, mean), (double)n);
        const double sd = __dsqrt_rn(var);
        stats[j] = mean;
        stats[d + j] = (sd < eps) ? 1.0 : sd;
    } else {                            // columns to [0, 1], ranges below eps clamped (:119-124)
        double lo = X[j], hi = X[j];
        for (int64_t i = 1; i < n; ++i) { const double v = X[i * d + j]; lo = fmin(lo, v); hi = fmax(hi, v); }
        stats[j] = lo;
        stats[d + j] = fmax(__dsub_rn(hi, lo), eps);
    }
}

__global__ __launch_bounds__(256)
void k_feat_col_apply(int64_t total, int d, const double* __restrict__ X, const double* __restrict__ stats,
                      int zero_nonfinite, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int j = (int)(e % d);
    double v = __ddiv_rn(__dsub_rn(X[e], stats[j]), stats[d + j]);
    if (zero_nonfinite && !(fabs(v) <= 1.7976931348623157e308)) v = 0.0;
    out[e] = (float)v;
}

}  // namespace

extern "C" int als_normalize_features(int64_t n, int d, const double* X, int method, double eps, float* out,
                                      double* colwork, int32_t* status, void* stream) {
    if (n < 1 || d < 1 || !X || !out || !status || method < M_NONE || method > M_COL_MINMAX) return ALS_E_BADARG;
    if ((method == M_COL_ZSCORE || method == M_COL_MINMAX) && !colwork) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int64_t total = n * d;
    const unsigned eb = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_feat_finite, dim3(eb < 4096 ? eb : 4096), dim3(256), 0, st, total, X, status);
    switch (method) {
        case M_NONE: hipLaunchKernelGGL(k_feat_cast, dim3(eb), dim3(256), 0, st, total, X, out); break;
        case M_ROW_L1:
            hipLaunchKernelGGL(k_feat_rows<1>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, d, X, eps, out);
            break;
        case M_ROW_L2:
            hipLaunchKernelGGL(k_feat_rows<2>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, n, d, X, eps, out);
            break;
        default:
            hipLaunchKernelGGL(k_feat_col_stats, dim3((unsigned)((d + 63) / 64)), dim3(64), 0, st, n, d, X, method, eps,
                               colwork);
            hipLaunchKernelGGL(k_feat_col_apply, dim3(eb), dim3(256), 0, st, total, d, X, colwork,
                               method == M_COL_ZSCORE ? 1 : 0, out);
    }
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
