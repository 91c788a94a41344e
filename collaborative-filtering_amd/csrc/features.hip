// Item-feature normalisation on the device: the arithmetic of the reference's
// scripts/prepare_features.py (`_row_l1` / `_row_l2` :95-106, `_col_zscore` :109-116, `_col_minmax` :119-124,
// `normalize_feature` :131-201, `_impute_col_median_inplace` :82-92) on a float64 [n][d] matrix in HBM, result
// float32 [n][d].
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

// --- median imputation (prepare_features.py:82-92): NaN / +-inf -> the median of the column's finite entries -------
// Order statistic k of a column by an MSB-first radix select on order-preserving 64-bit keys: eight passes of
// one byte, a 256-bin LDS histogram per pass.  One workgroup per (column, which): which = 0 selects element
// (m - 1) / 2, which = 1 element m / 2 of the m finite entries in ascending order; numpy's median is the mean of
// the two (the same element twice for odd m).  sel[which * d + j] = value, cnt[j] = m.
__device__ __forceinline__ unsigned long long order_key(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(256)
void k_feat_col_select(int64_t n, int d, const double* __restrict__ X, double* __restrict__ sel,
                       long long* __restrict__ cnt) {
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix;
    __shared__ long long s_k;
    const int j = blockIdx.x, which = blockIdx.y, t = threadIdx.x;
    unsigned long long prefix = 0;          // the key's bytes above the current one
    long long k = 0;
    for (int byte = 7; byte >= 0; --byte) {
        hist[t] = 0;
        __syncthreads();
        for (int64_t i = t; i < n; i += 256) {
            const double v = X[i * d + j];
            if (!(fabs(v) <= 1.7976931348623157e308)) continue;                 // NaN / inf: not a candidate
            const unsigned long long key = order_key(v);
            if (byte == 7 || (key >> (8 * (byte + 1))) == prefix) atomicAdd(&hist[(key >> (8 * byte)) & 255], 1u);
        }
        __syncthreads();
        if (t == 0) {
            if (byte == 7) {
                long long m = 0;
                for (int b = 0; b < 256; ++b) m += hist[b];
                if (which == 0) cnt[j] = m;
                k = (m == 0) ? -1 : (which == 0 ? (m - 1) / 2 : m / 2);
            } else {
                k = s_k;
            }
            int b = 0;
            if (k >= 0) {
                long long acc = 0;
                for (; b < 256; ++b) {
                    if (acc + hist[b] > k) break;
                    acc += hist[b];
                }
                k -= acc;
            }
            s_k = k;
            s_prefix = (prefix << 8) | (unsigned long long)(b & 255);
        }
        __syncthreads();
        prefix = s_prefix;
        k = s_k;
        __syncthreads();
    }
    if (t == 0) sel[which * d + j] = (k < 0) ? 0.0 : key_value(prefix);
}

__global__ __launch_bounds__(256)
void k_feat_fill_median(int64_t total, int d, double* __restrict__ X, const double* __restrict__ sel,
                        const long long* __restrict__ cnt) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const double v = X[e];
    if (fabs(v) <= 1.7976931348623157e308) return;
    const int j = (int)(e % d);
    // mean of the two middle elements as numpy forms it: (a + b) / 2; an all-missing column gets 0
    X[e] = cnt[j] > 0 ? __ddiv_rn(__dadd_rn(sel[j], sel[d + j]), 2.0) : 0.0;
}

}  // namespace

extern "C" int als_impute_col_median(int64_t n, int d, double* X, double* work, void* stream) {
    if (n < 1 || d < 1 || !X || !work) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    double* sel = work;                                     // [2][d]
    long long* cnt = reinterpret_cast<long long*>(work + 2 * d);    // [d]
    hipLaunchKernelGGL(k_feat_col_select, dim3((unsigned)d, 2), dim3(256), 0, st, n, d, X, sel, cnt);
    const int64_t total = n * d;
    hipLaunchKernelGGL(k_feat_fill_median, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, d, X, sel, cnt);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

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
