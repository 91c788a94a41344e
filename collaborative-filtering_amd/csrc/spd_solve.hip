// K5c: dense fp64 Cholesky solve of the W-step normal equations.
//
// Replaces the reference's `cholesky_solve(A, b)` on the (d k) x (d k) system of one feature
// (scripts/als.py:497-500).  N = d k is 64 ... a few thousand, so the factorisation is latency-,
// not flop-bound.  Right-looking blocked Cholesky, 64-column panels, two small launches per panel:
//
//   k_spd_pack       A (+ diag_add I), padded to NP = 64 T with an identity tail, plus one extra
//                    row tile whose first row is b^T: factorising the augmented matrix leaves
//                    y^T = (L^-1 b)^T in that row, i.e. the forward solve is free.
//   k_spd_panel      one workgroup per row tile r > j: the 64x64 diagonal block and the tile are
//                    eliminated together, 4 rows x 8 interleaved columns per thread in registers,
//                    one LDS round trip per pivot, the bulk of a pivot's rank-1 update deferred
//                    behind the publication of the next pivot (the diagonal block is eliminated
//                    redundantly by every workgroup, so nothing separates it from the triangular solve).
//   k_spd_update     one workgroup per trailing tile (r, c): C -= L(r,j) L(c,j)^T on the fp64
//                    matrix cores (v_mfma_f64_16x16x4_f64), operands straight from global memory.
//   k_spd_backsolve  L^T x = y by one workgroup: in-wave 64x64 triangular solves (readlane
//                    broadcasts) and column-parallel updates of the remaining right-hand side.
//
// Everything is fp64 with a fixed summation order (bitwise reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "als_hip.h"

namespace {

constexpr int NB = 64;                 // panel width = tile edge

__device__ __forceinline__ double dpp_xor1(double v) {
    // value of the neighbouring lane (lane ^ 1): quad_perm [1,0,3,2] on both halves
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256)
void k_spd_pack(int64_t N, int64_t NP, int T, const double* __restrict__ A, int64_t lda,
                const double* __restrict__ b, double diag_add, double* __restrict__ M, int32_t* status) {
    const int64_t total = (int64_t)(T + 1) * NB * NP;
    if (blockIdx.x == 0 && threadIdx.x == 0) *status = 0;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / NP, c = idx - r * NP;
        double v = 0.0;
        if (r < N) {
            if (c < N) v = A[r * lda + c] + (r == c ? diag_add : 0.0);
        } else if (r < NP) {
            v = (r == c) ? 1.0 : 0.0;
        } else if (r == NP) {
            if (c < N) v = b[c];
        }
        M[idx] = v;
    }
}

// Panel j for row tile r (r > j): eliminate the diagonal block D = M[j][j] and X = M[r][j] together
// (128 rows x 64 columns).  Thread (rg, g) = (tid >> 3, tid & 7) keeps rows 4 rg .. 4 rg + 3 (rg < 16: rows of
// D, else of X) and the interleaved columns g + 8 ci in registers.  Per pivot t one LDS round trip: the
// owners of column t publish it for all 128 rows together with 1/p and 1/sqrt(p); after the barrier every
// thread reads its 4 row entries and the <= 8 column multipliers it needs (rows 0..63 of the same array).
// The factored diagonal block goes to Ld[j] (not in place: other workgroups may still be loading D).
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void publish_pivot(double p, double* dst, bool report, int32_t* status, int index) {
    if (!(p > 0.0)) { if (report) atomicCAS(status, 0, index + 1); p = 1.0; }
    double y = __builtin_amdgcn_rsq(p);                   // ~2^-26 relative; two Newton steps reach fp64
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = fma(-p * y, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    dst[0] = y * y;                                       // 1 / p
    dst[1] = y;                                           // 1 / sqrt(p)
}

struct PanelCtx { double* cbuf; int32_t* status; int g, rg, row0, j; bool write_diag; };
struct StepRegs { double m[4]; double cv[8]; };      // multipliers and column values of one pivot

// One pivot of the panel, split in two so that the next pivot can be published as early as possible:
//   step_head<t>: after the barrier read 1/p, 1/sqrt(p), this thread's 4 row entries and its <= 8 column
//                 values; finalise column t; update ONLY column t + 1 and publish it with its pivot.
//   step_tail<t>: the remaining rank-1 updates of pivot t.  It is issued after the NEXT barrier, between
//                 the LDS reads of pivot t + 1 and their first use, so it fills that latency instead of
//                 delaying the publication.
// t is a template parameter so that every index into x[][] is static (a 64-trip `#pragma unroll`
// exceeds the unroller's budget and would push x[][] into scratch).
template <int t>
__device__ __forceinline__ void step_reads(const PanelCtx& cx, StepRegs& R, double& dv, double (&rv)[4]) {
    constexpr int CB = 2 * NB + 2;
    constexpr int ct = t >> 3;
    const double* cb = cx.cbuf + (t & 1) * CB;
    const double ip = cb[2 * NB];
    dv = cb[2 * NB + 1];
    const f64x2 r01 = *reinterpret_cast<const f64x2*>(cb + cx.row0);
    const f64x2 r23 = *reinterpret_cast<const f64x2*>(cb + cx.row0 + 2);
    rv[0] = r01.x; rv[1] = r01.y; rv[2] = r23.x; rv[3] = r23.y;
#pragma unroll
    for (int ci = ct; ci < 8; ++ci) R.cv[ci] = cb[cx.g + 8 * ci];
#pragma unroll
    for (int i = 0; i < 4; ++i) R.m[i] = -(rv[i] * ip);
}

template <int t>
__device__ __forceinline__ void step_head(double (&x)[4][8], const PanelCtx& cx, const StepRegs& R, double dv,
                                          const double (&rv)[4]) {
    constexpr int CB = 2 * NB + 2;
    constexpr int gt = t & 7, ct = t >> 3;
    const int g = cx.g;
    if (g == gt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i][ct] = rv[i] * dv;          // final L[row][t]
    }
    if constexpr (t + 1 < NB) {
        constexpr int g1 = (t + 1) & 7, c1 = (t + 1) >> 3;
        double* nb = cx.cbuf + ((t + 1) & 1) * CB;
        if (g == g1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i][c1] = fma(R.m[i], R.cv[c1], x[i][c1]);
            *reinterpret_cast<f64x2*>(nb + cx.row0) = f64x2{x[0][c1], x[1][c1]};
            *reinterpret_cast<f64x2*>(nb + cx.row0 + 2) = f64x2{x[2][c1], x[3][c1]};
            if (cx.rg == (t + 1) / 4)
                publish_pivot(x[(t + 1) & 3][c1], nb + 2 * NB, cx.write_diag, cx.status, cx.j * NB + t + 1);
        }
    }
}

template <int t>
__device__ __forceinline__ void step_tail(double (&x)[4][8], const PanelCtx& cx, const StepRegs& R) {
    constexpr int gt = t & 7, ct = t >> 3;
    constexpr int g1 = (t + 1) & 7, c1 = (t + 1) >> 3;
    const int g = cx.g;
    // columns c = g + 8 ci > t, minus column t + 1 (done in step_head): ci > ct always, ci == ct for g > gt
#pragma unroll
    for (int ci = ct; ci < 8; ++ci) {
        bool on = (ci > ct) || (g > gt);
        if (ci == c1) on = on && (g != g1);
        if (on) {
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i][ci] = fma(R.m[i], R.cv[ci], x[i][ci]);
        }
    }
    // pin the updates here: otherwise the compiler sinks the FMAs of late columns across many barriers and
    // keeps the operands of all of them alive (hundreds of spilled registers)
#pragma unroll
    for (int ci = ct; ci < 8; ++ci)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(x[i][ci]));
}

template <int t>
__device__ __forceinline__ void panel_steps(double (&x)[4][8], const PanelCtx& cx, const StepRegs& prev) {
    __syncthreads();
    StepRegs R;
    double dv, rv[4];
    step_reads<t>(cx, R, dv, rv);
    if constexpr (t > 0) step_tail<t - 1>(x, cx, prev);
    step_head<t>(x, cx, R, dv, rv);
    if constexpr (t + 1 < NB) panel_steps<t + 1>(x, cx, R);
    else step_tail<t>(x, cx, R);
}

__device__ __forceinline__ void panel(double* __restrict__ M, int64_t ld, int j, int r, bool write_diag,
                                   double* __restrict__ Ld, double* __restrict__ cbuf /* [2][CB] LDS */,
                                   int32_t* status) {
    const int tid = threadIdx.x, g = tid & 7, rg = tid >> 3;
    const bool diag = rg < 16;
    const int row0 = 4 * rg;                              // 0..127 inside the stacked [D; X] panel
    const int64_t grow0 = diag ? (int64_t)j * NB + row0 : (int64_t)r * NB + (row0 - NB);
    double* src = M + grow0 * ld + (int64_t)j * NB + g;
    double x[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) x[i][ci] = src[(int64_t)i * ld + 8 * ci];

    if (g == 0) {                                         // publish column 0 and the first pivot
#pragma unroll
        for (int i = 0; i < 4; ++i) cbuf[row0 + i] = x[i][0];
        if (rg == 0) publish_pivot(x[0][0], cbuf + 2 * NB, write_diag, status, j * NB);
    }
    PanelCtx cx{cbuf, status, g, rg, row0, j, write_diag};
    StepRegs none{};
    panel_steps<0>(x, cx, none);
    if (!diag) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) src[(int64_t)i * ld + 8 * ci] = x[i][ci];
    } else if (write_diag) {
        double* dst = Ld + (int64_t)j * NB * NB + row0 * NB + g;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ci = 0; ci < 8; ++ci) dst[i * NB + 8 * ci] = (g + 8 * ci <= row0 + i) ? x[i][ci] : 0.0;
    }
    __syncthreads();
}

// C(r, c) -= L(r, j) L(c, j)^T, one 64x64 tile per call on the fp64 matrix cores: wave w owns block row w
// (16 rows) x 4 block columns.  v_mfma_f64_16x16x4_f64 operands are one double per lane, A[row = lane & 15]
// [k = lane >> 4] and B[k = lane >> 4][col = lane & 15], read straight from global memory (no LDS); the
// result block has col = lane & 15, row = (lane >> 4) + 4 reg.  All loads are issued before the first MFMA.
__device__ __forceinline__ void update_tile(double* __restrict__ M, int64_t ld, int j, int r, int c) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const double* La = M + ((int64_t)r * NB + 16 * wv + l15) * ld + (int64_t)j * NB + l4;
    const double* Lb = M + ((int64_t)c * NB + l15) * ld + (int64_t)j * NB + l4;
    double* C = M + ((int64_t)r * NB + 16 * wv + l4) * ld + (int64_t)c * NB + l15;
    double cold[4][4], a[16], b[4][16];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int i = 0; i < 4; ++i) cold[nb][i] = C[(int64_t)(4 * i) * ld + 16 * nb];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) a[ks] = La[4 * ks];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) b[nb][ks] = Lb[(int64_t)(16 * nb) * ld + 4 * ks];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[nb][ks], acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) C[(int64_t)(4 * i) * ld + 16 * nb] = cold[nb][i] - acc[i];
    }
}

__global__ __launch_bounds__(256)
void k_spd_panel(int64_t ld, double* __restrict__ M, double* __restrict__ Ld, int32_t* status, int j) {
    __shared__ __attribute__((aligned(16))) double cbuf[2 * (2 * NB + 2)];
    panel(M, ld, j, j + 1 + (int)blockIdx.x, blockIdx.x == 0, Ld, cbuf, status);
}

// trailing tiles (r, c) of panel j: j < c < T, c <= r <= T, enumerated column by column
__global__ __launch_bounds__(256)
void k_spd_update(int T, int64_t ld, double* __restrict__ M, int j) {
    int id = blockIdx.x, c = j + 1;
    while (id >= T - c + 1) { id -= T - c + 1; ++c; }
    update_tile(M, ld, j, c + id, c);
}

// L^T x = y; y is row NP of M.  One workgroup of 512 threads; s (the running right-hand side) in LDS.
// Bound by what one CU can stream (the strictly lower triangle once, ~60 GB/s): ~160 us at N = 1216.
__global__ __launch_bounds__(512)
void k_spd_backsolve(int T, int64_t N, int64_t ld, const double* __restrict__ M, const double* __restrict__ Ld,
                     double* __restrict__ x_out) {
    extern __shared__ double lds[];
    double* s = lds;                         // [NP]
    double* xs = lds + ld;                   // [NB]
    constexpr int BT = 512;
    const int tid = threadIdx.x;
    const int64_t NP = ld;
    for (int64_t c = tid; c < NP; c += BT) s[c] = M[NP * ld + c];
    __syncthreads();
    for (int j = T - 1; j >= 0; --j) {
        if (tid < 64) {
            const double* Dj = Ld + (int64_t)j * NB * NB;
            double dcol[NB];                 // column `tid` of the diagonal block, rows 0..63 (coalesced per row)
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) dcol[c2] = Dj[c2 * NB + tid];
            double v = s[j * NB + tid];
            double dd = 1.0;
#pragma unroll
            for (int c2 = 0; c2 < NB; ++c2) dd = (tid == c2) ? dcol[c2] : dd;
            const double rd = 1.0 / dd;
#pragma unroll
            for (int c2 = NB - 1; c2 >= 0; --c2) {
                const double xc = readlane_d(v * rd, c2);
                if (tid < c2) v = fma(-dcol[c2], xc, v);
                if (tid == c2) v = xc;
            }
            xs[tid] = v;
            if ((int64_t)j * NB + tid < N) x_out[(int64_t)j * NB + tid] = v;
        }
        __syncthreads();
        const double* Lrow = M + (int64_t)j * NB * ld;
        for (int c = tid; c < j * NB; c += BT) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll 4
            for (int i = 0; i < NB; i += 4) {
                a0 = fma(Lrow[(int64_t)i * ld + c], xs[i], a0);
                a1 = fma(Lrow[(int64_t)(i + 1) * ld + c], xs[i + 1], a1);
                a2 = fma(Lrow[(int64_t)(i + 2) * ld + c], xs[i + 2], a2);
                a3 = fma(Lrow[(int64_t)(i + 3) * ld + c], xs[i + 3], a3);
            }
            s[c] -= (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
    }
}

inline int64_t tiles_of(int64_t N) { return (N + NB - 1) / NB; }

}  // namespace

extern "C" size_t als_spd_solve_workspace_bytes(int64_t N) {
    if (N < 1 || N > ALS_SPD_MAX_N) return 0;
    const int64_t T = tiles_of(N), NP = T * NB;
    return (size_t)((T + 1) * NB * NP + T * NB * NB) * sizeof(double) + 256;
}

extern "C" int als_spd_solve_f64(int64_t N, const double* A, int64_t lda, const double* b, double diag_add,
                                 double* x, void* workspace, int32_t* status, void* stream) {
    if (N < 1 || N > ALS_SPD_MAX_N || !A || !b || !x || !workspace || !status || lda < N) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    const int T = (int)tiles_of(N);
    const int64_t NP = (int64_t)T * NB;
    double* M = (double*)((char*)workspace + 256);
    const int64_t total = (int64_t)(T + 1) * NB * NP;
    double* Ld = M + total;
    const int pack_grid = (int)((total + 256 * 8 - 1) / (256 * 8) < 2048 ? (total + 256 * 8 - 1) / (256 * 8) : 2048);
    hipLaunchKernelGGL(k_spd_pack, dim3(pack_grid), dim3(256), 0, st, N, NP, T, A, lda, b, diag_add, M, status);
    for (int j = 0; j < T; ++j) {
        hipLaunchKernelGGL(k_spd_panel, dim3((unsigned)(T - j)), dim3(256), 0, st, NP, M, Ld, status, j);
        const int64_t nt = (int64_t)(T - j - 1) * (T - j + 2) / 2;      // sum_{c=j+1}^{T-1} (T - c + 1)
        if (nt > 0) hipLaunchKernelGGL(k_spd_update, dim3((unsigned)nt), dim3(256), 0, st, T, NP, M, j);
    }
    const size_t lds_back = (size_t)(NP + NB) * sizeof(double);      // <= 64 KB for N <= ALS_SPD_MAX_N
    hipLaunchKernelGGL(k_spd_backsolve, dim3(1), dim3(512), lds_back, st, T, N, NP, M, Ld, x);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
