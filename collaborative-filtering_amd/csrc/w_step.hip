// K3/K4: normal equations of the feature-projection (W) step.
//
// Replaces reference scripts/als.py:469-499 - the N_obs x (d k) design matrix and its Gram are never
// formed.  With G_i = U_i^T U_i (the item Gram the V-step of the same iteration accumulated,
// als_row_solve `gram_out`) the reference's ridge system for feature f is
//     A_f = sum_i (x_i x_i^T) (x) G_i  (+ lambda I, added by the caller)
//     b_f = sum_i x_i (x) h_{f,i},   h_{f,i} = g_i + G_i xw_{f,i},   g_i = U_i^T rho_i - G_i z_i
// with rho the residual without any U.Z term, z_i = V_i + sum_f xw_{f,i}, xw_{f,i} = W_f^T x_{f,i}
// (old W: the reference's Jacobi-across-features quirk, scripts/als.py:474-489).  U_i^T rho_i follows
// from the V-step by-products: rhs_i - (b_i_new - b_i_old) colsum_i.  Everything is kept in perm space
// (als_device.hpp); accumulation is fp64.
//
//   k_w_item_vectors   one wave per item: h_{f,i} for every feature (perm space, fp32); G_i z and all G_i xw_f as one
//                      product on the f32 matrix cores (gsym_matmul16)
//   k_w_accumulate     one workgroup per (feature-column pair a <= a', item chunk): partial lower blocks of
//                      sum_i x_ia x_ia' G_i and (a == a') partial b rows; contributing items compacted per tile
//   k_w_reduce         fixed-order sum over the chunks, written into the compact (d*k)^2 system (storage order)
#include "als_device.hpp"
#include "als_hip.h"

namespace {

__device__ __forceinline__ float gsym(const float* __restrict__ G, int ld, int r, int c) {
    // G holds the lower 16x16 blocks (block row >= block col) of a symmetric matrix.  The two halves of a
    // diagonal block come from different accumulation orders and are equal only to rounding: always read
    // the lower-triangle element, so that gsym(r, c) == gsym(c, r) bitwise.
    return (r >= c) ? G[r * ld + c] : G[c * ld + r];
}

// Y = Gsym X for up to 16 vectors at once on the f32 matrix cores, straight from the lower blocks in global
// memory - no LDS image of the Gram (66 KB at k = 128: two items per CU), so the kernels that use it run at the
// occupancy their registers allow.  xs: LDS, [16][KP] floats, vector v in row v (unused rows zero); ys: LDS,
// [16][KP], receives Y.  Per lower block (I, K): one b128 load in row layout (lane (c, q): G[16I + c][16K + 4q ..
// 4q+3], the A operand of Y_I += G_IK X_K, contraction index 4q + e in step e) and four dword loads in column
// layout (G[16I + 4q + e][16K + c], the A operand of Y_K += G_IK^T X_I); a diagonal block takes its lower triangle
// from either layout (the two halves come from different accumulation orders: Gsym(r, c) = G[max][min], as gsym).
// 8 (4 on the diagonal) v_mfma_f32_16x16x4_f32 per block: 256 per item at k = 128.
template <int KB>
__device__ __forceinline__ void gsym_matmul16(const float* __restrict__ G, const float* __restrict__ xs,
                                              float* __restrict__ ys, int lane) {
    constexpr int KP = KCfg<KB>::KP;
    const int c = lane & 15, q = lane >> 4;
    f32x4 xb[KB], y[KB];
#pragma unroll
    for (int K = 0; K < KB; ++K) {
        xb[K] = *reinterpret_cast<const f32x4*>(xs + c * KP + 16 * K + 4 * q);      // X[16K + 4q + e][vector c]
        y[K] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int I = 0; I < KB; ++I) {
        // all loads of block row I first (5 (I + 1) in flight), then its MFMAs
        f32x4 ga[KB], gc[KB];
#pragma unroll
        for (int K = 0; K <= I; ++K) {
            const float* B = G + (16 * I) * KP + 16 * K;
            ga[K] = *reinterpret_cast<const f32x4*>(B + c * KP + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) gc[K][e] = B[(4 * q + e) * KP + c];
        }
#pragma unroll
        for (int K = 0; K <= I; ++K) {
            if (I == K) {
                f32x4 a;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (4 * q + e <= c) ? ga[K][e] : gc[K][e];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], xb[I][e], y[I], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) y[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[K][e], xb[K][e], y[I], 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[K] = __builtin_amdgcn_mfma_f32_16x16x4f32(gc[K][e], xb[I][e], y[K], 0, 0, 0);
            }
        }
    }
    // C/D layout: lane (c, q), register r = Y[16I + 4q + r][vector c]
#pragma unroll
    for (int I = 0; I < KB; ++I) *reinterpret_cast<f32x4*>(ys + c * KP + 16 * I + 4 * q) = y[I];
    wave_lds_sync();
}

template <int KB>
__global__ __launch_bounds__(64)
void k_w_item_vectors(int k, int64_t i0, int64_t i1, const float* __restrict__ gram,
                      const float* __restrict__ rhs, const float* __restrict__ colsum,
                      const float* __restrict__ V, const float* __restrict__ b_new,
                      const float* __restrict__ b_old, int nfeat, int D, const float* __restrict__ X,
                      const int* __restrict__ feat_off, const float* __restrict__ W,
                      float* __restrict__ H, int64_t nrows_h) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR;
    const int64_t i = i0 + blockIdx.x;
    if (i >= i1) return;
    const int lane = threadIdx.x;
    const float* G = gram + i * KP * KP;
    // vectors z, xw_0 ... xw_{nfeat-1} (rows 0 ... nfeat of xs, the rest zero) -> G z, G xw_f in ys, all at once
    __shared__ __attribute__((aligned(16))) float xs[16 * KP], ys[16 * KP];
    int p[NR], col[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        p[rr] = min(lane + 64 * rr, KP - 1);
        col[rr] = perm_to_col<KB>(p[rr]);
        const bool mine = lane + 64 * rr < KP;
        float s = V[i * KP + col[rr]];
        for (int a = 0; a < D; ++a) s = fmaf(X[i * D + a], W[(int64_t)a * KP + col[rr]], s);    // z = V + sum_f xw_f
        if (mine) xs[p[rr]] = s;
        for (int f = 0; f < 15; ++f) {
            float t = 0.f;
            if (f < nfeat)
                for (int a = feat_off[f]; a < feat_off[f + 1]; ++a) t = fmaf(X[i * D + a], W[(int64_t)a * KP + col[rr]], t);
            if (mine) xs[(f + 1) * KP + p[rr]] = t;
        }
    }
    wave_lds_sync();
    gsym_matmul16<KB>(G, xs, ys, lane);
    const float db = b_new[i] - b_old[i];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (lane + 64 * rr < KP) {
            const float g = rhs[i * KP + p[rr]] - db * colsum[i * KP + p[rr]] - ys[p[rr]];
            for (int f = 0; f < nfeat; ++f)
                H[((int64_t)f * nrows_h + i) * KP + p[rr]] = g + ys[(f + 1) * KP + p[rr]];
        }
    }
}

// The same per-item vectors from FLOAT64 by-products (als_w_params::f64: the V-step ran in fp64 and left its Gram,
// right-hand side and column sums as doubles) - the accuracy path for feature projections with lambda_w ~ 0, where
// the (d k)^2 system is singular up to the 1e-10 the reference adds and only an fp64-consistent assembly keeps its
// null space clean (scripts/als.py:497-500).  Plain fp64 FMAs, one wave per item, every lane its own rows of G.
__device__ __forceinline__ double gsym64(const double* __restrict__ G, int ld, int r, int c) {
    return (r >= c) ? G[r * ld + c] : G[c * ld + r];
}

template <int KB, int NV>
__device__ __forceinline__ void gsym_matvecs_f64(const double* __restrict__ G, const double* __restrict__ xs,
                                                 double (&y)[NV][KCfg<KB>::NR], int nv, int lane) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) y[v][rr] = 0.0;
    for (int c = 0; c < KP; ++c) {
        double g[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) g[rr] = gsym64(G, KP, min(lane + 64 * rr, KP - 1), c);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            if (v < nv) {
                const double x = xs[v * KP + c];
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) y[v][rr] = fma(g[rr], x, y[v][rr]);
            }
        }
    }
}

template <int KB>
__global__ __launch_bounds__(64)
void k_w_item_vectors_f64(int k, int64_t i0, int64_t i1, const double* __restrict__ gram,
                          const double* __restrict__ rhs, const double* __restrict__ colsum,
                          const float* __restrict__ V, const float* __restrict__ b_new,
                          const float* __restrict__ b_old, int nfeat, int D, const float* __restrict__ X,
                          const int* __restrict__ feat_off, const double* __restrict__ W64,
                          double* __restrict__ H, int64_t nrows_h) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, NV = 9;
    const int64_t i = i0 + blockIdx.x;
    if (i >= i1) return;
    const int lane = threadIdx.x;
    const double* G = gram + i * KP * KP;
    __shared__ double xs[NV * KP];
    int p[NR], col[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        p[rr] = min(lane + 64 * rr, KP - 1);
        col[rr] = perm_to_col<KB>(p[rr]);
        const bool mine = lane + 64 * rr < KP;
        double zsum = (double)V[i * KP + col[rr]];
        for (int f = 0; f < NV - 1; ++f) {
            double t = 0.0;
            if (f < nfeat && col[rr] < k)
                for (int a = feat_off[f]; a < feat_off[f + 1]; ++a) t = fma((double)X[i * D + a], W64[(int64_t)a * k + col[rr]], t);
            zsum += t;
            if (mine) xs[(f + 1) * KP + p[rr]] = t;
        }
        if (mine) xs[p[rr]] = zsum;                      // z = V + sum_f xw_f
    }
    wave_lds_sync();
    double y[NV][NR];
    gsym_matvecs_f64<KB, NV>(G, xs, y, nfeat + 1, lane);
    const double db = (double)b_new[i] - (double)b_old[i];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (lane + 64 * rr < KP) {
            const double g = rhs[i * KP + p[rr]] - db * colsum[i * KP + p[rr]] - y[0][rr];
#pragma unroll
            for (int f = 0; f < NV - 1; ++f)
                if (f < nfeat) H[((int64_t)f * nrows_h + i) * KP + p[rr]] = g + y[f + 1][rr];
        }
    }
}

template <int KB>
__global__ __launch_bounds__(64)
void k_item_stats_f64(int64_t i0, int64_t i1, int ld, const double* __restrict__ gram, const double* __restrict__ rhs,
                      const double* __restrict__ colsum, const double* __restrict__ sumr, const double* __restrict__ sumr2,
                      const int64_t* __restrict__ indptr, const float* __restrict__ Z, const float* __restrict__ b_new,
                      const float* __restrict__ b_old, float* __restrict__ stat_out) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR;
    const int64_t i = i0 + blockIdx.x;
    if (i >= i1) return;
    const int lane = threadIdx.x;
    const double* G = gram + i * KP * KP;
    __shared__ double xs[KP];
    int p[NR];
    double z[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        p[rr] = min(lane + 64 * rr, KP - 1);
        z[rr] = (lane + 64 * rr < KP) ? (double)Z[i * ld + perm_to_col<KB>(p[rr])] : 0.0;
        if (lane + 64 * rr < KP) xs[p[rr]] = z[rr];
    }
    wave_lds_sync();
    double y[1][NR];
    gsym_matvecs_f64<KB, 1>(G, xs, y, 1, lane);
    double zgz = 0.0, zr = 0.0, zc = 0.0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
        if (lane + 64 * rr < KP) {
            zgz += z[rr] * y[0][rr];
            zr += z[rr] * rhs[i * KP + p[rr]];
            zc += z[rr] * colsum[i * KP + p[rr]];
        }
    zgz = wave_sum_d(zgz); zr = wave_sum_d(zr); zc = wave_sum_d(zc);
    if (lane == 0) {
        const double n = (double)(indptr[i + 1] - indptr[i]);
        const double b = b_new[i], bo = b_old[i], sr = sumr[i], sr2 = sumr2[i];
        const double s1 = sr - n * b - zc;
        const double s2 = (sr2 - 2.0 * b * sr + n * b * b) - 2.0 * (zr + (bo - b) * zc) + zgz;
        stat_out[2 * i] = (float)s1;
        stat_out[2 * i + 1] = (float)s2;
    }
}

// Closed-form residual sums of one item when Z != V (features present), one wave per item:
//   d = rho - b_new - u . z_i over the item's ratings, rho = r - mu - b_u
//   sum d   = sum rho - n b_new - z . (U_i^T 1)
//   sum d^2 = sum (rho - b_new)^2 - 2 z . U_i^T (rho - b_new) + z^T G_i z
// from the V-step by-products (G_i, U_i^T (rho - b_old), U_i^T 1, sum rho, sum rho^2).  fp64 from the dot
// products on.  Replaces the standalone pass over the ratings (k_residual_stats) in fits with features.
template <int KB>
__global__ __launch_bounds__(64)
void k_item_stats(int64_t i0, int64_t i1, int ld, const float* __restrict__ gram, const float* __restrict__ rhs,
                  const float* __restrict__ colsum, const float* __restrict__ sumr, const float* __restrict__ sumr2,
                  const int64_t* __restrict__ indptr, const float* __restrict__ Z, const float* __restrict__ b_new,
                  const float* __restrict__ b_old, float* __restrict__ stat_out) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR;
    const int64_t i = i0 + blockIdx.x;
    if (i >= i1) return;
    const int lane = threadIdx.x;
    const float* G = gram + i * KP * KP;
    __shared__ __attribute__((aligned(16))) float xs[16 * KP], ys[16 * KP];
    int p[NR];
    float z[NR], gz[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        p[rr] = min(lane + 64 * rr, KP - 1);
        z[rr] = (lane + 64 * rr < KP) ? Z[i * ld + perm_to_col<KB>(p[rr])] : 0.f;
        if (lane + 64 * rr < KP) {
            xs[p[rr]] = z[rr];
#pragma unroll
            for (int v = 1; v < 16; ++v) xs[v * KP + p[rr]] = 0.f;
        }
    }
    wave_lds_sync();
    gsym_matmul16<KB>(G, xs, ys, lane);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) gz[rr] = ys[p[rr]];
    double zgz = 0.0, zr = 0.0, zc = 0.0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
        if (lane + 64 * rr < KP) {
            zgz += (double)z[rr] * (double)gz[rr];
            zr += (double)z[rr] * (double)rhs[i * KP + p[rr]];
            zc += (double)z[rr] * (double)colsum[i * KP + p[rr]];
        }
    zgz = wave_sum_d(zgz); zr = wave_sum_d(zr); zc = wave_sum_d(zc);
    if (lane == 0) {
        const double n = (double)(indptr[i + 1] - indptr[i]);
        const double b = b_new[i], bo = b_old[i], sr = sumr[i], sr2 = sumr2[i];
        const double s1 = sr - n * b - zc;
        const double s2 = (sr2 - 2.0 * b * sr + n * b * b) - 2.0 * (zr + (bo - b) * zc) + zgz;
        stat_out[2 * i] = (float)s1;
        stat_out[2 * i + 1] = (float)s2;
    }
}

// lower 16x16 block q of the KB x KB block grid (row-major over I >= J): q -> (I, J)
__device__ __forceinline__ void lower_block(int q, int& I, int& J) {
    I = 0;
    while (q > I) { q -= I + 1; ++I; }
    J = q;
}

// grid: (npairs, nchunks); block 256.  One workgroup sums w_i G_i (w_i = x_ia x_ia') over the items of
// its chunk for one feature-column pair a <= a', lower 16x16 blocks only: thread t owns element
// (t >> 4, t & 15) of every lower block (NACC fp64 accumulators).  The chunk is walked in tiles of 256
// items: the tile's contributing items (w != 0) are compacted, in item order, into an LDS list, and the
// list is consumed four items at a time so that 4 NACC independent loads are in flight per thread (a
// serial `if (w == 0) continue` loop pays one dependent-load latency per item).  Fixed summation order.
template <int KB, typename T>
__global__ __launch_bounds__(256)
void k_w_accumulate(int64_t i0, int64_t i1, int nchunks, const T* __restrict__ gram, int D, int foff, int d,
                    const float* __restrict__ X, const T* __restrict__ Hf, double* __restrict__ partA,
                    double* __restrict__ partB) {
    constexpr int KP = KCfg<KB>::KP, NACC = KCfg<KB>::NACC;
    __shared__ int l_item[256];
    __shared__ float l_w[256], l_xa[256];
    __shared__ int l_cnt[4];
    int pair = blockIdx.x, a = 0;
    while (pair >= d - a) { pair -= d - a; ++a; }
    const int a2 = a + pair;
    const int chunk = blockIdx.y;
    const int64_t per = (i1 - i0 + nchunks - 1) / nchunks;
    const int64_t cb = i0 + chunk * per, ce = min(i1, cb + per);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int rr = t >> 4, cc = t & 15;
    int offT[NACC];                                // element offsets inside G_i (diagonal blocks: lower twin)
#pragma unroll
    for (int q = 0; q < NACC; ++q) {
        int I, J;
        lower_block(q, I, J);
        offT[q] = (I == J && rr < cc) ? (16 * I + cc) * KP + 16 * J + rr : (16 * I + rr) * KP + 16 * J + cc;
    }
    double acc[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = 0.0;
    double accb = 0.0;                             // thread t < KP: b row entry (only when a == a2)
    const bool want_b = (a == a2) && t < KP;
    for (int64_t tile = cb; tile < ce; tile += 256) {
        const int64_t i = tile + t;
        float xa = 0.f, w = 0.f;
        if (i < ce) {
            xa = X[i * D + foff + a];
            w = xa * X[i * D + foff + a2];
        }
        const unsigned long long m = __ballot(w != 0.f);
        if (lane == 0) l_cnt[wv] = __popcll(m);
        __syncthreads();
        int base = 0, total = 0;
#pragma unroll
        for (int v = 0; v < 4; ++v) { base += (v < wv) ? l_cnt[v] : 0; total += l_cnt[v]; }
        if (w != 0.f) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            l_item[pos] = (int)(i - tile);
            l_w[pos] = w;
            l_xa[pos] = xa;
        }
        __syncthreads();
        int n = 0;
        for (; n + 4 <= total; n += 4) {
            T g[4][NACC], h[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t it = tile + l_item[n + u];
                const T* G = gram + it * KP * KP;
#pragma unroll
                for (int q = 0; q < NACC; ++q) g[u][q] = G[offT[q]];
                h[u] = want_b ? Hf[it * KP + t] : (T)0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double w64 = (double)l_w[n + u];
#pragma unroll
                for (int q = 0; q < NACC; ++q) acc[q] += w64 * (double)g[u][q];
                accb += (double)l_xa[n + u] * (double)h[u];
            }
        }
        for (; n < total; ++n) {
            const int64_t it = tile + l_item[n];
            const T* G = gram + it * KP * KP;
            const double w64 = (double)l_w[n];
#pragma unroll
            for (int q = 0; q < NACC; ++q) acc[q] += w64 * (double)G[offT[q]];
            if (want_b) accb += (double)l_xa[n] * (double)Hf[it * KP + t];
        }
        __syncthreads();
    }
    double* pA = partA + ((int64_t)blockIdx.x * nchunks + chunk) * (NACC * 256);
#pragma unroll
    for (int q = 0; q < NACC; ++q) pA[q * 256 + t] = acc[q];
    if (want_b) partB[((int64_t)a * nchunks + chunk) * KP + t] = accb;
}

// Fixed-order sum over the chunks, written straight into the compact system of the feature:
// A [(d k)][(d k)], row/col index a k + storage column, B [d k]; padded perm positions are dropped.
// Block (a, a') of A is symmetric itself (G_i is), and block (a', a) is its transpose: every lower
// element lands in up to four places with the same value (bitwise symmetric A).
template <int KB>
__global__ __launch_bounds__(256)
void k_w_reduce(int k, int nchunks, int d, const double* __restrict__ partA, const double* __restrict__ partB,
                double* __restrict__ A, double* __restrict__ B) {
    // grid (npairs, NACC + 1): y < NACC sums one lower block of the pair, y == NACC the b row (a == a' only)
    constexpr int KP = KCfg<KB>::KP, NACC = KCfg<KB>::NACC;
    int pair = blockIdx.x, a = 0;
    while (pair >= d - a) { pair -= d - a; ++a; }
    const int a2 = a + pair;
    const int64_t N = (int64_t)d * k;
    const int t = threadIdx.x, rr = t >> 4, cc = t & 15;
    const int q = blockIdx.y;
    if (q < NACC) {
        const double* src = partA + (int64_t)blockIdx.x * nchunks * (NACC * 256) + q * 256 + t;
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < nchunks; ++c) s += src[(int64_t)c * (NACC * 256)];
        int I, J;
        lower_block(q, I, J);
        if (I == J && rr < cc) return;             // the canonical twin (cc, rr) of this thread's element writes it
        const int r = perm_to_col<KB>(16 * I + rr), c2 = perm_to_col<KB>(16 * J + cc);
        if (r >= k || c2 >= k) return;
        const int64_t ra = (int64_t)a * k, rb = (int64_t)a2 * k;
        A[(ra + r) * N + rb + c2] = s;
        A[(ra + c2) * N + rb + r] = s;
        A[(rb + c2) * N + ra + r] = s;
        A[(rb + r) * N + ra + c2] = s;
    } else if (a == a2) {
        for (int tt = t; tt < KP; tt += 256) {
            const int col = perm_to_col<KB>(tt);
            if (col >= k) continue;
            double s = 0.0;
#pragma unroll 8
            for (int c = 0; c < nchunks; ++c) s += partB[((int64_t)a * nchunks + c) * KP + tt];
            B[(int64_t)a * k + col] = s;
        }
    }
}

template <int KB>
int launch_w(const als_w_params* p, hipStream_t st) {
    constexpr int KP = KCfg<KB>::KP;
    const int64_t nit = p->item_end - p->item_begin;
    if (nit <= 0) return 0;
    if (p->phase == 0) {
        if (p->f64)
            hipLaunchKernelGGL(k_w_item_vectors_f64<KB>, dim3((unsigned)nit), dim3(64), 0, st, p->k, p->item_begin,
                               p->item_end, (const double*)p->gram, (const double*)p->rhs, (const double*)p->colsum,
                               p->V, p->b_new, p->b_old, p->nfeat, p->D, p->X, p->feat_off, (const double*)p->W,
                               (double*)p->H, p->nrows_h);
        else
            hipLaunchKernelGGL(k_w_item_vectors<KB>, dim3((unsigned)nit), dim3(64), 0, st, p->k, p->item_begin,
                               p->item_end, p->gram, p->rhs, p->colsum, p->V, p->b_new, p->b_old, p->nfeat, p->D,
                               p->X, p->feat_off, p->W, p->H, p->nrows_h);
    } else {
        const int d = p->feat_d, npairs = d * (d + 1) / 2;
        if (p->f64)
            hipLaunchKernelGGL((k_w_accumulate<KB, double>), dim3(npairs, p->nchunks), dim3(256), 0, st, p->item_begin,
                               p->item_end, p->nchunks, (const double*)p->gram, p->D, p->feat_col0, d, p->X,
                               (const double*)p->H + (int64_t)p->feat_index * p->nrows_h * KP, p->partA, p->partB);
        else
            hipLaunchKernelGGL((k_w_accumulate<KB, float>), dim3(npairs, p->nchunks), dim3(256), 0, st, p->item_begin,
                               p->item_end, p->nchunks, p->gram, p->D, p->feat_col0, d, p->X,
                               p->H + (int64_t)p->feat_index * p->nrows_h * KP, p->partA, p->partB);
        hipLaunchKernelGGL(k_w_reduce<KB>, dim3(npairs, KCfg<KB>::NACC + 1), dim3(256), 0, st, p->k, p->nchunks, d, p->partA, p->partB,
                           p->A_out, p->B_out);
    }
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

extern "C" int als_w_normal_equations(const als_w_params* p, void* stream) {
    if (!p) return ALS_E_BADARG;
    const int ld = als_padded_k(p->k);
    if (ld < 0) return ALS_E_BADK;
    if (p->ld != ld || !p->gram || !p->X || !p->H || p->D <= 0 || p->item_end < p->item_begin) return ALS_E_BADARG;
    if (p->phase == 0) {
        if (!p->rhs || !p->colsum || !p->V || !p->b_new || !p->b_old || !p->feat_off || !p->W ||
            p->nfeat < 1 || p->nfeat > 8)
            return ALS_E_BADARG;
    } else if (p->phase == 1) {
        if (!p->partA || !p->partB || !p->A_out || !p->B_out || p->feat_d < 1 || p->nchunks < 1 ||
            p->feat_index < 0 || p->feat_col0 < 0 || p->feat_col0 + p->feat_d > p->D)
            return ALS_E_BADARG;
    } else {
        return ALS_E_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (ld / 16) {
        case 1: return launch_w<1>(p, st);
        case 2: return launch_w<2>(p, st);
        case 3: return launch_w<3>(p, st);
        case 4: return launch_w<4>(p, st);
        case 5: return launch_w<5>(p, st);
        case 6: return launch_w<6>(p, st);
        case 7: return launch_w<7>(p, st);
        case 8: return launch_w<8>(p, st);
        case 9: return launch_w<9>(p, st);
        case 10: return launch_w<10>(p, st);
    }
    return ALS_E_BADK;
}

namespace {
template <int KB>
int launch_item_stats(int64_t i0, int64_t i1, int ld, const float* gram, const float* rhs, const float* colsum,
                      const float* sumr, const float* sumr2, const int64_t* indptr, const float* Z,
                      const float* b_new, const float* b_old, float* stat_out, hipStream_t st) {
    hipLaunchKernelGGL(k_item_stats<KB>, dim3((unsigned)(i1 - i0)), dim3(64), 0, st, i0, i1, ld, gram, rhs, colsum,
                       sumr, sumr2, indptr, Z, b_new, b_old, stat_out);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
}  // namespace

namespace {
template <int KB>
int launch_item_stats_f64(int64_t i0, int64_t i1, int ld, const double* gram, const double* rhs, const double* colsum,
                          const double* sumr, const double* sumr2, const int64_t* indptr, const float* Z,
                          const float* b_new, const float* b_old, float* stat_out, hipStream_t st) {
    hipLaunchKernelGGL(k_item_stats_f64<KB>, dim3((unsigned)(i1 - i0)), dim3(64), 0, st, i0, i1, ld, gram, rhs, colsum,
                       sumr, sumr2, indptr, Z, b_new, b_old, stat_out);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
}  // namespace

extern "C" int als_item_stats_f64(int k, int ld, int64_t item_begin, int64_t item_end, const double* gram,
                                  const double* rhs, const double* colsum, const double* sumr, const double* sumr2,
                                  const int64_t* indptr, const float* Z, const float* b_new, const float* b_old,
                                  float* stat_out, void* stream) {
    const int kp = als_padded_k(k);
    if (kp < 0) return ALS_E_BADK;
    if (ld != kp || item_end < item_begin || !gram || !rhs || !colsum || !sumr || !sumr2 || !indptr || !Z ||
        !b_new || !b_old || !stat_out)
        return ALS_E_BADARG;
    if (item_end == item_begin) return 0;
    hipStream_t st = (hipStream_t)stream;
#define ALS_IS64(KBV) case KBV: return launch_item_stats_f64<KBV>(item_begin, item_end, ld, gram, rhs, colsum, sumr, \
                                                                 sumr2, indptr, Z, b_new, b_old, stat_out, st)
    switch (ld / 16) {
        ALS_IS64(1); ALS_IS64(2); ALS_IS64(3); ALS_IS64(4); ALS_IS64(5); ALS_IS64(6); ALS_IS64(7); ALS_IS64(8); ALS_IS64(9);
        ALS_IS64(10);
    }
#undef ALS_IS64
    return ALS_E_BADK;
}

extern "C" int als_item_stats(int k, int ld, int64_t item_begin, int64_t item_end, const float* gram,
                              const float* rhs, const float* colsum, const float* sumr, const float* sumr2,
                              const int64_t* indptr, const float* Z, const float* b_new, const float* b_old,
                              float* stat_out, void* stream) {
    const int kp = als_padded_k(k);
    if (kp < 0) return ALS_E_BADK;
    if (ld != kp || item_end < item_begin || !gram || !rhs || !colsum || !sumr || !sumr2 || !indptr || !Z ||
        !b_new || !b_old || !stat_out)
        return ALS_E_BADARG;
    if (item_end == item_begin) return 0;
    hipStream_t st = (hipStream_t)stream;
#define ALS_IS(KBV) case KBV: return launch_item_stats<KBV>(item_begin, item_end, ld, gram, rhs, colsum, sumr, sumr2, \
                                                           indptr, Z, b_new, b_old, stat_out, st)
    switch (ld / 16) {
        ALS_IS(1); ALS_IS(2); ALS_IS(3); ALS_IS(4); ALS_IS(5); ALS_IS(6); ALS_IS(7); ALS_IS(8); ALS_IS(9); ALS_IS(10);
    }
#undef ALS_IS
    return ALS_E_BADK;
}
