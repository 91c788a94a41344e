// Item-item similarity graph on the device: cosine products + per-row top-k (+ the classification step of the
// max-symmetrisation).  Replaces the O(n^2 d) host build of the reference (scripts/als.py:224-240:
// S = Xn Xn^T, zero diagonal, keep the top-k of every row, S = max(S, S^T)) without the n x n matrix.
//
// k_topk_sim: one wavefront per 16 rows.  The 16 x 16 block of similarities against 16 columns is one chain of
// v_mfma_f32_16x16x4_f32 over the feature dimension (operands straight from a [d/4][n][4] copy of the normalised
// features: 256 contiguous bytes per load instruction); every lane then holds 4 similarities (rows 4q .. 4q+3,
// column c) and tests them against its rows' thresholds.  Survivors are appended - ballot + prefix popcount, no
// atomics - to a per-row buffer in LDS; a row whose buffer fills is compacted by a 256-element bitonic sort of
// (list + buffer) that leaves the current top-k and the new threshold.  Entries are ordered by
// (similarity descending, column index ascending): a TOTAL order, so the result does not depend on arrival
// order - among equal similarities at the top-k boundary the LOWEST column indices win (numpy's argpartition,
// scripts/als.py:235, keeps an implementation-defined subset there; with all-distinct similarities the graphs are
// identical).
//
// k_graph_classify: max(S, S^T) on the top-k lists.  For a directed edge (i -> j, s): if i is in j's list as well
// the pair is mutual (both rows already hold s: the two dot products are the same fp32 FMA chain); otherwise
// S^T contributes 0 there, so max(s, 0) keeps the edge - and adds its mirror (j, i, s) - only for s > 0.
// Zero similarities are not edges.  The caller sorts the surviving COO entries into CSR.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "als_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TK_MAX = ALS_TOPK_MAX;      // longest list
constexpr int TK_BUF = 128;               // survivors buffered per row between compactions
constexpr int TK_CAP = TK_MAX + TK_BUF;   // 256 keys of 8 bytes per row: 32 KB per wave

__device__ __forceinline__ unsigned enc_f32(float s) {
    const unsigned b = __float_as_uint(s + 0.0f);                 // -0.0 -> +0.0
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    const unsigned lo = __shfl_xor((unsigned)v, m, 64), hi = __shfl_xor((unsigned)(v >> 32), m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// 256 keys, 4 per lane (element i = lane + 64 v), sorted DESCENDING by a bitonic network
__device__ __forceinline__ void sort256_desc(unsigned long long (&k)[4], int lane) {
#pragma unroll
    for (int size = 2; size <= 256; size <<= 1) {
#pragma unroll
        for (int j = size >> 1; j >= 1; j >>= 1) {
            if (j >= 64) {
                const int dv = j >> 6;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    if ((v & dv) == 0) {
                        const int i = lane + 64 * v;
                        const bool desc = (i & size) == 0;
                        const unsigned long long a = k[v], b = k[v | dv];
                        const bool sw = desc ? (a < b) : (a > b);
                        k[v] = sw ? b : a;
                        k[v | dv] = sw ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = lane + 64 * v;
                    const bool desc = (i & size) == 0;
                    const bool lower = (i & j) == 0;                       // this element is the lower index of the pair
                    const unsigned long long o = shfl_xor_u64(k[v], j);
                    const bool take_max = (lower == desc);
                    k[v] = take_max ? (k[v] > o ? k[v] : o) : (k[v] < o ? k[v] : o);
                }
            }
        }
    }
}

template <int NS>
__global__ __launch_bounds__(64)
void k_topk_sim(int64_t n, int64_t n_pad, const float* __restrict__ XT, int topk, float* __restrict__ top_val,
                int32_t* __restrict__ top_idx, int32_t* __restrict__ top_cnt) {
    __shared__ unsigned long long keys[16][TK_CAP];
    const int lane = threadIdx.x, c = lane & 15, q = lane >> 4;
    const int64_t rowbase = (int64_t)blockIdx.x * 16;
    for (int e = lane; e < 16 * TK_CAP; e += 64) (&keys[0][0])[e] = 0ull;
    float a[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a[s] = XT[((int64_t)s * n_pad + rowbase + c) * 4 + q];
    unsigned long long thr[4] = {0ull, 0ull, 0ull, 0ull};     // key of the current top-k-th entry of rows 4q + e (0: list not full)
    int cnt[4] = {0, 0, 0, 0};                                 // buffered survivors (same value in the 16 lanes of a q group)
    int nlist[4] = {0, 0, 0, 0};                               // valid entries of the sorted list
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();

    auto compact = [&](int r) {        // wave-uniform r: list + buffer of row r -> sorted list, new threshold
        const int rq = r >> 2, re = r & 3;
        int rc = 0, rn = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) { rc = (re == e) ? cnt[e] : rc; rn = (re == e) ? nlist[e] : rn; }
        rc = __builtin_amdgcn_readlane(rc, 16 * rq);
        rn = __builtin_amdgcn_readlane(rn, 16 * rq);
        unsigned long long k[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = lane + 64 * v;
            const bool live = (i < rn) || (i >= TK_MAX && i < TK_MAX + rc);
            k[v] = live ? keys[r][i] : 0ull;
        }
        sort256_desc(k, lane);
#pragma unroll
        for (int v = 0; v < 2; ++v) keys[r][lane + 64 * v] = k[v];          // TK_MAX = 128 = first two elements per lane
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int nn = min(topk, rn + rc);
        const unsigned long long t = (nn == topk) ? keys[r][topk - 1] : 0ull;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (q == rq && e == re) { thr[e] = t; cnt[e] = 0; nlist[e] = nn; }
    };

    const int64_t nblk = (n + 15) / 16;
    for (int64_t cb = 0; cb < nblk; ++cb) {
        const int64_t col = cb * 16 + c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float b = XT[((int64_t)s * n_pad + col) * 4 + q];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b, acc, 0, 0, 0);
        }
        bool full = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t row = rowbase + 4 * q + e;
            // the zeroed diagonal takes part in the selection like any other entry (the reference selects over the
            // whole row after np.fill_diagonal(S, 0): scripts/als.py:229-236); zeros are never edges
            const float sim = (col == row) ? 0.f : acc[e];
            const unsigned long long key = ((unsigned long long)enc_f32(sim) << 32) | (0xFFFFFFFFu - (unsigned)col);
            const bool pass = col < n && row < n && key > thr[e];
            const unsigned long long m = __ballot(pass);
            const unsigned sub = (unsigned)(m >> (16 * q)) & 0xFFFFu;
            if (pass) keys[4 * q + e][TK_MAX + cnt[e] + __popc(sub & ((1u << c) - 1u))] = key;
            cnt[e] += __popc(sub);
            full = full || cnt[e] > TK_BUF - 16;
        }
        unsigned long long need = __ballot(full);
        if (need) {                     // some row's buffer could overflow in the next block
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            __builtin_amdgcn_wave_barrier();
            for (int r = 0; r < 16; ++r) {
                int rc = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) rc = ((r & 3) == e) ? cnt[e] : rc;
                rc = __builtin_amdgcn_readlane(rc, 16 * (r >> 2));
                if (rc > TK_BUF - 16) compact(r);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int r = 0; r < 16; ++r) {
        int rc = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) rc = ((r & 3) == e) ? cnt[e] : rc;
        rc = __builtin_amdgcn_readlane(rc, 16 * (r >> 2));
        if (rc > 0) compact(r);
        int rn = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) rn = ((r & 3) == e) ? nlist[e] : rn;
        rn = __builtin_amdgcn_readlane(rn, 16 * (r >> 2));
        const int64_t row = rowbase + r;
        if (row < n) {
            for (int t = lane; t < topk; t += 64) {
                const unsigned long long key = keys[r][t];
                const bool ok = t < rn;
                top_val[row * topk + t] = ok ? dec_f32((unsigned)(key >> 32)) : 0.f;
                top_idx[row * topk + t] = ok ? (int32_t)(0xFFFFFFFFu - (unsigned)key) : -1;
            }
            if (lane == 0) top_cnt[row] = rn;
        }
    }
}

__global__ __launch_bounds__(256)
void k_graph_classify(int64_t n, int topk, const float* __restrict__ top_val, const int32_t* __restrict__ top_idx,
                      const int32_t* __restrict__ top_cnt, uint8_t* __restrict__ own, uint8_t* __restrict__ mirror) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * topk) return;
    const int64_t i = e / topk;
    const int t = (int)(e - i * topk);
    uint8_t o = 0, mr = 0;
    if (t < top_cnt[i]) {
        const float s = top_val[e];
        const int32_t j = top_idx[e];
        if (s != 0.f) {
            bool mutual = false;
            const int cj = top_cnt[j];
            for (int u = 0; u < cj; ++u) mutual = mutual || (top_idx[(int64_t)j * topk + u] == (int32_t)i);
            if (mutual) o = 1;
            else if (s > 0.f) { o = 1; mr = 1; }
        }
    }
    own[e] = o;
    mirror[e] = mr;
}

template <int NS>
int launch_topk(int64_t n, int64_t n_pad, const float* XT, int topk, float* tv, int32_t* ti, int32_t* tc, hipStream_t st) {
    hipLaunchKernelGGL(k_topk_sim<NS>, dim3((unsigned)((n + 15) / 16)), dim3(64), 0, st, n, n_pad, XT, topk, tv, ti, tc);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

extern "C" int als_topk_similarity(int64_t n, int64_t n_pad, int nsteps, const float* XT, int topk, float* top_val,
                                   int32_t* top_idx, int32_t* top_cnt, void* stream) {
    if (n < 1 || n_pad < n || (n_pad & 15) || !XT || !top_val || !top_idx || !top_cnt || topk < 1 || topk > TK_MAX ||
        n >= ((int64_t)1 << 31))
        return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    switch (nsteps) {
        case 1: return launch_topk<1>(n, n_pad, XT, topk, top_val, top_idx, top_cnt, st);
        case 2: return launch_topk<2>(n, n_pad, XT, topk, top_val, top_idx, top_cnt, st);
        case 4: return launch_topk<4>(n, n_pad, XT, topk, top_val, top_idx, top_cnt, st);
        case 5: return launch_topk<5>(n, n_pad, XT, topk, top_val, top_idx, top_cnt, st);
        case 8: return launch_topk<8>(n, n_pad, XT, topk, top_val, top_idx, top_cnt, st);
        case 16: return launch_topk<16>(n, n_pad, XT, topk, top_val, top_idx, top_cnt, st);
    }
    return ALS_E_BADARG;
}

extern "C" int als_graph_classify(int64_t n, int topk, const float* top_val, const int32_t* top_idx,
                                  const int32_t* top_cnt, uint8_t* own, uint8_t* mirror, void* stream) {
    if (n < 1 || topk < 1 || !top_val || !top_idx || !top_cnt || !own || !mirror) return ALS_E_BADARG;
    const int64_t total = n * topk;
    hipLaunchKernelGGL(k_graph_classify, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n, topk,
                       top_val, top_idx, top_cnt, own, mirror);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}
