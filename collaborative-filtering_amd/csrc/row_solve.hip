// K1: per-row normal-equation build (MFMA Gram) + Cholesky solve + bias update.
//
// Replaces the bodies of the user loop (reference scripts/als.py:414-433), the
// item loop (scripts/als.py:436-466) and cholesky_solve (scripts/helpers.py:5-20).
//
// One wavefront owns one task (a row, or a <=4096-rating segment of a long
// row).  The gathered factor rows never touch LDS on their way to the matrix
// cores: lane (c,q) loads the KB contiguous floats F[idx][KB*c..] of rating
// 4*s+q, which in "perm space" (als_device.hpp) are position c of every
// 16-column block, i.e. exactly the A/B operands of v_mfma_f32_16x16x4_f32.
// The k x k Gram is accumulated in KB(KB+1)/2 upper 16x16 blocks (symmetry),
// dumped once to LDS and factorised there by the same wave.
#include "als_device.hpp"
#include "als_hip.h"

namespace {

template <int KB>
struct RowAcc {
    f32x4 acc[KCfg<KB>::NACC];
    float rhs[KB];
    float cs[KB];
    float sumr;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < KCfg<KB>::NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < KB; ++b) { rhs[b] = 0.f; cs[b] = 0.f; }
        sumr = 0.f;
    }
};

// ---------------------------------------------------------------------------
// 64 ratings: lane t holds (idx_l, r_l) of rating t; 16 steps of 4 ratings.
// ---------------------------------------------------------------------------
template <int KB, bool TAIL>
__device__ __forceinline__ void process_chunk(RowAcc<KB>& A, int idx_l, float r_l, int nvalid,
                                              const float* __restrict__ F, int ld, int c, int q) {
    constexpr int GS = KCfg<KB>::GS;
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += GS) {
        if (TAIL && 4 * g0 >= nvalid) break;
        float f[GS][KB];
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            if (!TAIL || 4 * (g0 + s) < nvalid) {
                const int src = 4 * (g0 + s) + q;
                const int idx_t = bperm_i(idx_l, src);       // invalid ratings carry idx 0
                load_frow<KB>(F + (size_t)idx_t * ld + KB * c, f[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            if (!TAIL || 4 * (g0 + s) < nvalid) {
                const int src = 4 * (g0 + s) + q;
                const float r_t = bperm_f(r_l, src);          // 0 for invalid ratings
                if (TAIL) {
                    const bool ok = src < nvalid;
#pragma unroll
                    for (int b = 0; b < KB; ++b) f[s][b] = ok ? f[s][b] : 0.f;
                }
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    A.rhs[b] = fmaf(f[s][b], r_t, A.rhs[b]);
                    A.cs[b] += f[s][b];
                }
                int a = 0;
#pragma unroll
                for (int bi = 0; bi < KB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < KB; ++bj) {
                        A.acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(f[s][bi], f[s][bj], A.acc[a], 0, 0, 0);
                        ++a;
                    }
            }
        }
    }
}

template <int KB>
__device__ __forceinline__ void gram_accumulate(RowAcc<KB>& A, const int32_t* __restrict__ idxp,
                                                const float* __restrict__ valp, int len,
                                                const float* __restrict__ F, int ld,
                                                const float* __restrict__ bias_other, float mu,
                                                float bself, int lane) {
    const int c = lane & 15, q = lane >> 4;
    // software pipeline over 64-rating chunks: indices two chunks ahead,
    // value + opposite-side bias one chunk ahead of the factor-row gathers
    int idx1 = (lane < len) ? idxp[lane] : 0;
    int idx2 = (64 + lane < len) ? idxp[64 + lane] : 0;
    float val1 = (lane < len) ? valp[lane] : 0.f;
    float bo1 = (lane < len) ? bias_other[idx1] : 0.f;
    for (int base = 0; base < len; base += 64) {
        const int idx0 = idx1;
        const float val0 = val1, bo0 = bo1;
        idx1 = idx2;
        const int t2 = base + 128 + lane;
        idx2 = (t2 < len) ? idxp[t2] : 0;
        const int t1 = base + 64 + lane;
        val1 = (t1 < len) ? valp[t1] : 0.f;
        bo1 = (t1 < len) ? bias_other[idx1] : 0.f;
        const int nvalid = min(64, len - base);
        const bool ok = lane < nvalid;
        const float rb = val0 - mu - bo0;
        A.sumr += ok ? rb : 0.f;
        const float r0 = ok ? (rb - bself) : 0.f;
        if (nvalid == 64) process_chunk<KB, false>(A, idx0, r0, 64, F, ld, c, q);
        else              process_chunk<KB, true>(A, idx0, r0, nvalid, F, ld, c, q);
    }
}

// ---------------------------------------------------------------------------
// partial normal equations <-> workspace (per-lane raw values, fixed order)
// ---------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void store_partial(const RowAcc<KB>& A, float* __restrict__ ws, int lane) {
    int it = 0;
#pragma unroll
    for (int a = 0; a < KCfg<KB>::NACC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) ws[(it++) * 64 + lane] = A.acc[a][r];
#pragma unroll
    for (int b = 0; b < KB; ++b) ws[(it++) * 64 + lane] = A.rhs[b];
#pragma unroll
    for (int b = 0; b < KB; ++b) ws[(it++) * 64 + lane] = A.cs[b];
    ws[it * 64 + lane] = A.sumr;
}

template <int KB>
__device__ __forceinline__ void add_partial(RowAcc<KB>& A, const float* __restrict__ ws, int lane) {
    int it = 0;
#pragma unroll
    for (int a = 0; a < KCfg<KB>::NACC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) A.acc[a][r] += ws[(it++) * 64 + lane];
#pragma unroll
    for (int b = 0; b < KB; ++b) A.rhs[b] += ws[(it++) * 64 + lane];
#pragma unroll
    for (int b = 0; b < KB; ++b) A.cs[b] += ws[(it++) * 64 + lane];
    A.sumr += ws[it * 64 + lane];
}

// ---------------------------------------------------------------------------
// in-LDS / in-register Cholesky.  On exit (both variants):
//   Al[j*LD + i] = L[i][j] for i > j   (row j of LDS = column j of L)
//   dinv[j]      = 1 / L[j][j]
// returns false if a pivot was not positive.
// ---------------------------------------------------------------------------
// KP <= 64: lane i keeps row i of the trailing matrix in registers (static
// register indices: the column loop is unrolled through templates).  Column j
// of L is published once to LDS (row j of Al - the layout the callers want
// anyway) and the rank-1 update reads it back as wave-uniform ds_read_b128
// broadcasts: 4 multipliers per DS instruction, no SGPR traffic.  Only the
// pivot of the next step travels by v_readlane.
template <int KB, int J>
__device__ __forceinline__ void chol_step(float* __restrict__ Al, int lane, float (&a)[KCfg<KB>::KP],
                                          float& di, bool& spd) {
    constexpr int KP = KCfg<KB>::KP, LD = KCfg<KB>::LD;
    const float d = readlane_f(a[J], J);
    spd = spd && (d > 0.f);
    float inv = __builtin_amdgcn_rsqf(d);
    inv = inv * fmaf(-0.5f * d * inv, inv, 1.5f);     // one Newton step: <= 1 ulp
    const float lij = a[J] * inv;                     // L[i][J] for lanes i >= J
    a[J] = lij;
    if (lane == J) di = inv;
    if (lane < KP) Al[J * LD + lane] = lij;           // lanes < J store the unused upper part
    if constexpr (J + 1 < KP) {
        // the next pivot column does not wait for the LDS round trip
        a[J + 1] = fmaf(-lij, readlane_f(lij, J + 1), a[J + 1]);
        constexpr int G0 = (J + 2) / 4, NG = KP / 4 - G0;
        if constexpr (NG > 0) {
            // issue every broadcast read of this step back to back, then consume: left to
            // itself the compiler keeps ~2 reads in flight and exposes the LDS latency per read
            f32x4 lc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g)
                lc[g] = *reinterpret_cast<const f32x4*>(Al + J * LD + 4 * (G0 + g));   // uniform address
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int cidx = 4 * (G0 + g) + e;
                    if (cidx > J + 1) a[cidx] = fmaf(-lij, lc[g][e], a[cidx]);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int KB, int J>
__device__ __forceinline__ void chol_steps(float* __restrict__ Al, int lane, float (&a)[KCfg<KB>::KP],
                                           float& di, bool& spd) {
    if constexpr (J < KCfg<KB>::KP) {
        chol_step<KB, J>(Al, lane, a, di, spd);
        chol_steps<KB, J + 1>(Al, lane, a, di, spd);
    }
}

// On exit a[] is lane i's column of the symmetric completion of L (what
// solve_regs wants), di = 1/L[i][i], and Al[j*LD + i] = L[i][j].
template <int KB>
__device__ __forceinline__ bool chol_regs(float* __restrict__ Al, int lane,
                                          float (&a)[KCfg<KB>::KP], float& di) {
    constexpr int KP = KCfg<KB>::KP, LD = KCfg<KB>::LD;
    static_assert(LD % 4 == 0, "uniform b128 reads need 16-byte aligned rows");
    const int i = min(lane, KP - 1);
#pragma unroll
    for (int p = 0; p < KP; ++p) a[p] = (p <= lane) ? Al[p * LD + i] : 0.f;   // A[p][i], p <= i (upper)
    wave_lds_sync();
    bool spd = true;
    di = 0.f;
    chol_steps<KB, 0>(Al, lane, a, di, spd);
    wave_lds_sync();
#pragma unroll
    for (int g4 = 0; g4 < KP / 4; ++g4) {
        const f32x4 lt = *reinterpret_cast<const f32x4*>(Al + i * LD + 4 * g4);            // L[p][i], p > i
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * g4 + e > lane) a[4 * g4 + e] = lt[e];
    }
    return spd;
}

// any KP: everything stays in LDS; lane owns matrix columns lane, lane+64, ...
template <int KB>
__device__ __forceinline__ bool chol_lds(float* __restrict__ Al, float* __restrict__ dinv, int lane) {
    constexpr int KP = KCfg<KB>::KP, LD = KCfg<KB>::LD, NR = KCfg<KB>::NR;
    bool spd = true;
    for (int j = 0; j < KP; ++j) {
        const float d = Al[j * LD + j];
        spd = spd && (d > 0.f);
        const float inv = 1.0f / sqrtf(d);
        float lj[NR];
        wave_lds_sync();
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            lj[rr] = 0.f;
            if (i > j && i < KP) { lj[rr] = Al[j * LD + i] * inv; Al[j * LD + i] = lj[rr]; }
        }
        if (lane == 0) dinv[j] = inv;
        wave_lds_sync();
        for (int cidx = j + 1; cidx < KP; ++cidx) {
            const float lc = Al[j * LD + cidx];           // broadcast L[cidx][j]
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int i = lane + 64 * rr;
                if (i >= cidx && i < KP) Al[cidx * LD + i] = fmaf(-lj[rr], lc, Al[cidx * LD + i]);
            }
        }
        wave_lds_sync();
    }
    return spd;
}

// ---------------------------------------------------------------------------
// tail: reduce, dump to LDS, regularise, factorise, solve / emit factor
// ---------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void finish_row(RowAcc<KB>& A, const als_row_solve_params& P, int row,
                                           float* __restrict__ lds, int lane) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, LD = C::LD, NR = C::NR;
    const int c = lane & 15, q = lane >> 4;
    float* Al = lds;
    float* vrhs = lds + KP * LD;
    float* vcs = vrhs + KP;
    float* dinv = vcs + KP;

    // cross-lane reductions: rhs / colsum over the four q groups, sumr over the wave
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        A.rhs[b] += __shfl_xor(A.rhs[b], 16, 64); A.rhs[b] += __shfl_xor(A.rhs[b], 32, 64);
        A.cs[b] += __shfl_xor(A.cs[b], 16, 64);   A.cs[b] += __shfl_xor(A.cs[b], 32, 64);
    }
    const float sumr = wave_sum(A.sumr);

    // accumulators (C/D layout: col = c, row = 4q + r) -> LDS, upper blocks
    {
        int a = 0;
#pragma unroll
        for (int bi = 0; bi < KB; ++bi)
#pragma unroll
            for (int bj = bi; bj < KB; ++bj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) Al[(16 * bi + 4 * q + r) * LD + 16 * bj + c] = A.acc[a][r];
                ++a;
            }
    }
    if (q == 0) {
#pragma unroll
        for (int b = 0; b < KB; ++b) { vrhs[16 * b + c] = A.rhs[b]; vcs[16 * b + c] = A.cs[b]; }
    }
    wave_lds_sync();

    const int64_t r64 = row;
    if (P.gram_out) {
        float* G = P.gram_out + r64 * KP * KP;
        for (int p = 0; p < KP; ++p)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int i = lane + 64 * rr;
                if (i < KP) G[p * KP + i] = Al[p * LD + i];
            }
    }

    // optional per-row by-products (perm space): rhs, column sums, sum of r + bias_self
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        if (i < KP) {
            if (P.rhs_out) P.rhs_out[r64 * KP + i] = vrhs[i];
            if (P.colsum_out) P.colsum_out[r64 * KP + i] = vcs[i];
        }
    }
    if (P.sumr_out && lane == 0) P.sumr_out[row] = sumr;

    // regulariser on the diagonal; padded columns get a unit pivot
    const float lam = (P.lambda_row ? P.lambda_row[row] : P.lambda_scalar) + ALS_EPS
                    + (P.diag_extra ? P.diag_extra[row] : 0.f);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        if (i < KP) Al[i * LD + i] += (perm_to_col<KB>(i) < P.k) ? lam : 1.0f;
    }
    wave_lds_sync();

    const float nnz = (float)(P.indptr[row + 1] - P.indptr[row]);
    const float lb = P.lambda_bias_row ? P.lambda_bias_row[row] : P.lambda_bias_scalar;

    if constexpr (KB <= 4) {
        // ---- k <= 64: factor and solve in registers --------------------------
        float a[KP];
        float di;
        const bool spd = chol_regs<KB>(Al, lane, a, di);
        if (!spd && lane == 0) atomicMax(P.status, row + 1);
        const int i = min(lane, KP - 1);
        if (P.factor_out) {
            float* M = P.factor_out + r64 * KP * KP;
#pragma unroll
            for (int p = 0; p < KP; ++p)
                if (lane < KP) M[p * KP + lane] = (p == lane) ? di : a[p];
            return;
        }
        const int col = perm_to_col<KB>(i);
        float rb = vrhs[i];
        if (P.rhs_extra) rb += P.rhs_extra[r64 * P.ld + col];
        const float x = solve_regs<KP>(a, di, rb, lane);
        float dot = 0.f;
        if (lane < KP) {
            P.X_out[r64 * P.ld + col] = x;
            dot = vcs[lane] * x;
        }
        dot = wave_sum(dot);
        if (lane == 0) P.bias_out[row] = (sumr - dot) / (nnz + lb + ALS_EPS);
    } else {
        // ---- k > 64: everything through LDS ----------------------------------
        const bool spd = chol_lds<KB>(Al, dinv, lane);
        if (!spd && lane == 0) atomicMax(P.status, row + 1);
        if (P.factor_out) {
            float* M = P.factor_out + r64 * KP * KP;
            for (int p = 0; p < KP; ++p)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) {
                    const int i = lane + 64 * rr;
                    if (i < KP) {
                        const int lo = min(p, i), hi = max(p, i);
                        M[p * KP + i] = (p == i) ? dinv[i] : Al[lo * LD + hi];
                    }
                }
            return;
        }
        if (P.rhs_extra) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int i = lane + 64 * rr;
                if (i < KP) vrhs[i] += P.rhs_extra[r64 * P.ld + perm_to_col<KB>(i)];
            }
            wave_lds_sync();
        }
        solve_lds<KB>(Al, dinv, vrhs, lane);
        float dot = 0.f;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            if (i < KP) {
                const float x = vrhs[i];
                P.X_out[r64 * P.ld + perm_to_col<KB>(i)] = x;
                dot = fmaf(vcs[i], x, dot);
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) P.bias_out[row] = (sumr - dot) / (nnz + lb + ALS_EPS);
    }
}

template <int KB>
__global__ __launch_bounds__(64 * KCfg<KB>::WPW, KCfg<KB>::MINW)
void k_row_tasks(const als_row_solve_params P) {
    using C = KCfg<KB>;
    __shared__ float lds_all[C::WPW * C::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * C::WPW + wave;
    if (tid >= P.ntasks) return;
    const als_task t = P.tasks[tid];
    const int row = t.row;
    const int64_t rbeg = P.indptr[row], rend = P.indptr[row + 1];
    const int64_t beg = rbeg + (int64_t)t.seg * ALS_SPLIT_CHUNK;
    const int len = (int)min((int64_t)ALS_SPLIT_CHUNK, rend - beg);
    const float mu = (float)(*P.mu);
    const float bself = P.bias_self[row];

    RowAcc<KB> A;
    A.zero();
    gram_accumulate<KB>(A, P.indices + beg, P.vals + beg, len, P.F, P.ld, P.bias_other, mu, bself, lane);
    if (t.slot >= 0) {
        store_partial<KB>(A, (float*)P.workspace + (size_t)t.slot * C::SLOT_ITEMS * 64, lane);
        return;
    }
    finish_row<KB>(A, P, row, lds_all + wave * C::LDS_FLOATS, lane);
}

template <int KB>
__global__ __launch_bounds__(64 * KCfg<KB>::WPW, KCfg<KB>::MINW)
void k_row_long(const als_row_solve_params P) {
    using C = KCfg<KB>;
    __shared__ float lds_all[C::WPW * C::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * C::WPW + wave;
    if (tid >= P.nlong) return;
    const als_long_row lr = P.long_rows[tid];
    RowAcc<KB> A;
    A.zero();
    const float* ws = (const float*)P.workspace + (size_t)lr.slot0 * C::SLOT_ITEMS * 64;
    for (int s = 0; s < lr.nslots; ++s) add_partial<KB>(A, ws + (size_t)s * C::SLOT_ITEMS * 64, lane);
    finish_row<KB>(A, P, lr.row, lds_all + wave * C::LDS_FLOATS, lane);
}

template <int KB>
int launch_row_solve(const als_row_solve_params* p, hipStream_t st) {
    using C = KCfg<KB>;
    if (p->ntasks > 0) {
        const unsigned grid = (unsigned)((p->ntasks + C::WPW - 1) / C::WPW);
        hipLaunchKernelGGL(k_row_tasks<KB>, dim3(grid), dim3(64 * C::WPW), 0, st, *p);
    }
    if (p->nlong > 0) {
        const unsigned grid = (unsigned)((p->nlong + C::WPW - 1) / C::WPW);
        hipLaunchKernelGGL(k_row_long<KB>, dim3(grid), dim3(64 * C::WPW), 0, st, *p);
    }
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

extern "C" int als_version(void) { return ALS_HIP_VERSION; }
extern "C" int als_padded_k(int k) { return (k < 1 || k > ALS_MAX_K) ? ALS_E_BADK : 16 * ((k + 15) / 16); }
extern "C" int als_perm_index(int k, int c) {
    const int ld = als_padded_k(k);
    if (ld < 0 || c < 0 || c >= ld) return ALS_E_BADARG;
    const int KB = ld / 16;
    return 16 * (c % KB) + c / KB;
}
extern "C" int64_t als_partial_slot_bytes(int k) {
    const int ld = als_padded_k(k);
    if (ld < 0) return ALS_E_BADK;
    const int KB = ld / 16;
    return (int64_t)(KB * (KB + 1) / 2 * 4 + 2 * KB + 1) * 64 * sizeof(float);
}

extern "C" int als_row_solve(const als_row_solve_params* p, void* stream) {
    if (!p) return ALS_E_BADARG;
    const int ld = als_padded_k(p->k);
    if (ld < 0) return ALS_E_BADK;
    if (p->ld != ld || !p->indptr || !p->indices || !p->vals || !p->F || !p->bias_self ||
        !p->bias_other || !p->mu || !p->status || p->ntasks < 0 || p->nlong < 0)
        return ALS_E_BADARG;
    if (p->ntasks > 0 && !p->tasks) return ALS_E_BADARG;
    if (p->nlong > 0 && (!p->long_rows || !p->workspace)) return ALS_E_BADARG;
    if (p->factor_out) {
        if (!p->rhs_out || !p->colsum_out || !p->sumr_out) return ALS_E_BADARG;
    } else if (!p->X_out || !p->bias_out) {
        return ALS_E_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (ld / 16) {
        case 1: return launch_row_solve<1>(p, st);
        case 2: return launch_row_solve<2>(p, st);
        case 3: return launch_row_solve<3>(p, st);
        case 4: return launch_row_solve<4>(p, st);
        case 5: return launch_row_solve<5>(p, st);
        case 6: return launch_row_solve<6>(p, st);
        case 7: return launch_row_solve<7>(p, st);
        case 8: return launch_row_solve<8>(p, st);
        case 9: return launch_row_solve<9>(p, st);
        case 10: return launch_row_solve<10>(p, st);
    }
    return ALS_E_BADK;
}
