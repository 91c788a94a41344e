// K2: one dependency level of the Gauss-Seidel Laplacian sweep.
//
// Replaces the graph part of the reference item loop (scripts/als.py:453-461,
// 464-466).  The reference reads `self.V` live, so item i sees the rows j < i
// already updated in this sweep.  Only that cheap tail is sequential: the Gram,
// the right-hand side U_i^T r_i and the Cholesky factor do not depend on V and
// were produced for all items in parallel by als_row_solve (factor-only mode).
// Items of one level ("Level schedule") do not neighbour each
// other, so one launch solves a whole level with one wavefront per item:
//   b = rhs_i + alpha * sum_j S_ij V_j ;  V_i = (L L^T)^{-1} b ;  bias update.
#include <atomic>
#include <cstdlib>
#include "als_device.hpp"
#include "als_hip.h"

extern "C" int als_gs_sweep(const als_gs_sweep_params* p, void* stream);

namespace {

template <int KB>
struct GsCfg {   // waves per workgroup: the k > 64 path keeps a [KP][KP+1] image per wave in LDS
    static constexpr int WPW = (KB <= 4) ? 4 : (KB <= 6 ? 2 : 1);
};

template <int KB>
__global__ __launch_bounds__(64 * GsCfg<KB>::WPW)
void k_gs_level(const als_gs_sweep_params P) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    constexpr int LD = KP + 1;                         // k > 64 path: L image [KP][LD] + 3 vectors
    constexpr int IMG = KP * LD + 3 * KP;
    __shared__ float lds_all[(KB <= 4) ? 1 : GsCfg<KB>::WPW * IMG];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * GsCfg<KB>::WPW + wave;
    if (tid >= P.nitems) return;
    const int item = P.items[tid];
    const int64_t i64 = item;
    const int64_t s0 = P.S_ptr[item], s1 = P.S_ptr[item + 1];
    const float nnz = (float)(P.indptr[item + 1] - P.indptr[item]);
    const float lb = P.lambda_bias_row ? P.lambda_bias_row[item] : P.lambda_bias_scalar;
    const float* M = P.factor + i64 * KP * KP;

    // graph term in perm space: lane (+64*rr) <-> perm position
    int col[NR];
    float g[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int p = min(lane + 64 * rr, KP - 1);
        col[rr] = perm_to_col<KB>(p);
        g[rr] = 0.f;
    }
    // k <= 64: the factor column of this lane (16 KB per item, independent of V) is
    // requested first so that it streams in underneath the neighbour gather
    constexpr int KPR = (KB <= 4) ? KP : 1;
    float a[KPR];
    float di = 0.f, rhs_i = 0.f, cs_i = 0.f;
    if constexpr (KB <= 4) {
        const int i = min(lane, KP - 1);
#pragma unroll
        for (int p = 0; p < KP; ++p) a[p] = M[p * KP + i];
        di = M[i * KP + i];
        rhs_i = P.rhs[i64 * KP + i];
        cs_i = P.colsum[i64 * KP + i];
    }
    // neighbours: one coalesced pass loads 64 (index, weight) pairs, lane per neighbour;
    // the V rows are then gathered 16 at a time so that 16 row loads are in flight
    for (int64_t t0 = s0; t0 < s1; t0 += 64) {
        const int nn = (int)min((int64_t)64, s1 - t0);
        const int sj_l = (lane < nn) ? P.S_idx[t0 + lane] : item;
        const float sv_l = (lane < nn) ? P.S_val[t0 + lane] : 0.f;
        for (int e0 = 0; e0 < nn; e0 += 16) {
            float vv[16][NR];
            float sv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int sj = __shfl(sj_l, e0 + e, 64);            // lanes >= nn carry (item, 0)
                sv[e] = __shfl(sv_l, e0 + e, 64);
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) vv[e][rr] = P.V[(int64_t)sj * P.ld + col[rr]];
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) g[rr] = fmaf(sv[e], vv[e][rr], g[rr]);
        }
    }

    if constexpr (KB <= 4) {
        const float rb = rhs_i + P.alpha * g[0];
        float y = 0.f;
        const float x = solve_regs<KP>(a, di, rb, lane, &y);
        float dot = 0.f, xr = 0.f, yy = 0.f, xx = 0.f;
        if (lane < KP) {
            P.V[i64 * P.ld + col[0]] = x;
            dot = cs_i * x;
            xr = rhs_i * x;
            yy = y * y;
            xx = x * x;
        }
        dot = wave_sum(dot);
        const float bnew = (P.sumr[item] - dot) / (nnz + lb + ALS_EPS);
        const float bold = P.bias[item];
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) P.bias[item] = bnew;
        if (P.stat_out) {       // closed-form residual sums of this item (see row_solve.hip)
            xr = wave_sum(xr); yy = wave_sum(yy); xx = wave_sum(xx);
            if (lane == 0) {
                const float sumr = P.sumr[item], sumr2 = P.sumr2[item];
                const double bn = bnew, dt = dot;             // a difference of large sums: combined in double
                const double s1 = (double)sumr - (double)nnz * bn;
                const double s2 = (double)sumr2 - 2.0 * bn * (double)sumr + (double)nnz * bn * bn;
                const double cross = (double)xr + ((double)bold - bn) * dt;
                const double quad = (double)yy - (double)P.lambda_eff[item] * (double)xx;
                P.stat_out[2 * i64] = (float)(s1 - dt);
                P.stat_out[2 * i64 + 1] = (float)(s2 - 2.0 * cross + quad);
            }
        }
    } else {
        float* Al = lds_all + wave * IMG;
        float* vec = Al + KP * LD;
        float* dinv = vec + 2 * KP;
        for (int p = 0; p < KP; ++p)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int i = lane + 64 * rr;
                if (i < KP) {
                    const float v = M[p * KP + i];
                    if (i > p) Al[p * LD + i] = v;          // L[i][p]
                    else if (i == p) dinv[i] = v;
                }
            }
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            if (i < KP) vec[i] = P.rhs[i64 * KP + i] + P.alpha * g[rr];
        }
        wave_lds_sync();
        solve_lds<KB, LD>(Al, dinv, vec, lane);
        float dot = 0.f;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            if (i < KP) {
                const float x = vec[i];
                P.V[i64 * P.ld + col[rr]] = x;
                dot = fmaf(P.colsum[i64 * KP + i], x, dot);
            }
        }
        dot = wave_sum(dot);
        if (lane == 0) P.bias[item] = (P.sumr[item] - dot) / (nnz + lb + ALS_EPS);
    }
}

// ---------------------------------------------------------------------------
// The level kernel on FLOAT64 by-products (als_gs_sweep_params::f64: factor, rhs, colsum, sumr, sumr2 are doubles
// written by als_row_solve with gram_mode ALS_GRAM_F64 and byproducts_f64): neighbour sum, both substitutions, bias
// and statistics in fp64 - solve_dtype="float64" end to end (the reference's type, scripts/als.py:455-466).  The
// factor streams from memory (one coalesced row per step); an accuracy path, not tuned.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <int KB>
__global__ __launch_bounds__(64)
void k_gs_level_f64(const als_gs_sweep_params P) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    const int lane = threadIdx.x;
    if ((int64_t)blockIdx.x >= P.nitems) return;
    const int item = P.items[blockIdx.x];
    const int64_t i64 = item;
    const int64_t s0 = P.S_ptr[item], s1 = P.S_ptr[item + 1];
    const double* M = (const double*)P.factor + i64 * KP * KP;
    const double* rhs = (const double*)P.rhs + i64 * KP;
    const double* colsum = (const double*)P.colsum + i64 * KP;
    int ic[NR], col[NR];
    double g[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        ic[rr] = min(lane + 64 * rr, KP - 1);
        col[rr] = perm_to_col<KB>(ic[rr]);
        g[rr] = 0.0;
    }
    for (int64_t t = s0; t < s1; ++t) {                     // S[i] @ V in index order (scripts/als.py:458)
        const int sj = P.S_idx[t];
        const double sv = (double)P.S_val[t];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) g[rr] = fma(sv, (double)P.V[(int64_t)sj * P.ld + col[rr]], g[rr]);
    }
    double di[NR], rs[NR], rhs_i[NR], y[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        di[rr] = M[ic[rr] * KP + ic[rr]];
        rhs_i[rr] = rhs[ic[rr]];
        rs[rr] = (rhs_i[rr] + (double)P.alpha * g[rr]) * di[rr];
    }
    for (int j = 0; j < KP; ++j) {                          // L y = b (pre-scaled running vector)
        double sel = rs[0];
#pragma unroll
        for (int rr = 1; rr < NR; ++rr) sel = ((j >> 6) == rr) ? rs[rr] : sel;
        const double yj = readlane_f64(sel, j & 63);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const double cf = (lane + 64 * rr > j) ? M[j * KP + ic[rr]] * di[rr] : 0.0;
            rs[rr] = fma(-cf, yj, rs[rr]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) { y[rr] = rs[rr]; rs[rr] *= di[rr]; }
    for (int i = KP - 1; i >= 0; --i) {                     // L^T x = y
        double sel = rs[0];
#pragma unroll
        for (int rr = 1; rr < NR; ++rr) sel = ((i >> 6) == rr) ? rs[rr] : sel;
        const double xi = readlane_f64(sel, i & 63);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const double cf = (lane + 64 * rr < i) ? M[i * KP + ic[rr]] * di[rr] : 0.0;
            rs[rr] = fma(-cf, xi, rs[rr]);
        }
    }
    double dot = 0.0, xr = 0.0, yy = 0.0, xx = 0.0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
        if (lane + 64 * rr < KP) {
            const double x = rs[rr];
            P.V[i64 * P.ld + col[rr]] = (float)x;
            dot = fma(colsum[ic[rr]], x, dot); xr = fma(rhs_i[rr], x, xr);
            yy = fma(y[rr], y[rr], yy); xx = fma(x, x, xx);
        }
    dot = wave_sum_d(dot);
    const double nnz = (double)(P.indptr[item + 1] - P.indptr[item]);
    const double lb = (double)(P.lambda_bias_row ? P.lambda_bias_row[item] : P.lambda_bias_scalar);
    const double sumr = ((const double*)P.sumr)[item];
    const double bnew = (sumr - dot) / (nnz + lb + 1e-10);
    const double bold = (double)P.bias[item];
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) P.bias[item] = (float)bnew;
    if (P.stat_out) {
        xr = wave_sum_d(xr); yy = wave_sum_d(yy); xx = wave_sum_d(xx);
        if (lane == 0) {
            const double sumr2 = ((const double*)P.sumr2)[item];
            const double s1v = sumr - nnz * bnew;
            const double s2v = sumr2 - 2.0 * bnew * sumr + nnz * bnew * bnew;
            const double cross = xr + (bold - bnew) * dot;
            const double quad = yy - (double)P.lambda_eff[item] * xx;
            P.stat_out[2 * i64] = (float)(s1v - dot);
            P.stat_out[2 * i64 + 1] = (float)(s2v - 2.0 * cross + quad);
        }
    }
}

// ---------------------------------------------------------------------------
// K2': the whole sweep as ONE persistent, synchronisation-free launch.
//
// Items are listed in (level, id) order and dealt round-robin to the `nwaves` waves of the launch; each wave
// walks its items in that order.  An item waits only for the neighbours it really depends on (j < i and
// swept - flagged by the sign bit of `Sw`); there is no level barrier.  Progress: the earliest unfinished
// item in the global order has all dependencies finished and is the next item of its wave - provided that
// wave is running.  The grid is sized to fit the device at once (occupancy query, per device), so every wave
// is resident unless something else holds CUs (a kernel of another stream or process); its workgroups then
// start when that something ends, the resident waves spin meanwhile.  Every spin is bounded (SPIN_LIMIT,
// ~40 ms - two orders of magnitude above a legitimate hop): on expiry the wave raises `err`, every other wave
// bails out at its next poll, and the caller falls back to the per-level launches (als_gs_sweep_levels),
// which need no co-residency.  (A ticket counter handing items to whichever wave is free needs no residency
// assumption at all, but 10^5 device-scope atomics on one address serialise at ~75 ns each: measured 7.7 ms
// per sweep at cfg 4 against 1.46 ms.)
//
// Hand-off: solved rows travel through the publication buffer `pub` (same shape as V), every word of which
// the launcher resets to GS_SENTINEL.  A producer stores its row there with agent-scope (sc1, write-
// through) stores; a consumer reads the words it needs with agent-scope (L1-bypassing) loads and polls
// until none is the sentinel.  Data and "ready" travel in the same 4-byte word (cdna_hip_programming.md,
// data-tagged granules), so no flag, no drain before a flag and no ordering between words is needed; V
// itself is written with plain stores for the launches that follow.  Every spin is bounded: on timeout the
// wave raises `err` and carries on with whatever it read (wrong numbers, no hang) - the host turns it
// into an error.
// ---------------------------------------------------------------------------
// "not yet published" marker of the dataflow sweep: a quiet NaN with a payload no arithmetic produces
constexpr unsigned GS_SENTINEL = 0x7fc0dea1u;

__device__ __forceinline__ float ld_agent(const float* p) {
    return __int_as_float(__hip_atomic_load(reinterpret_cast<const int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<int*>(p), __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Profiling builds (-DALS_GS_STAMPS, profiles/sweep_phase_stamps.py): per-phase s_memtime totals of the dataflow sweep,
// summed over all waves and items: [0] items, [1] requests (descriptor, factor column, rhs) until all have landed,
// [2] non-dependency gather, [3] dependency batches (polls + waits), [4] substitutions, [5] publication + epilogue.
#ifdef ALS_GS_STAMPS
__device__ unsigned long long g_gs_stamps[8];
#define GS_STAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[slot] += now_ - st_t; st_t = now_; } while (0)
#else
#define GS_STAMP(slot) do { } while (0)
#endif

// k <= 64: factor column in registers, 4 waves per workgroup.  k > 64, two forms:
//   image  (STREAM = false) one wave per workgroup, the item's factor staged as a [KP][KP+1] LDS image before
//          the waits - shortest hop (chain-bound sweeps), but 67 KB per wave leave 2 waves per CU;
//   stream (STREAM = true)  no LDS, the factor streams from memory during the substitutions, 4 waves per
//          workgroup, 8 per CU - 4x the items in flight, for sweeps far wider than the resident waves
//          (10^6 items at k = 128: throughput-, not chain-bound).
template <int KB, bool STREAM>
struct DfCfg {
    static constexpr bool IMAGE = KB > 4 && !STREAM;
    static constexpr int WPW = IMAGE ? 1 : 4;
    static constexpr int LD = KCfg<KB>::KP + 1;
    static constexpr int IMG = IMAGE ? KCfg<KB>::KP * LD + 3 * KCfg<KB>::KP : 1;
};

template <int KB, bool STREAM>
__global__ __launch_bounds__((64 * DfCfg<KB, STREAM>::WPW), (DfCfg<KB, STREAM>::IMAGE ? 1 : 2))
void k_gs_dataflow(const als_gs_sweep_params P, const int32_t* __restrict__ Sw, float* pub,
                   int32_t* err, int64_t nitems, int nwaves, const float* __restrict__ nondep) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    using D = DfCfg<KB, STREAM>;
    constexpr int LD = D::LD;
    constexpr bool IMAGE = D::IMAGE;
    // s_memtime ticks at the shader clock (~2.2 GHz measured, profiles/r03_placed_sweep.txt), not at 100 MHz as
    // rounds 1-2 assumed: 2^22 ticks were 1.9 ms - enough for a resident launch, but a spurious "not resident" as soon
    // as anything else holds compute units for a moment.  2^27 ticks = ~60 ms.
    constexpr unsigned long long SPIN_LIMIT = 1ull << 27;
    __shared__ float lds_img[D::IMG];
    float* Al = lds_img;
    float* vec = lds_img + (IMAGE ? KP * LD : 0);
    float* dinv = vec + (IMAGE ? 2 * KP : 0);
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * D::WPW + (threadIdx.x >> 6);
    int ic[NR], col[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        ic[rr] = min(lane + 64 * rr, KP - 1);
        col[rr] = perm_to_col<KB>(ic[rr]);
    }
#ifdef ALS_GS_STAMPS
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#endif
    for (int64_t it = gw; it < nitems; it += nwaves) {
        const int item = P.items[it];
        const int64_t i64 = item;
        const int64_t s0 = P.S_ptr[item], s1 = P.S_ptr[item + 1];
        const float nnz = (float)(P.indptr[item + 1] - P.indptr[item]);
        const float lb = P.lambda_bias_row ? P.lambda_bias_row[item] : P.lambda_bias_scalar;
        const float* M = P.factor + i64 * KP * KP;
        // k <= 64: the lane's factor column (written by an earlier launch, plain loads) is requested
        // first and streams in underneath the waits; k > 64 streams it during the solve
        constexpr int KPR = (KB <= 4) ? KP : 1;
        float a[KPR];
        float di0 = 0.f;
        if constexpr (KB <= 4) {
#pragma unroll
            for (int p = 0; p < KP; ++p) a[p] = M[p * KP + ic[0]];
            di0 = M[ic[0] * KP + ic[0]];
        } else if constexpr (IMAGE) {
            // k > 64: the whole factor (M = L + L^T, 1/diag on the diagonal: row p is also column p) goes
            // into this wave's LDS image now, underneath the waits, 8 row loads in flight
            for (int p0 = 0; p0 < KP; p0 += 8) {
                float t[8][NR];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) t[u][rr] = M[(p0 + u) * KP + ic[rr]];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr)
                        if (lane + 64 * rr < KP) {
                            Al[(p0 + u) * LD + ic[rr]] = t[u][rr];
                            if (ic[rr] == p0 + u) dinv[ic[rr]] = t[u][rr];
                        }
            }
        }
        float rhs_i[NR], cs_i[NR], g[NR];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            rhs_i[rr] = P.rhs[i64 * KP + ic[rr]];
            cs_i[rr] = P.colsum[i64 * KP + ic[rr]];
            g[rr] = 0.f;
        }
        // Pass 1 over the WHOLE neighbour list first: every neighbour that is NOT a dependency (j > i, or
        // not swept) is gathered right away, so that none of these loads is issued behind a dependency
        // wait.  Pass 2: the dependencies, always in ascending position order and GB per batch, each batch
        // as soon as all its words are there.  The order of the floating-point sum never depends on timing:
        // the sweep is bitwise reproducible.
        constexpr int GB = (NR == 1) ? 16 : 8;
#ifdef ALS_GS_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        st_acc[0] += 1;
        GS_STAMP(1);
#endif
        if (nondep) {           // pass 1 was done for all items in parallel by k_gs_nondep (same sums, same order)
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) g[rr] = nondep[i64 * P.ld + col[rr]];
        } else
        for (int64_t t0 = s0; t0 < s1; t0 += 64) {
            const int nn = (int)min((int64_t)64, s1 - t0);
            const int raw = (lane < nn) ? Sw[t0 + lane] : item;
            const int sj_l = raw & 0x7fffffff;
            const float sv_now = (lane < nn && raw >= 0) ? P.S_val[t0 + lane] : 0.f;
            for (int e0 = 0; e0 < nn; e0 += GB) {     // (dependency lanes ride along with weight 0)
                float vv[GB][NR], sv[GB];
#pragma unroll
                for (int e = 0; e < GB; ++e) {
                    const int sj = __shfl(sj_l, min(e0 + e, 63), 64);   // lanes >= nn carry (item, 0)
                    sv[e] = (e0 + e < 64) ? __shfl(sv_now, min(e0 + e, 63), 64) : 0.f;
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) vv[e][rr] = ld_agent(P.V + (int64_t)sj * P.ld + col[rr]);
                }
#pragma unroll
                for (int e = 0; e < GB; ++e)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) g[rr] = fmaf(sv[e], vv[e][rr], g[rr]);
            }
        }
        GS_STAMP(2);
        // (the next chunk's indices and weights are fetched before this chunk's dependencies are waited for)
        int raw_n = (s0 + lane < s1) ? Sw[s0 + lane] : item;
        float sv_n = (s0 + lane < s1) ? P.S_val[s0 + lane] : 0.f;
        for (int64_t t0 = s0; t0 < s1; t0 += 64) {
            const int raw = raw_n;
            const float sv_l = sv_n;
            if (t0 + 64 < s1) {
                raw_n = (t0 + 64 + lane < s1) ? Sw[t0 + 64 + lane] : item;
                sv_n = (t0 + 64 + lane < s1) ? P.S_val[t0 + 64 + lane] : 0.f;
            }
            const int sj_l = raw & 0x7fffffff;
            const bool need = raw < 0;
            constexpr int NB = 64 / GB;
            unsigned long long dep = __ballot(need);
            if (dep) {
                unsigned long long bm[NB];
                float gp[NB][NR];
#pragma unroll
                for (int bb = 0; bb < NB; ++bb) {
                    unsigned long long m = 0;
#pragma unroll
                    for (int u = 0; u < GB; ++u)
                        if (dep) { m |= dep & (~dep + 1); dep &= dep - 1; }
                    bm[bb] = m;
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) gp[bb][rr] = 0.f;
                }
                bool bail = false;
#pragma unroll
                for (int round = 0; round < 2; ++round) {
#pragma unroll
                    for (int bb = 0; bb < NB; ++bb) {
                        if (bm[bb] == 0) continue;
                        // The rows of this batch come from the publication buffer, whose every word is the
                        // sentinel until its producer has stored it: data and "ready" travel in the same
                        // 4-byte word, one round trip, no flag, no ordering between words needed.
                        unsigned long long m = bm[bb];
                        float w[GB], v[GB][NR];
                        int64_t off[GB];
#pragma unroll
                        for (int u = 0; u < GB; ++u) {
                            const int src = m ? (int)__builtin_ctzll(m) : 0;
                            w[u] = m ? __shfl(sv_l, src, 64) : 0.f;
                            off[u] = m ? (int64_t)__shfl(sj_l, src, 64) * P.ld : -1;
                            if (m) m &= m - 1;
                        }
                        bool rdy = true;
#pragma unroll
                        for (int u = 0; u < GB; ++u)
#pragma unroll
                            for (int rr = 0; rr < NR; ++rr) {
                                v[u][rr] = (off[u] >= 0) ? ld_agent(pub + off[u] + col[rr]) : 0.f;
                                rdy = rdy && (__float_as_uint(v[u][rr]) != GS_SENTINEL);
                            }
                        if (!bail && !__all(rdy)) {
                            if (round == 0) continue;                 // not ready yet: after the ready ones
                            const unsigned long long tstart = __builtin_amdgcn_s_memtime();
                            bail = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                            while (!bail && !__all(rdy)) {            // a timeout anywhere is sticky
                                __builtin_amdgcn_s_sleep(1);
                                rdy = true;
#pragma unroll
                                for (int u = 0; u < GB; ++u)
#pragma unroll
                                    for (int rr = 0; rr < NR; ++rr) {
                                        if (off[u] >= 0 && __float_as_uint(v[u][rr]) == GS_SENTINEL)
                                            v[u][rr] = ld_agent(pub + off[u] + col[rr]);
                                        rdy = rdy && (__float_as_uint(v[u][rr]) != GS_SENTINEL);
                                    }
                                if (__builtin_amdgcn_s_memtime() - tstart > SPIN_LIMIT ||
                                    __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                                    if (lane == 0) atomicExch(err, 1);
                                    bail = true;
                                }
                            }
                        }
                        bm[bb] = 0;
#pragma unroll
                        for (int u = 0; u < GB; ++u)
#pragma unroll
                            for (int rr = 0; rr < NR; ++rr) gp[bb][rr] = fmaf(w[u], v[u][rr], gp[bb][rr]);
                    }
                }
#pragma unroll
                for (int bb = 0; bb < NB; ++bb)
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr) g[rr] += gp[bb][rr];
            }
        }
        GS_STAMP(3);
        float x[NR], y[NR];
        if constexpr (KB <= 4) {
            x[0] = solve_regs<KP>(a, di0, rhs_i[0] + P.alpha * g[0], lane, &y[0]);
        } else if constexpr (!IMAGE) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) x[rr] = rhs_i[rr] + P.alpha * g[rr];
            solve_stream<KB>(M, x, lane, y);
        } else {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
                if (lane + 64 * rr < KP) vec[ic[rr]] = rhs_i[rr] + P.alpha * g[rr];
            wave_lds_sync();
            solve_lds<KB, LD>(Al, dinv, vec, lane, y);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) x[rr] = vec[ic[rr]];
            wave_lds_sync();                 // the image is refilled for the next item
        }
#ifdef ALS_GS_STAMPS
        if (x[0] == 1.2345e-30f) st_acc[4] += 1;      // (keeps the stamp behind the substitutions)
        GS_STAMP(4);
#endif
        // publish first (write-through stores into the publication buffer; each word is its own "ready"
        // flag); V itself is read again only by later launches.  The bias and the statistics below are
        // nobody's dependency.
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
            if (lane + 64 * rr < KP) {
                st_agent(pub + i64 * P.ld + col[rr], x[rr]);
                P.V[i64 * P.ld + col[rr]] = x[rr];
            }
        float dot = 0.f, xr = 0.f, yy = 0.f, xx = 0.f;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
            if (lane + 64 * rr < KP) {
                dot = fmaf(cs_i[rr], x[rr], dot); xr = fmaf(rhs_i[rr], x[rr], xr);
                yy = fmaf(y[rr], y[rr], yy); xx = fmaf(x[rr], x[rr], xx);
            }
        dot = wave_sum(dot);
        const float sumr = P.sumr[item];
        const float bnew = (sumr - dot) / (nnz + lb + ALS_EPS);
        const float bold = P.bias[item];
        __builtin_amdgcn_sched_barrier(0);
        if (lane == 0) P.bias[item] = bnew;
        if (P.stat_out) {
            xr = wave_sum(xr); yy = wave_sum(yy); xx = wave_sum(xx);
            if (lane == 0) {
                const float sumr2 = P.sumr2[item];
                const double bn = bnew, dt = dot;             // a difference of large sums: combined in double
                const double s1v = (double)sumr - (double)nnz * bn;
                const double s2v = (double)sumr2 - 2.0 * bn * (double)sumr + (double)nnz * bn * bn;
                const double cross = (double)xr + ((double)bold - bn) * dt;
                const double quad = (double)yy - (double)P.lambda_eff[item] * (double)xx;
                P.stat_out[2 * i64] = (float)(s1v - dt);
                P.stat_out[2 * i64 + 1] = (float)(s2v - 2.0 * cross + quad);
            }
        }
        GS_STAMP(5);
    }
#ifdef ALS_GS_STAMPS
    if (lane == 0)
        for (int q = 0; q < 6; ++q) atomicAdd(&g_gs_stamps[q], st_acc[q]);
#endif
}

// Pass 1 of the dataflow sweep for ALL listed items at once: g_i = sum over the neighbours j of i that are NOT
// dependencies (j > i, or not swept in this call: their rows do not change during the sweep) of S_ij V_j, with
// exactly the chunking and summation order of the in-sweep loop (64 neighbours per chunk, GB rows in flight,
// dependency lanes riding along with weight 0) - bitwise the same sums, off the sweep's dependency chain.
template <int KB>
__global__ __launch_bounds__(256)
void k_gs_nondep(const als_gs_sweep_params P, const int32_t* __restrict__ Sw, int64_t nitems, float* __restrict__ out) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    const int lane = threadIdx.x & 63;
    const int64_t it = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= nitems) return;
    const int item = P.items[it];
    const int64_t i64 = item;
    const int64_t s0 = P.S_ptr[item], s1 = P.S_ptr[item + 1];
    int col[NR];
    float g[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) { col[rr] = perm_to_col<KB>(min(lane + 64 * rr, KP - 1)); g[rr] = 0.f; }
    constexpr int GB = (NR == 1) ? 16 : 8;
    for (int64_t t0 = s0; t0 < s1; t0 += 64) {
        const int nn = (int)min((int64_t)64, s1 - t0);
        const int raw = (lane < nn) ? Sw[t0 + lane] : item;
        const int sj_l = raw & 0x7fffffff;
        const float sv_now = (lane < nn && raw >= 0) ? P.S_val[t0 + lane] : 0.f;
        for (int e0 = 0; e0 < nn; e0 += GB) {
            float vv[GB][NR], sv[GB];
#pragma unroll
            for (int e = 0; e < GB; ++e) {
                const int sj = __shfl(sj_l, min(e0 + e, 63), 64);
                sv[e] = (e0 + e < 64) ? __shfl(sv_now, min(e0 + e, 63), 64) : 0.f;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) vv[e][rr] = P.V[(int64_t)sj * P.ld + col[rr]];
            }
#pragma unroll
            for (int e = 0; e < GB; ++e)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) g[rr] = fmaf(sv[e], vv[e][rr], g[rr]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
        if (lane + 64 * rr < KP) out[i64 * P.ld + col[rr]] = g[rr];
}

template <int KB, bool STREAM>
int launch_gs_dataflow_as(const als_gs_sweep_params* p, const int32_t* Sw, float* pub,
                          int32_t* err, int64_t nitems, hipStream_t st, int* waves_out, const float* nondep) {
    // Every wave of the launch must be co-resident (a non-resident workgroup would never start while the
    // resident ones wait for its items): the grid is sized from the occupancy the runtime reports for this
    // kernel - registers and LDS included - times the number of CUs, capped at 8 waves per CU.
    constexpr int WPW = DfCfg<KB, STREAM>::WPW;
    constexpr int MAXDEV = 64;
    static std::atomic<int> per_cu_of[MAXDEV], ncu_of[MAXDEV];      // per device; zero-initialised; benign races
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) return ALS_E_LAUNCH;
    int per_cu = per_cu_of[dev].load(std::memory_order_relaxed), ncu = ncu_of[dev].load(std::memory_order_relaxed);
    if (per_cu == 0 || ncu == 0) {
        int nb = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_gs_dataflow<KB, STREAM>, 64 * WPW, 0) != hipSuccess ||
            nb < 1)
            return ALS_E_LAUNCH;
        ncu = prop.multiProcessorCount;
        per_cu = nb * WPW > 8 ? 8 / WPW : nb;
        if (per_cu < 1) per_cu = 1;
        ncu_of[dev].store(ncu, std::memory_order_relaxed);
        per_cu_of[dev].store(per_cu, std::memory_order_relaxed);
    }
    if (waves_out) { *waves_out = ncu * per_cu * WPW; return 0; }
    int nwg = ncu * per_cu;
    if ((int64_t)nwg * WPW > nitems) nwg = (int)((nitems + WPW - 1) / WPW);
    if (nwg < 1) return 0;
    hipLaunchKernelGGL((k_gs_dataflow<KB, STREAM>), dim3(nwg), dim3(64 * WPW), 0, st, *p, Sw, pub, err, nitems,
                       nwg * WPW, nondep);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

template <int KB>
int launch_gs_dataflow(const als_gs_sweep_params* p, const int32_t* Sw, float* pub,
                       int32_t* err, int64_t nitems, hipStream_t st, float* nondep) {
    if (nondep && nitems > 0)
        hipLaunchKernelGGL(k_gs_nondep<KB>, dim3((unsigned)((nitems + 3) / 4)), dim3(256), 0, st, *p, Sw, nitems, nondep);
    if constexpr (KB > 4) {
        // far more items than the image form keeps in flight: the sweep is throughput-bound, stream the factor
        int image_waves = 0;
        const int rc = launch_gs_dataflow_as<KB, false>(p, Sw, pub, err, nitems, st, &image_waves, nondep);
        if (rc != 0) return rc;
        static const char* force = getenv("ALS_GS_FORM");           // "stream" / "image": tests and experiments
        const bool stream = force ? (force[0] == 's') : nitems > (int64_t)256 * image_waves;
        if (stream) return launch_gs_dataflow_as<KB, true>(p, Sw, pub, err, nitems, st, nullptr, nondep);
    }
    return launch_gs_dataflow_as<KB, false>(p, Sw, pub, err, nitems, st, nullptr, nondep);
}

template <int KB>
int launch_gs(const als_gs_sweep_params* p, hipStream_t st) {
    constexpr int WPW = GsCfg<KB>::WPW;
    if (p->nitems <= 0) return 0;
    if (p->f64) {
        hipLaunchKernelGGL(k_gs_level_f64<KB>, dim3((unsigned)p->nitems), dim3(64), 0, st, *p);
        return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
    }
    const unsigned grid = (unsigned)((p->nitems + WPW - 1) / WPW);
    hipLaunchKernelGGL(k_gs_level<KB>, dim3(grid), dim3(64 * WPW), 0, st, *p);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

namespace {
// publish[0 .. nwords) = word (nwords is a multiple of 16: rows x padded k); 16 bytes per thread and step
__global__ __launch_bounds__(256)
void k_fill_words(uint32_t* __restrict__ dst, int64_t nwords, uint32_t word) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = {word, word, word, word};
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nwords / 4; e += (int64_t)gridDim.x * 256)
        reinterpret_cast<u32x4*>(dst)[e] = v;
}
}  // namespace

extern "C" int als_gs_sweep_dataflow(const als_gs_sweep_params* p, const int32_t* S_idx_wait, float* publish,
                                     int64_t nrows, float* nondep, int32_t* err, void* stream) {
    if (!p || !S_idx_wait || !publish || !err || nrows < 0 || p->f64) return ALS_E_BADARG;   // (fp64 by-products: level launches)
    const int ld = als_padded_k(p->k);
    if (ld < 0) return ALS_E_BADK;
    if (p->ld != ld || p->nitems < 0 || !p->S_ptr || !p->S_val || !p->factor || !p->rhs || !p->colsum ||
        !p->sumr || !p->indptr || !p->V || !p->bias || (p->nitems > 0 && !p->items))
        return ALS_E_BADARG;
    if (p->stat_out && (!p->sumr2 || !p->lambda_eff)) return ALS_E_BADARG;
    hipStream_t st = (hipStream_t)stream;
    // (a fill kernel, not hipMemsetD32Async: with TWO captured iteration graphs alive - W-step / no-W-step variant -
    // that each hold a memset node, replays went wrong from the first switch back on ROCm 7.2; with the reset as a
    // kernel node they are bitwise the eager fit, profiles/graph_early_stop_stress.py)
    if (p->nitems > 0 && nrows > 0) {
        const int64_t nwords = nrows * ld;
        const unsigned grid = (unsigned)((nwords / 4 + 255) / 256 < 8192 ? (nwords / 4 + 255) / 256 : 8192);
        hipLaunchKernelGGL(k_fill_words, dim3(grid > 0 ? grid : 1), dim3(256), 0, st, (uint32_t*)publish, nwords,
                           (uint32_t)GS_SENTINEL);
    }
    switch (ld / 16) {
        case 1: return launch_gs_dataflow<1>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 2: return launch_gs_dataflow<2>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 3: return launch_gs_dataflow<3>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 4: return launch_gs_dataflow<4>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 5: return launch_gs_dataflow<5>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 6: return launch_gs_dataflow<6>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 7: return launch_gs_dataflow<7>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 8: return launch_gs_dataflow<8>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 9: return launch_gs_dataflow<9>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
        case 10: return launch_gs_dataflow<10>(p, S_idx_wait, publish, err, p->nitems, st, nondep);
    }
    return ALS_E_BADK;
}

#ifdef ALS_GS_STAMPS
extern "C" int als_debug_gs_stamps(unsigned long long* out) {          // profiling builds only: read and reset
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gs_stamps), sizeof(z)) != hipSuccess) return ALS_E_LAUNCH;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gs_stamps), z, sizeof(z)) == hipSuccess ? 0 : ALS_E_LAUNCH;
}
#endif

extern "C" int als_gs_sweep_levels(const als_gs_sweep_params* p, const int64_t* level_offsets,
                                   int64_t nlevels, void* stream) {
    if (!p || nlevels < 0 || (nlevels > 0 && (!level_offsets || !p->items))) return ALS_E_BADARG;
    for (int64_t l = 0; l < nlevels; ++l) {
        als_gs_sweep_params q = *p;
        q.items = p->items + level_offsets[l];
        q.nitems = level_offsets[l + 1] - level_offsets[l];
        if (q.nitems < 0) return ALS_E_BADARG;
        const int rc = als_gs_sweep(&q, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

extern "C" int als_gs_sweep(const als_gs_sweep_params* p, void* stream) {
    if (!p) return ALS_E_BADARG;
    const int ld = als_padded_k(p->k);
    if (ld < 0) return ALS_E_BADK;
    if (p->ld != ld || p->nitems < 0 || !p->S_ptr || !p->factor || !p->rhs || !p->colsum ||
        !p->sumr || !p->indptr || !p->V || !p->bias)
        return ALS_E_BADARG;
    if (p->nitems > 0 && !p->items) return ALS_E_BADARG;
    if (p->stat_out && (!p->sumr2 || !p->lambda_eff || (ld > 64 && !p->f64))) return ALS_E_BADARG;   // fused stats: k <= 64 (fp32 form)
    hipStream_t st = (hipStream_t)stream;
    switch (ld / 16) {
        case 1: return launch_gs<1>(p, st);
        case 2: return launch_gs<2>(p, st);
        case 3: return launch_gs<3>(p, st);
        case 4: return launch_gs<4>(p, st);
        case 5: return launch_gs<5>(p, st);
        case 6: return launch_gs<6>(p, st);
        case 7: return launch_gs<7>(p, st);
        case 8: return launch_gs<8>(p, st);
        case 9: return launch_gs<9>(p, st);
        case 10: return launch_gs<10>(p, st);
    }
    return ALS_E_BADK;
}
