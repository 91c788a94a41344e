// K1 in float64 (gram_mode == ALS_GRAM_F64): the reference's arithmetic type end to end.
//
// The reference builds and solves every normal equation in float64 (scripts/als.py:426-428, 455-461,
// scripts/helpers.py:5-20) and its tuner explores lambda down to 1e-4 (scripts/tune_params.py:100-101), where
// rows with fewer ratings than k have cond(A) ~ |G| / lambda: the fp32 rounding of the Gram (1e-7 |G|) is then
// amplified by 1 / lambda into the null-space components of the solution (measured: test RMSE off by 1e-2 at
// lambda = 1e-4).  This kernel keeps the fp32 STORAGE of the factors (so the Gram and the right-hand side
// are formed from the same rounded inputs, consistently) but accumulates F^T F and F^T r in fp64 on the fp64
// matrix cores (v_mfma_f64_16x16x4_f64), factorises and solves in fp64, and rounds only x and the bias to fp32.
//
// One wavefront per task, same task list / perm space / outputs as the fp32 kernel (row_solve.hip):
//   1. Gram: lane (c, q) loads the KB floats F[idx][KB c ...] of rating 4 s + q (position c of every 16-column
//      block), converts them and feeds them to the fp64 MFMA as both operands; at most 28 accumulator blocks
//      (224 registers) are live, wider models take several passes over the row's ratings (block rows).
//      The lower 16x16 blocks go to an LDS image (fp64, 20 KB at k = 64, 110 KB at k = 160).
//   2. Right-looking blocked Cholesky on that image: per 16-column panel every lane takes the panel part of
//      its matrix rows into registers, the panel is eliminated on the VALU (readlane broadcasts), written
//      back, and the rank-16 trailing update runs on the fp64 matrix cores with LDS operands.  The forward
//      substitution rides along (the right-hand side is one more value per lane).
//   3. Transposed solve from the image, bias update, statistics - all fp64.
// Rows longer than ALS_SPLIT_CHUNK: the image of every segment goes to a workspace slot (fp64: slots are
// als_partial_slot_bytes_f64(k) bytes), k_sum_slots_f64 folds them in slot order, k_row_long_f64 finishes.
#include "als_device.hpp"
#include "als_hip.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int blk64(int I, int K) { return I * (I + 1) / 2 + K; }

template <int KB>
struct F64Cfg {
    static constexpr int KP = 16 * KB;
    static constexpr int NACC = KB * (KB + 1) / 2;
    static constexpr int NR = (KP + 63) / 64;
    static constexpr int IMG = NACC * 256;                 // doubles: lower 16x16 blocks, row-major inside a block
    static constexpr int SLOT = IMG + 2 * KP + 2;          // doubles of one partial slot: image, rhs, colsum, sumr, sumr2
    static constexpr int MAXB = 28;                        // accumulator blocks per Gram pass (8 registers each)
    // last block row (exclusive) of the pass that starts at block row I0
    static constexpr int pass_end(int I0) {
        int n = 0, I = I0;
        while (I < KB && (n + I + 1 <= MAXB || I == I0)) { n += I + 1; ++I; }
        return I;
    }
};

__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_d(double v, int src) {
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------
// one Gram pass: block rows [I0, I1) of the lower triangle, all ratings of the task
// ---------------------------------------------------------------------------------------------------------
template <int KB, int I0, int I1, bool FIRST>
__device__ __forceinline__ void gram_pass_f64(const als_row_solve_params& P, int64_t beg, int len, double mu,
                                              double bself, double* __restrict__ img, double (&rhs)[KB],
                                              double (&cs)[KB], double& sumr, double& sumr2, int lane) {
    constexpr int NB = blk64(I1, 0) - blk64(I0, 0);
    constexpr int GS = 4;                                    // rating steps whose gathers are in flight together
    const int c = lane & 15, q = lane >> 4;
    const float* Fc = P.F + KB * c;
    f64x4 acc[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) acc[a] = f64x4{0.0, 0.0, 0.0, 0.0};
    for (int base = 0; base < len; base += 64) {
        const int t = base + lane;
        const bool ok = t < len;
        const int idx = ok ? P.indices[beg + t] : P.F_zero_row;
        double r_l = 0.0;
        if (FIRST) {            // r = R - (mu + b_self + b_other) in fp64 (scripts/als.py:425, 447)
            const float v = ok ? P.vals[beg + t] : 0.f;
            const float bo = ok ? P.bias_other[idx] : 0.f;
            const double rb = ok ? ((double)v - mu - (double)bo) : 0.0;
            sumr += rb;
            sumr2 = fma(rb, rb, sumr2);
            r_l = ok ? rb - bself : 0.0;
        }
        const int off_l = idx * P.ld;
        const int nvalid = min(64, len - base);
#pragma unroll 1
        for (int g0 = 0; g0 < 16; g0 += GS) {
            if (4 * g0 >= nvalid) break;
            float f[GS][KB];
            double r_t[GS];
#pragma unroll
            for (int s = 0; s < GS; ++s) {
                const int off = bperm_i(off_l, 4 * (g0 + s) + q);
                if (FIRST) r_t[s] = bperm_d(r_l, 4 * (g0 + s) + q);
                load_frow<KB>(Fc + (uint32_t)off, f[s]);
            }
#pragma unroll
            for (int s = 0; s < GS; ++s) {
                double fd[KB];
#pragma unroll
                for (int b = 0; b < KB; ++b) fd[b] = (double)f[s][b];
                if (FIRST) {
#pragma unroll
                    for (int b = 0; b < KB; ++b) { rhs[b] = fma(fd[b], r_t[s], rhs[b]); cs[b] += fd[b]; }
                }
#pragma unroll
                for (int I = I0; I < I1; ++I)
#pragma unroll
                    for (int K = 0; K <= I; ++K)
                        acc[blk64(I, K) - blk64(I0, 0)] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            fd[I], fd[K], acc[blk64(I, K) - blk64(I0, 0)], 0, 0, 0);
            }
        }
    }
    // accumulator register i of lane (c, q) is element (row q + 4 i, col c) of its block
#pragma unroll
    for (int I = I0; I < I1; ++I)
#pragma unroll
        for (int K = 0; K <= I; ++K)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                img[blk64(I, K) * 256 + (q + 4 * i) * 16 + c] = acc[blk64(I, K) - blk64(I0, 0)][i];
}

template <int KB, int I0, bool FIRST>
__device__ __forceinline__ void gram_passes_f64(const als_row_solve_params& P, int64_t beg, int len, double mu,
                                                double bself, double* __restrict__ img, double (&rhs)[KB],
                                                double (&cs)[KB], double& sumr, double& sumr2, int lane) {
    if constexpr (I0 < KB) {
        constexpr int I1 = F64Cfg<KB>::pass_end(I0);
        gram_pass_f64<KB, I0, I1, FIRST>(P, beg, len, mu, bself, img, rhs, cs, sumr, sumr2, lane);
        gram_passes_f64<KB, I1, false>(P, beg, len, mu, bself, img, rhs, cs, sumr, sumr2, lane);
    }
}

// rhs / colsum from "block b, position c, partial over q" to "perm position i = lane + 64 rr"
template <int KB>
__device__ __forceinline__ void to_rows_f64(double (&rhs)[KB], double (&cs)[KB], double (&rhs_p)[F64Cfg<KB>::NR],
                                            double (&cs_p)[F64Cfg<KB>::NR], int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        rhs[b] += __shfl_xor(rhs[b], 16, 64); rhs[b] += __shfl_xor(rhs[b], 32, 64);
        cs[b] += __shfl_xor(cs[b], 16, 64);   cs[b] += __shfl_xor(cs[b], 32, 64);
    }
#pragma unroll
    for (int rr = 0; rr < F64Cfg<KB>::NR; ++rr) {
        double bsel = 0.0, csel = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (4 * rr + e < KB) {
                bsel = (q == e) ? rhs[4 * rr + e] : bsel;
                csel = (q == e) ? cs[4 * rr + e] : csel;
            }
        rhs_p[rr] = bsel;
        cs_p[rr] = csel;
    }
}

// ---------------------------------------------------------------------------------------------------------
// regularise, factorise, solve / emit the factor.  img: lower blocks of F^T F; rhs_p / cs_p: lane = perm row.
// ---------------------------------------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void finish_row_f64(const als_row_solve_params& P, int row, double* __restrict__ img,
                                               double (&rhs_p)[F64Cfg<KB>::NR], double (&cs_p)[F64Cfg<KB>::NR],
                                               double sumr, double sumr2, int lane) {
    using C = F64Cfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    const int c = lane & 15, q = lane >> 4;
    const int64_t r64 = row;
    int colrow[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) colrow[rr] = perm_to_col<KB>(min(lane + 64 * rr, KP - 1));

    // by-products: fp32 as the fp32 kernel writes them, or - byproducts_f64 - as doubles for the fp64 W-step
    const bool b64 = P.byproducts_f64 != 0;
    if (P.gram_out) {           // F^T F without lambda, perm space, lower 16x16 blocks
        float* G = P.gram_out + r64 * KP * KP;
        double* G64 = (double*)P.gram_out + r64 * KP * KP;
        for (int I = 0; I < KB; ++I)
            for (int K = 0; K <= I; ++K)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double v = img[blk64(I, K) * 256 + (q + 4 * i) * 16 + c];
                    if (b64) G64[(16 * I + q + 4 * i) * KP + 16 * K + c] = v;
                    else G[(16 * I + q + 4 * i) * KP + 16 * K + c] = (float)v;
                }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        if (i < KP) {
            if (P.rhs_out) { if (b64) ((double*)P.rhs_out)[r64 * KP + i] = rhs_p[rr]; else P.rhs_out[r64 * KP + i] = (float)rhs_p[rr]; }
            if (P.colsum_out) { if (b64) ((double*)P.colsum_out)[r64 * KP + i] = cs_p[rr]; else P.colsum_out[r64 * KP + i] = (float)cs_p[rr]; }
        }
    }
    if (P.sumr_out && lane == 0) { if (b64) ((double*)P.sumr_out)[row] = sumr; else P.sumr_out[row] = (float)sumr; }
    if (P.sumr2_out && lane == 0) { if (b64) ((double*)P.sumr2_out)[row] = sumr2; else P.sumr2_out[row] = (float)sumr2; }

    // regulariser on the diagonal; padded columns get 1 (scripts/als.py:426, 450-455: lambda + 1e-10 (+ alpha D_i))
    const double lam = (double)(P.lambda_row ? P.lambda_row[row] : P.lambda_scalar) + 1e-10
                     + (double)(P.diag_extra ? P.diag_extra[row] : 0.f);
    if (q == 0) {
        for (int J = 0; J < KB; ++J)
            img[blk64(J, J) * 256 + c * 16 + c] += (perm_to_col<KB>(16 * J + c) < P.k) ? lam : 1.0;
    }
    wave_lds_sync();

    double b[NR], y[NR], dinv[NR], rhs0[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        rhs0[rr] = rhs_p[rr];
        b[rr] = rhs_p[rr];
        if (P.rhs_extra && !P.factor_out && lane + 64 * rr < KP) b[rr] += (double)P.rhs_extra[r64 * P.ld + colrow[rr]];
        y[rr] = 0.0; dinv[rr] = 0.0;
    }
    bool bad = false;

    // ---- blocked Cholesky, 16-column panels --------------------------------------------------------------
#pragma unroll 1
    for (int J = 0; J < KB; ++J) {
        // the panel part of the lane's matrix rows: p[rr][t] = A[i][16 J + t], i = lane + 64 rr >= 16 J
        double p[NR][16];
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = min(max(lane + 64 * rr, 16 * J), KP - 1);        // lanes above the panel: dummy row
            const double* src = img + blk64(i >> 4, J) * 256 + (i & 15) * 16;
#pragma unroll
            for (int t = 0; t < 16; ++t) p[rr][t] = src[t];
        }
#pragma unroll
        for (int T = 0; T < 16; ++T) {
            const int piv = 16 * J + T, RP = piv >> 6, LP = piv & 63;
            double dsel = p[0][T], bsel = b[0];
#pragma unroll
            for (int rr = 1; rr < NR; ++rr) { dsel = (RP == rr) ? p[rr][T] : dsel; bsel = (RP == rr) ? b[rr] : bsel; }
            const double d = readlane_d(dsel, LP);
            bad = bad || !(d > 0.0);                                         // not positive definite (or NaN)
            const double inv = 1.0 / __builtin_sqrt(d);
            double l[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) { l[rr] = p[rr][T] * inv; p[rr][T] = l[rr]; }
            const double yt = readlane_d(bsel, LP) * inv;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                b[rr] = fma(-l[rr], yt, b[rr]);
                const bool own = (RP == rr) && (lane == LP);
                y[rr] = own ? yt : y[rr];
                dinv[rr] = own ? inv : dinv[rr];
            }
#pragma unroll
            for (int t2 = T + 1; t2 < 16; ++t2) {
                const int pr = 16 * J + t2, R2 = pr >> 6, L2 = pr & 63;
                double lsel = l[0];
#pragma unroll
                for (int rr = 1; rr < NR; ++rr) lsel = (R2 == rr) ? l[rr] : lsel;
                const double mlt = readlane_d(lsel, L2);                     // L[16 J + t2][16 J + T]
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) p[rr][t2] = fma(-l[rr], mlt, p[rr][t2]);
            }
        }
        // L block column J back to the image (rows at or below the panel; the diagonal block's upper part is
        // never read)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            if (i >= 16 * J && i < KP) {
                double* dst = img + blk64(i >> 4, J) * 256 + (i & 15) * 16;
#pragma unroll
                for (int t = 0; t < 16; ++t) dst[t] = p[rr][t];
            }
        }
        wave_lds_sync();
        // trailing update on the fp64 matrix cores: block (I, K) -= L_IJ L_KJ^T for J < K <= I.
        // operands: A[row = c][k = q + 4 s] = L_IJ[c][4 s + q], B[k][col = c] = L_KJ[c][4 s + q]
        for (int I = J + 1; I < KB; ++I) {
            double aop[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) aop[s] = img[blk64(I, J) * 256 + c * 16 + 4 * s + q];
            for (int K = J + 1; K <= I; ++K) {
                double* Cb = img + blk64(I, K) * 256;
                f64x4 acc;
                double bop[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = Cb[(q + 4 * i) * 16 + c];
#pragma unroll
                for (int s = 0; s < 4; ++s) bop[s] = -img[blk64(K, J) * 256 + c * 16 + 4 * s + q];
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[s], bop[s], acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) Cb[(q + 4 * i) * 16 + c] = acc[i];
            }
        }
        wave_lds_sync();
    }
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicMax(P.status, row + 1);

    if (P.factor_out) {
        // symmetric completion of L with 1 / L_ii on the diagonal, perm space, fp32 (input of the sweep kernels):
        // M[p][i] = L[i][p] (p < i), L[p][i] (p > i)
        float* M = P.factor_out + r64 * KP * KP;
        double* M64 = (double*)P.factor_out + r64 * KP * KP;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            if (i < KP) {
                for (int p = 0; p < KP; ++p) {
                    const int hi = max(p, i), lo = min(p, i);
                    const double v = (p == i) ? dinv[rr] : img[blk64(hi >> 4, lo >> 4) * 256 + (hi & 15) * 16 + (lo & 15)];
                    if (b64) __builtin_nontemporal_store(v, M64 + p * KP + i);
                    else __builtin_nontemporal_store((float)v, M + p * KP + i);
                }
            }
        }
        return;
    }

    // ---- L^T x = y: lane (+64 rr) owns unknown i and reads its column L[p][i], p > i -----------------------
    double xs[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) xs[rr] = y[rr];
#pragma unroll 1
    for (int p = KP - 1; p >= 0; --p) {
        const int RP = p >> 6, LP = p & 63;
        double xsel = xs[0] * dinv[0];
#pragma unroll
        for (int rr = 1; rr < NR; ++rr) xsel = (RP == rr) ? xs[rr] * dinv[rr] : xsel;
        const double xp = readlane_d(xsel, LP);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            const bool below = i < p;
            const int ic = below ? i : 0;
            const double lpi = img[blk64(p >> 4, ic >> 4) * 256 + (p & 15) * 16 + (ic & 15)];
            xs[rr] = below ? fma(-lpi, xp, xs[rr]) : ((i == p) ? xp : xs[rr]);
        }
    }

    double dot = 0.0, xr = 0.0, yy = 0.0, xx = 0.0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (lane + 64 * rr < KP) {
            P.X_out[r64 * P.ld + colrow[rr]] = (float)xs[rr];
            dot = fma(cs_p[rr], xs[rr], dot);
            xr = fma(rhs0[rr], xs[rr], xr);
            yy = fma(y[rr], y[rr], yy);
            xx = fma(xs[rr], xs[rr], xx);
        }
    }
    dot = wave_sum_f64(dot);
    const double nnz = (double)(P.indptr[row + 1] - P.indptr[row]);
    const double lb = (double)(P.lambda_bias_row ? P.lambda_bias_row[row] : P.lambda_bias_scalar);
    const double bnew = (sumr - dot) / (nnz + lb + 1e-10);                   // scripts/als.py:431-433, 464-466
    const double bold = (double)P.bias_self[row];           // read before the store: bias_out may alias bias_self
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) P.bias_out[row] = (float)bnew;
    if (P.stat_out) {           // closed-form residual sums (DESIGN.md "Statistics"), here without fp32 cancellation
        xr = wave_sum_f64(xr); yy = wave_sum_f64(yy); xx = wave_sum_f64(xx);
        if (lane == 0) {
            const double s1 = sumr - nnz * bnew;
            const double s2 = sumr2 - 2.0 * bnew * sumr + nnz * bnew * bnew;
            const double cross = xr + (bold - bnew) * dot;
            const double quad = yy - lam * xx;
            P.stat_out[2 * r64] = (float)(s1 - dot);
            P.stat_out[2 * r64 + 1] = (float)(s2 - 2.0 * cross + quad);
        }
    }
}

template <int KB>
__global__ __launch_bounds__(64)
void k_row_tasks_f64(const als_row_solve_params P) {
    using C = F64Cfg<KB>;
    __shared__ __attribute__((aligned(16))) double img[C::IMG];
    const int lane = threadIdx.x;
    const int64_t tid = blockIdx.x;
    if (tid >= P.ntasks) return;
    const als_task t = P.tasks[tid];
    const int row = t.row;
    const int64_t rbeg = P.indptr[row], rend = P.indptr[row + 1];
    const int64_t beg = rbeg + (int64_t)t.seg * ALS_SPLIT_CHUNK;
    const int len = (int)min((int64_t)ALS_SPLIT_CHUNK, rend - beg);
    const double mu = *P.mu;
    const double bself = (double)P.bias_self[row];
    double rhs[KB], cs[KB], sumr = 0.0, sumr2 = 0.0;
#pragma unroll
    for (int b = 0; b < KB; ++b) { rhs[b] = 0.0; cs[b] = 0.0; }
    gram_passes_f64<KB, 0, true>(P, beg, len, mu, bself, img, rhs, cs, sumr, sumr2, lane);
    double rhs_p[C::NR], cs_p[C::NR];
    to_rows_f64<KB>(rhs, cs, rhs_p, cs_p, lane);
    sumr = wave_sum_f64(sumr);
    sumr2 = wave_sum_f64(sumr2);
    wave_lds_sync();
    if (t.slot >= 0) {          // segment of a split row: image + vectors to the workspace slot
        double* ws = (double*)P.workspace + (size_t)t.slot * C::SLOT;
        for (int e = lane; e < C::IMG; e += 64) ws[e] = img[e];
#pragma unroll
        for (int rr = 0; rr < C::NR; ++rr)
            if (lane + 64 * rr < C::KP) {
                ws[C::IMG + lane + 64 * rr] = rhs_p[rr];
                ws[C::IMG + C::KP + lane + 64 * rr] = cs_p[rr];
            }
        if (lane == 0) { ws[C::IMG + 2 * C::KP] = sumr; ws[C::IMG + 2 * C::KP + 1] = sumr2; }
        return;
    }
    finish_row_f64<KB>(P, row, img, rhs_p, cs_p, sumr, sumr2, lane);
}

// partial slots of a split row summed into its first slot, one thread per element, slots in ascending order
template <int KB>
__global__ __launch_bounds__(256)
void k_sum_slots_f64(const als_long_row* __restrict__ long_rows, double* __restrict__ workspace) {
    constexpr int N = F64Cfg<KB>::SLOT;
    const als_long_row lr = long_rows[blockIdx.x];
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= N || lr.nslots < 2) return;
    double* w0 = workspace + (size_t)lr.slot0 * N + e;
    double acc = w0[0];
    for (int s = 1; s < lr.nslots; ++s) acc += w0[(size_t)s * N];
    w0[0] = acc;
}

template <int KB>
__global__ __launch_bounds__(64)
void k_row_long_f64(const als_row_solve_params P) {
    using C = F64Cfg<KB>;
    __shared__ __attribute__((aligned(16))) double img[C::IMG];
    const int lane = threadIdx.x;
    if ((int64_t)blockIdx.x >= P.nlong) return;
    const als_long_row lr = P.long_rows[blockIdx.x];
    const double* ws = (const double*)P.workspace + (size_t)lr.slot0 * C::SLOT;
    for (int e = lane; e < C::IMG; e += 64) img[e] = ws[e];
    double rhs_p[C::NR], cs_p[C::NR];
#pragma unroll
    for (int rr = 0; rr < C::NR; ++rr) {
        const int i = min(lane + 64 * rr, C::KP - 1);
        rhs_p[rr] = ws[C::IMG + i];
        cs_p[rr] = ws[C::IMG + C::KP + i];
    }
    const double sumr = ws[C::IMG + 2 * C::KP], sumr2 = ws[C::IMG + 2 * C::KP + 1];
    wave_lds_sync();
    finish_row_f64<KB>(P, lr.row, img, rhs_p, cs_p, sumr, sumr2, lane);
}

// Rows handed over by the fp32 kernels (als_row_solve_params::cond_limit): the first *redo_count entries of
// redo_rows, each redone as ONE task whatever its length (no segments: the fp64 Gram has no accumulation-length
// issue and such rows are few), by a fixed grid of waves that walk the list - the count is only known on the
// device.  Same outputs as the fp32 kernel would have written.
template <int KB>
__global__ __launch_bounds__(64)
void k_row_redo_f64(const als_row_solve_params P) {
    using C = F64Cfg<KB>;
    __shared__ __attribute__((aligned(16))) double img[C::IMG];
    const int lane = threadIdx.x;
    const int count = *P.redo_count;
    const double mu = *P.mu;
    for (int e = blockIdx.x; e < count; e += gridDim.x) {
        const int row = P.redo_rows[e];
        const int64_t beg = P.indptr[row];
        const int len = (int)(P.indptr[row + 1] - beg);
        const double bself = (double)P.bias_self[row];
        double rhs[KB], cs[KB], sumr = 0.0, sumr2 = 0.0;
#pragma unroll
        for (int b = 0; b < KB; ++b) { rhs[b] = 0.0; cs[b] = 0.0; }
        gram_passes_f64<KB, 0, true>(P, beg, len, mu, bself, img, rhs, cs, sumr, sumr2, lane);
        double rhs_p[C::NR], cs_p[C::NR];
        to_rows_f64<KB>(rhs, cs, rhs_p, cs_p, lane);
        sumr = wave_sum_f64(sumr);
        sumr2 = wave_sum_f64(sumr2);
        wave_lds_sync();
        finish_row_f64<KB>(P, row, img, rhs_p, cs_p, sumr, sumr2, lane);
        wave_lds_sync();                    // the image is reused by the next row
    }
}

template <int KB>
int launch_row_redo_f64(const als_row_solve_params* p, hipStream_t st) {
    // enough waves to fill the device when many rows were flagged, cheap when none was (they exit at once)
    static int grid_of[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ALS_E_LAUNCH;
    if (grid_of[dev] == 0) {
        int nb = 0;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_row_redo_f64<KB>, 64, 0) != hipSuccess || nb < 1)
            return ALS_E_LAUNCH;
        grid_of[dev] = prop.multiProcessorCount * (nb > 8 ? 8 : nb);
    }
    hipLaunchKernelGGL(k_row_redo_f64<KB>, dim3((unsigned)grid_of[dev]), dim3(64), 0, st, *p);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

template <int KB>
int launch_row_solve_f64(const als_row_solve_params* p, hipStream_t st) {
    using C = F64Cfg<KB>;
    if (p->ntasks > 0)
        hipLaunchKernelGGL(k_row_tasks_f64<KB>, dim3((unsigned)p->ntasks), dim3(64), 0, st, *p);
    if (p->nlong > 0) {
        hipLaunchKernelGGL(k_sum_slots_f64<KB>, dim3((unsigned)p->nlong, (C::SLOT + 255) / 256), dim3(256), 0, st,
                           p->long_rows, (double*)p->workspace);
        hipLaunchKernelGGL(k_row_long_f64<KB>, dim3((unsigned)p->nlong), dim3(64), 0, st, *p);
    }
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

extern "C" int64_t als_partial_slot_bytes_f64(int k) {
    const int ld = als_padded_k(k);
    if (ld < 0) return ALS_E_BADK;
    const int KB = ld / 16;
    return (int64_t)(KB * (KB + 1) / 2 * 256 + 2 * ld + 2) * sizeof(double);
}

// called by als_row_solve (row_solve.hip) after the fp32 launches of a call with cond_limit > 0
int als_row_redo_f64_dispatch(const als_row_solve_params* p, hipStream_t st) {
    switch (p->ld / 16) {
        case 1: return launch_row_redo_f64<1>(p, st);
        case 2: return launch_row_redo_f64<2>(p, st);
        case 3: return launch_row_redo_f64<3>(p, st);
        case 4: return launch_row_redo_f64<4>(p, st);
        case 5: return launch_row_redo_f64<5>(p, st);
        case 6: return launch_row_redo_f64<6>(p, st);
        case 7: return launch_row_redo_f64<7>(p, st);
        case 8: return launch_row_redo_f64<8>(p, st);
        case 9: return launch_row_redo_f64<9>(p, st);
        case 10: return launch_row_redo_f64<10>(p, st);
    }
    return ALS_E_BADK;
}

// called by als_row_solve (row_solve.hip) after its argument checks when gram_mode == ALS_GRAM_F64
int als_row_solve_f64_dispatch(const als_row_solve_params* p, hipStream_t st) {
    switch (p->ld / 16) {
        case 1: return launch_row_solve_f64<1>(p, st);
        case 2: return launch_row_solve_f64<2>(p, st);
        case 3: return launch_row_solve_f64<3>(p, st);
        case 4: return launch_row_solve_f64<4>(p, st);
        case 5: return launch_row_solve_f64<5>(p, st);
        case 6: return launch_row_solve_f64<6>(p, st);
        case 7: return launch_row_solve_f64<7>(p, st);
        case 8: return launch_row_solve_f64<8>(p, st);
        case 9: return launch_row_solve_f64<9>(p, st);
        case 10: return launch_row_solve_f64<10>(p, st);
    }
    return ALS_E_BADK;
}
