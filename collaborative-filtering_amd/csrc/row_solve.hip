// K1: per-row normal-equation build (MFMA Gram) + blocked Cholesky solve + bias update.
//
// Replaces the bodies of the user loop (reference scripts/als.py:414-433), the
// item loop (scripts/als.py:436-466) and cholesky_solve (scripts/helpers.py:5-20).
//
// One wavefront owns one task (a row, or a <=4096-rating segment of a long row).
//
// Gram.  The gathered factor rows never touch LDS on their way to the matrix
// cores: lane (c,q) loads the KB contiguous floats F[idx][KB*c..] of rating
// 4*s+q, which in "perm space" (als_device.hpp) are position c of every
// 16-column block, i.e. exactly the A/B operands of v_mfma_f32_16x16x4_f32.
// Only the KB(KB+1)/2 lower 16x16 blocks are accumulated (symmetry).
//
// Cholesky.  Right-looking, 16-column panels.  The accumulators that hold the
// Gram ARE the trailing matrix and stay in registers in MFMA C/D layout.  Per
// panel J: its block column is dumped to LDS, every lane picks up its row of the
// panel (16 registers), the panel is factorised on the VALU (pivot and
// multipliers broadcast with v_readlane), written back, and the rank-16 update
// of all trailing blocks runs on the matrix cores with operands read straight
// from the panel image in LDS.  The forward substitution rides along in the
// panel loop (the right-hand side is one more value per lane), so only the
// transposed solve is a separate pass.  VALU work is O(k^2 * 16), the O(k^3)
// part is MFMA.  L lives in LDS as swizzled block columns: 10 KB at k = 64.
#include <type_traits>
#include "als_device.hpp"
#include "als_hip.h"
#include "row_common.hpp"

namespace {

template <int KB>
struct RowAcc {
    f32x4 acc[KCfg<KB>::NACC];      // lower blocks, index I(I+1)/2 + K  (K <= I)
    float rhs[KB];
    float cs[KB];
    float sumr;
    float sumr2;
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int a = 0; a < KCfg<KB>::NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < KB; ++b) { rhs[b] = 0.f; cs[b] = 0.f; }
        sumr = 0.f; sumr2 = 0.f;
    }
};

// ---------------------------------------------------------------------------
// 64 ratings: lane t holds (off_l, r_l) of rating t; 16 steps of 4 ratings.
// off_l is the element offset of the rating's factor row (idx * ld); ratings
// beyond the end of the row point at F's all-zero row with r = 0, so they add
// nothing and no masking is needed.
// ---------------------------------------------------------------------------
template <int KB, bool FULL>
__device__ __forceinline__ void process_chunk(RowAcc<KB>& A, int off_l, float r_l, int nvalid,
                                              const float* __restrict__ Fc, int q) {
    constexpr int GS = KCfg<KB>::GS;
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += GS) {
        if (!FULL && 4 * g0 >= nvalid) break;
        float f[GS][KB];
        int off_t[GS];
        float r_t[GS];
        // all cross-lane moves of the group first (one LDS-crossbar round trip), then the loads
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            const int src = FULL ? 16 * q + g0 + s : 4 * (g0 + s) + q;     // full chunks: rows 16 ratings apart (see below)
            off_t[s] = bperm_i(off_l, src);
            r_t[s] = bperm_f(r_l, src);
        }
#pragma unroll
        for (int s = 0; s < GS; ++s)
            if (FULL || 4 * (g0 + s) < nvalid) load_frow<KB>(Fc + (uint32_t)off_t[s], f[s]);
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            if (FULL || 4 * (g0 + s) < nvalid) {
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    A.rhs[b] = fmaf(f[s][b], r_t[s], A.rhs[b]);
                    A.cs[b] += f[s][b];
                }
#pragma unroll
                for (int bi = 0; bi < KB; ++bi)
#pragma unroll
                    for (int bj = 0; bj <= bi; ++bj)
                        A.acc[blk_idx(bi, bj)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                            f[s][bi], f[s][bj], A.acc[blk_idx(bi, bj)], 0, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Same 64 ratings on the fp16 matrix cores (v_mfma_f32_16x16x32_f16), fp32-equivalent.
// Every gathered float is scaled by S (an exact power of two, one per factor matrix: als_factor_scale) and split
// into two fp16 terms by rounding to nearest,
//     x S = h + l + e,   h = fp16(x S),  l = fp16(x S - h),  |e| <= 2^-23 |x S|         (split2, row_common.hpp)
// and a Gram block is the fp32-accumulated sum of the three cross products hh + hl + lh (all exact in fp32:
// 11 x 11 bits); the dropped ll is <= 2^-24 relative.  The accumulators hold S^2 G and are unscaled once per
// row (finish_row).  Measured against fp64 (profiles/ubench/gram_f16x2.hip, profiles/r03_ubench_gram_f16x2.txt):
// the same error as the exact 3-way bf16 split of rounds 1-2 (six products per block) and as the f32 MFMA - at
// half the matrix instructions and 6 instead of 11 VALU per pair of elements (856 instead of 1522 cycles per
// 32-rating group and SIMD).  The 16-bit matrix cores run beside the VALU (f32 MFMA does not - DESIGN.md
// section 4).  Lane (c,q) takes ratings 8q..8q+7 of a 32-rating group: the 8 values it loads per block are
// exactly its 8 k-elements of the A/B operand.
// ---------------------------------------------------------------------------

template <int KB, bool FULL>
__device__ __forceinline__ void process_chunk_f16x2(RowAcc<KB>& A, int off_l, float r_l, int nvalid,
                                                    const float* __restrict__ Fc, int q, float S) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!FULL && 32 * g >= nvalid) break;
        int off_t[8];
        float r_t[8];
        float f[8][KB];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // which rating goes to which k-element is free (A and B operands use the same assignment).  Full
            // chunks: lane group q takes ratings 16 q + 8 g + j, so the four rows ONE gather instruction fetches are
            // 16 ratings apart (8 apart with 32 g + 8 q + j: cfg 4 V-step +2.7 %, U-step +1.5 %; 1 apart: V-step
            // +7 % - profiles/r03_ab_gather_row_spread.txt).  A partial chunk keeps its valid ratings in group 0.
            const int src = FULL ? 16 * q + 8 * g + j : 32 * g + 8 * q + j;
            off_t[j] = bperm_i(off_l, src);
            r_t[j] = bperm_f(r_l, src);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) load_frow<KB>(Fc + (uint32_t)off_t[j], f[j]);
        i32x4 H[KB], L[KB];
#pragma unroll
        for (int b = 0; b < KB; ++b) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float x0 = f[j][b], x1 = f[j + 1][b];
                A.rhs[b] = fmaf(x0, r_t[j], A.rhs[b]);
                A.rhs[b] = fmaf(x1, r_t[j + 1], A.rhs[b]);
                A.cs[b] += x0 + x1;
                int hw, lw;
                split2(x0, x1, S, hw, lw);
                H[b][j >> 1] = hw; L[b][j >> 1] = lw;
            }
        }
#pragma unroll
        for (int bi = 0; bi < KB; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                f32x4 acc = A.acc[blk_idx(bi, bj)];
                const h16x8 hi = __builtin_bit_cast(h16x8, H[bi]), hj = __builtin_bit_cast(h16x8, H[bj]);
                const h16x8 li = __builtin_bit_cast(h16x8, L[bi]), lj = __builtin_bit_cast(h16x8, L[bj]);
                // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(li, hj, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, lj, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, hj, acc, 0, 0, 0);
                A.acc[blk_idx(bi, bj)] = acc;
            }
    }
}

// ---------------------------------------------------------------------------
// The same Gram from PRE-SPLIT operands (MODE 2, k = 49 ... 64; round 3).  als_row_solve first writes the two fp16
// terms of every element of the gathered table into a table of the same size and layout (`F_planes`, k_split_planes:
// per row and 16-lane column group c one 16-byte piece = (h0 h1)(h2 h3)(l0 l1)(l2 l3)); the gather then fetches
// h and l directly - the same 16 bytes per lane and rating - and what is left of the vector work is the 4 x 8
// transposition into the operand registers (32 v_perm_b32 per 32 ratings) instead of the split (96) and the
// right-hand side / column sums (64), which go to the matrix cores as well: one more operand R with the rows
// (r_hi, 1, r_lo, 0 ...) - the residuals split in two fp16 terms once per 64-rating chunk, lane per rating - gives
// E_b = R (H_b + L_b) per column block: row 0 + row 2 = S F^T r, row 1 = S F^T 1, found in the registers 0, 1, 2 of
// the lanes q = 0 at the very position finish_row expects the lane's partial sums.  Worth it where the gathered
// table is small and the launch is bound by vector issue - the U-step: ~60 instead of ~293 vector instructions
// and 38 instead of 30 matrix instructions per 32 ratings.
// ---------------------------------------------------------------------------
template <int KB, bool FULL>
__device__ __forceinline__ void process_chunk_planes(RowAcc<KB>& A, f32x4 (&E)[KB], int off_l, int rp_l, int nvalid,
                                                     const uint32_t* __restrict__ Pc, int q, int sel_c, int ones_c) {
    static_assert(KB == 4 || KB == 8, "pre-split operands: k = 49 ... 64 and 113 ... 128");
    constexpr int NW = KB / 2;              // words of h (and of l) per rating and lane: KB halves each
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        if (!FULL && 32 * g >= nvalid) break;
        int off_t[8], rp_t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int src = FULL ? 16 * q + 8 * g + j : 32 * g + 8 * q + j;
            off_t[j] = bperm_i(off_l, src);
            rp_t[j] = bperm_i(rp_l, src);
        }
        uint32_t w[8][KB];                   // per rating: (h pairs)[NW] then (l pairs)[NW]
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int v4 = 0; v4 < KB / 4; ++v4) {
                const u32x4 t = *reinterpret_cast<const u32x4*>(Pc + (uint32_t)off_t[j] + 4 * v4);
                w[j][4 * v4] = t[0]; w[j][4 * v4 + 1] = t[1]; w[j][4 * v4 + 2] = t[2]; w[j][4 * v4 + 3] = t[3];
            }
        i32x4 H[KB], L[KB], R;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            R[d] = (int)__builtin_amdgcn_perm((uint32_t)rp_t[2 * d + 1], (uint32_t)rp_t[2 * d], (uint32_t)sel_c) | ones_c;
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const uint32_t sel = (b & 1) ? 0x07060302u : 0x05040100u;
                H[b][d] = (int)__builtin_amdgcn_perm(w[2 * d + 1][b >> 1], w[2 * d][b >> 1], sel);
                L[b][d] = (int)__builtin_amdgcn_perm(w[2 * d + 1][NW + (b >> 1)], w[2 * d][NW + (b >> 1)], sel);
            }
        }
        const h16x8 rr = __builtin_bit_cast(h16x8, R);
#pragma unroll
        for (int bi = 0; bi < KB; ++bi) {
            const h16x8 hi = __builtin_bit_cast(h16x8, H[bi]), li = __builtin_bit_cast(h16x8, L[bi]);
            E[bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rr, li, E[bi], 0, 0, 0);
            E[bi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rr, hi, E[bi], 0, 0, 0);
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                f32x4 acc = A.acc[blk_idx(bi, bj)];
                const h16x8 hj = __builtin_bit_cast(h16x8, H[bj]), lj = __builtin_bit_cast(h16x8, L[bj]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(li, hj, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, lj, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, hj, acc, 0, 0, 0);
                A.acc[blk_idx(bi, bj)] = acc;
            }
        }
    }
}

// E (rows 0, 1, 2 of the lanes q = 0) -> the lane's partial right-hand side / column sums, unscaled; E restarts at zero
template <int KB>
__device__ __forceinline__ void fold_planes_rhs(RowAcc<KB>& A, f32x4 (&E)[KB], int q, float inv_s) {
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        A.rhs[b] += (q == 0) ? (E[b][0] + E[b][2]) * inv_s : 0.f;
        A.cs[b] += (q == 0) ? E[b][1] * inv_s : 0.f;
        E[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Long rows in f16x2 mode: the 16-bit MFMA's internal accumulation error grows linearly with the
// number of accumulated groups (measured with the bf16 form of rounds 1-2: 1e-5 relative at 4000 ratings), so
// every FLUSH_GROUPS*32 ratings the accumulators are added (fp32, round-to-nearest) into totals
// kept in the wave's LDS region - idle during the Gram phase and exactly NACC*256 floats - and
// restarted from zero.
constexpr int FLUSH_GROUPS = 16;

template <int KB>
__device__ __forceinline__ void flush_acc(RowAcc<KB>& A, float* __restrict__ Ls, int lane, bool first) {
    // (`first` is wave-uniform; all the reads of the running totals are issued before the first add so that
    // the flush costs one LDS round trip, not one per value)
    if (!first) {
#pragma unroll
        for (int a = 0; a < KCfg<KB>::NACC; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) A.acc[a][r] += Ls[(a * 4 + r) * 64 + lane];
    }
#pragma unroll
    for (int a = 0; a < KCfg<KB>::NACC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            Ls[(a * 4 + r) * 64 + lane] = A.acc[a][r];
            A.acc[a][r] = 0.f;
        }
}

template <int KB>
__device__ __forceinline__ void unflush_acc(RowAcc<KB>& A, const float* __restrict__ Ls, int lane) {
#pragma unroll
    for (int a = 0; a < KCfg<KB>::NACC; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) A.acc[a][r] += Ls[(a * 4 + r) * 64 + lane];
}

template <int KB, int MODE>
__device__ __forceinline__ void gram_accumulate(RowAcc<KB>& A, const int32_t* __restrict__ idxp,
                                                const float* __restrict__ valp, int len,
                                                const float* __restrict__ F, int ld, int zero_row,
                                                const float* __restrict__ bias_other, float mu,
                                                float bself, int lane, float* __restrict__ Ls, float S,
                                                const uint32_t* __restrict__ planes = nullptr) {
    const int c = lane & 15, q = lane >> 4;
    const float* Fc = F + KB * c;
    int nflush = 0;
    // MODE 2 (pre-split operands, KB == 4): extra accumulators for R (H + L), the operand-R selectors of this lane
    f32x4 E[MODE == 2 ? KB : 1];
    const uint32_t* Pc = planes + KB * c;
    const int sel_c = (c == 0) ? 0x05040100 : (c == 2) ? 0x07060302 : 0x0c0c0c0c;
    const int ones_c = (c == 1) ? 0x3C003C00 : 0;
    const float inv_s = __int_as_float((254 - ((__float_as_int(S) >> 23) & 0xff)) << 23);      // S is a power of two
    if constexpr (MODE == 2) {
#pragma unroll
        for (int b = 0; b < KB; ++b) E[b] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // software pipeline over 64-rating chunks: indices two chunks ahead,
    // value + opposite-side bias one chunk ahead of the factor-row gathers
    // indices / values are read once: non-temporal, so that they do not push the gathered factor rows (244 MiB of
    // U at cfg 4, just under the 256 MiB Infinity Cache) out of the last-level cache
    int idx1 = (lane < len) ? __builtin_nontemporal_load(idxp + lane) : zero_row;
    int idx2 = (64 + lane < len) ? __builtin_nontemporal_load(idxp + 64 + lane) : zero_row;
    float val1 = (lane < len) ? __builtin_nontemporal_load(valp + lane) : 0.f;
    float bo1 = (lane < len) ? bias_other[idx1] : 0.f;
    for (int base = 0; base < len; base += 64) {
        const int idx0 = idx1;
        const float val0 = val1, bo0 = bo1;
        idx1 = idx2;
        const int t2 = base + 128 + lane;
        idx2 = (t2 < len) ? __builtin_nontemporal_load(idxp + t2) : zero_row;
        const int t1 = base + 64 + lane;
        val1 = (t1 < len) ? __builtin_nontemporal_load(valp + t1) : 0.f;
        bo1 = (t1 < len) ? bias_other[idx1] : 0.f;
        const int nvalid = min(64, len - base);
        const bool ok = lane < nvalid;
        const float rb = val0 - mu - bo0;
        A.sumr += ok ? rb : 0.f;
        A.sumr2 = ok ? fmaf(rb, rb, A.sumr2) : A.sumr2;
        const float r0 = ok ? (rb - bself) : 0.f;
        const int off0 = idx0 * ld;                       // < 2^31 elements (checked by the launcher)
        if constexpr (MODE == 0) {
            if (nvalid == 64) process_chunk<KB, true>(A, off0, r0, 64, Fc, q);
            else              process_chunk<KB, false>(A, off0, r0, nvalid, Fc, q);
        } else {
            if constexpr (MODE == 2) {
                // the residual of the lane's rating in two fp16 terms, one dword: (r_hi, r_lo)
                const _Float16 rh = (_Float16)r0;
                const _Float16 rl = (_Float16)(r0 - (float)rh);
                const int rp = (int)__builtin_bit_cast(unsigned short, rh) | ((int)__builtin_bit_cast(unsigned short, rl) << 16);
                if (nvalid == 64) process_chunk_planes<KB, true>(A, E, off0, rp, 64, Pc, q, sel_c, ones_c);
                else              process_chunk_planes<KB, false>(A, E, off0, rp, nvalid, Pc, q, sel_c, ones_c);
            } else {
                if (nvalid == 64) process_chunk_f16x2<KB, true>(A, off0, r0, 64, Fc, q, S);
                else              process_chunk_f16x2<KB, false>(A, off0, r0, nvalid, Fc, q, S);
            }
            if (((base >> 6) + 1) % (FLUSH_GROUPS / 2) == 0 && base + 64 < len) {
                mfma_results_settle();
                if constexpr (MODE == 2) fold_planes_rhs<KB>(A, E, q, inv_s);
                flush_acc<KB>(A, Ls, lane, nflush == 0);
                ++nflush;
            }
        }
    }
    if (MODE != 0) mfma_results_settle();         // whoever reads the accumulators next (totals, partial slot, finish_row)
    if constexpr (MODE == 2) fold_planes_rhs<KB>(A, E, q, inv_s);
    if (MODE != 0 && nflush > 0) {
        unflush_acc<KB>(A, Ls, lane);
        wave_lds_sync();
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
    ws[(it++) * 64 + lane] = A.sumr;
    ws[it * 64 + lane] = A.sumr2;
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
    A.sumr += ws[(it++) * 64 + lane];
    A.sumr2 += ws[it * 64 + lane];
}

template <int KB, int J, int T, bool SOLVE>
__device__ __forceinline__ void panel_pivot(float (&p)[KCfg<KB>::NR][16], Chol<KB>& S, int lane) {
    constexpr int NR = KCfg<KB>::NR;
    constexpr int PIV = 16 * J + T, RP = PIV >> 6, LP = PIV & 63;
    constexpr int DR = (16 * J) >> 6, DL = (16 * J) & 63;
    // unscaled pivot column of the diagonal block, replicated into every 16-lane row: issued before the
    // rsqrt chain so the LDS round trip overlaps it (and the previous pivot's FMAs)
    const float prep = bperm_f(p[DR][T], DL + (lane & 15));
    const float d = readlane_f(p[RP][T], LP);
    const float inv = __builtin_amdgcn_rsqf(d);            // v_rsq_f32: 1 ulp, ample for the fp32 tolerance
    float l[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) { l[rr] = p[rr][T] * inv; p[rr][T] = l[rr]; }
    // multipliers L[16J+t2][16J+T] = lane t2 of every 16-lane row of lrep: each FMA picks its own with a DPP
    // row_newbcast:t2 operand - no v_readlane per multiplier
    float lrep = -(prep * inv);          // the sign of the update rides on the replicated column
    asm("s_nop 1" : "+v"(lrep));        // VALU write -> DPP read: 2 wait states, invisible to the hazard recogniser
    float yt = 0.f;
    if constexpr (SOLVE) yt = readlane_f(S.b[RP], LP) * inv;
    panel_trailing<KB, T + 1>(p, l, lrep);
    if constexpr (SOLVE) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) S.b[rr] = fmaf(-l[rr], yt, S.b[rr]);
        S.y[RP] = select_lanes<1ull << LP>(yt, S.y[RP]);
    }
}

template <int KB, int J, int T, bool SOLVE>
__device__ __forceinline__ void panel_pivots(float (&p)[KCfg<KB>::NR][16], Chol<KB>& S, int lane) {
    if constexpr (T < 16) {
        panel_pivot<KB, J, T, SOLVE>(p, S, lane);
        panel_pivots<KB, J, T + 1, SOLVE>(p, S, lane);
    }
}

// panel J: dump block column -> per-lane rows -> factorise -> write back -> MFMA trailing update
template <int KB, int J, bool SOLVE>
__device__ __forceinline__ void chol_panel(RowAcc<KB>& A, Chol<KB>& S, float* __restrict__ Ls, int lane) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    constexpr int OFF = C::lcol_off(J);
    const int c = lane & 15, q = lane >> 4;
    // 1. accumulators (C/D layout: row 4q + r, col c) of block column J -> LDS panel image
    {
        const int swzc = (((c >> 2) ^ q) << 2) + (c & 3);            // (row >> 2) & 3 == q here
#pragma unroll
        for (int I = J; I < KB; ++I)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Ls[OFF + (16 * (I - J) + 4 * q + r) * 16 + swzc] = A.acc[blk_idx(I, J)][r];
    }
    wave_lds_sync();
    // 2. every lane takes the panel part of its matrix rows
    float p[NR][16];
    const int si = (lane >> 2) & 3;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        const int rowi = min(max(i, 16 * J), KP - 1) - 16 * J;       // lanes above the panel: dummy row
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Ls + OFF + rowi * 16 + ((g ^ si) << 2));
            p[rr][4 * g] = v.x; p[rr][4 * g + 1] = v.y; p[rr][4 * g + 2] = v.z; p[rr][4 * g + 3] = v.w;
        }
    }
    wave_lds_sync();
    // 3. factorise the panel (rows above a pivot compute garbage in registers nobody reads)
    panel_pivots<KB, J, 0, SOLVE>(p, S, lane);
    // 4. L block column J back to LDS (the diagonal block's upper part is never read)
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        if (i >= 16 * J && i < KP) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {p[rr][4 * g], p[rr][4 * g + 1], p[rr][4 * g + 2], p[rr][4 * g + 3]};
                *reinterpret_cast<f32x4*>(Ls + OFF + (i - 16 * J) * 16 + ((g ^ si) << 2)) = v;
            }
        }
    }
    wave_lds_sync();
    // 1 / L_ii of the 16 pivots of this panel, once per panel: lane 16J + c reads its own diagonal entry back
    // (row c of the block column, swizzle group (c >> 2) ^ (c >> 2) = 0) instead of one select per pivot
    {
        constexpr int DR = (16 * J) >> 6, DL = (16 * J) & 63;
        const float dl = Ls[OFF + c * 16 + (c & 3)];
        S.di[DR] = select_lanes<0xFFFFull << DL>(__builtin_amdgcn_rcpf(dl), S.di[DR]);
    }
    // 5. trailing update on the matrix cores: acc(I,K) -= L_IJ * L_KJ^T for J < K <= I.
    //    Lane (c,q) reads L[16I + c][16J + 4q .. 4q+3]: element e is the operand of MFMA step e
    //    (contraction index 4q + e on both operands).
    if constexpr (J + 1 < KB) {
        f32x4 op[KB - J - 1];
        const int sg = ((q ^ ((c >> 2) & 3)) << 2);
#pragma unroll
        for (int I = J + 1; I < KB; ++I)
            op[I - J - 1] = *reinterpret_cast<const f32x4*>(Ls + OFF + (16 * (I - J) + c) * 16 + sg);
#pragma unroll
        for (int K = J + 1; K < KB; ++K) {
            const f32x4 nb = -op[K - J - 1];
#pragma unroll
            for (int I = K; I < KB; ++I)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    A.acc[blk_idx(I, K)] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        op[I - J - 1][e], nb[e], A.acc[blk_idx(I, K)], 0, 0, 0);
        }
    }
}

template <int KB, int J, bool SOLVE>
__device__ __forceinline__ void chol_panels(RowAcc<KB>& A, Chol<KB>& S, float* __restrict__ Ls, int lane) {
    if constexpr (J < KB) {
        chol_panel<KB, J, SOLVE>(A, S, Ls, lane);
        chol_panels<KB, J + 1, SOLVE>(A, S, Ls, lane);
    }
}

// ---------------------------------------------------------------------------
// Conditioning-driven precision (als_row_solve_params::cond_limit).  After the fp32 factorisation the lanes hold
// 1 / L_ii; kappa = (max L_ii / min L_ii)^2 is a lower bound of cond_2(A) (L_ii^2 are Schur-complement diagonals:
// max_i L_ii^2 <= lambda_max, min_i L_ii^2 >= lambda_min) and within a small factor of it for the matrices met here
// (G + lambda I with rank-deficient or low-rank G: the last pivots sit at ~lambda).  The fp32 rounding of the Gram
// (~3e-7 |G|) reaches the solution amplified by cond(A); a row whose kappa exceeds the limit - or whose fp32
// factorisation broke down - is handed to the fp64 kernel instead of being finished here.  14 + 3 vector
// instructions per row.  Returns true when the row is to be redone (nothing of it may be stored then).
// ---------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ bool row_needs_f64(const Chol<KB>& S, const als_row_solve_params& P, int row, bool spd,
                                              float mean_eig, float short_row_bound, int lane) {
    float dmx = 0.f, dmn = 0.f;          // max of 1 / L_ii and of L_ii over the lane's rows
#pragma unroll
    for (int rr = 0; rr < KCfg<KB>::NR; ++rr) {
        const bool in = lane + 64 * rr < KCfg<KB>::KP;
        const float d = in ? S.di[rr] : 0.f;
        dmx = fmaxf(dmx, d);
        dmn = fmaxf(dmn, in ? __builtin_amdgcn_rcpf(d) : 0.f);
    }
    dmx = wave_max_nonneg(fmaxf(dmx, 0.f));
    dmn = wave_max_nonneg(fmaxf(dmn, 0.f));
    const float r = dmx * dmn;
    // two lower bounds of cond_2(A): the pivot ratio, and (mean of the non-trivial eigenvalues) / (smallest pivot) -
    // lambda_max >= trace(G) / rank(G) + lambda.  The second one sees what the first misses: a row with a handful
    // of ratings has a few eigenvalues of the size of |f|^2 and k - n equal to lambda, while its largest PIVOT is
    // only max_c f_c^2 + lambda (at lambda = 1e-4: estimate 100 against a true 6400).
    // Both use the smallest PIVOT for lambda_min, which it can exceed by a factor of up to k.  For rows of fewer than
    // 4 k ratings - where the Gram's own smallest eigenvalues are zero or unreliable - lambda_min is therefore taken
    // as the regulariser itself (short_row_bound = mean eigenvalue / lambda, 0 for longer rows): pessimistic by at
    // most (lambda + sigma_min(G)) / lambda, and what makes the lambda = 1e-4 rows of k ... 4 k ratings go to fp64.
    const float kappa = spd ? fmaxf(fmaxf(r * r, mean_eig * dmx * dmx), short_row_bound) : __builtin_inff();
    if (P.cond_out && lane == 0) P.cond_out[row] = kappa;
    const bool redo = !(kappa <= P.cond_limit);
    if (redo && lane == 0) P.redo_rows[atomicAdd(P.redo_count, 1)] = row;
    return redo;
}

// trace of the accumulated Gram (before the regulariser goes on): its diagonal sits in the lanes q == c >> 2,
// register c & 3 of every diagonal block
template <int KB>
__device__ __forceinline__ float gram_trace(const RowAcc<KB>& A, int lane) {
    const int c = lane & 15, q = lane >> 4;
    float t = 0.f;
#pragma unroll
    for (int J = 0; J < KB; ++J) {
        const f32x4 a = A.acc[blk_idx(J, J)];
        const float lo = (c & 1) ? a[1] : a[0], hi = (c & 1) ? a[3] : a[2];
        t += (c & 2) ? hi : lo;
    }
    return wave_sum((q == (c >> 2)) ? t : 0.f);
}

// ---------------------------------------------------------------------------
// tail: reduce, regularise, factorise, solve / emit factor
// ---------------------------------------------------------------------------
template <int KB>
__device__ __forceinline__ void finish_row(RowAcc<KB>& A, const als_row_solve_params& P, int row,
                                           float* __restrict__ Ls, int lane) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    const int c = lane & 15, q = lane >> 4;
    const int64_t r64 = row;

    if (P.gram_mode == ALS_GRAM_F16X2) {        // the accumulators hold S^2 G (exact power of two)
        const float un = P.F_scale[1];
#pragma unroll
        for (int a = 0; a < C::NACC; ++a) A.acc[a] *= un;
    }

    // cross-lane reductions: rhs / colsum over the four q groups, sumr over the wave
#pragma unroll
    for (int b = 0; b < KB; ++b) {
        A.rhs[b] += __shfl_xor(A.rhs[b], 16, 64); A.rhs[b] += __shfl_xor(A.rhs[b], 32, 64);
        A.cs[b] += __shfl_xor(A.cs[b], 16, 64);   A.cs[b] += __shfl_xor(A.cs[b], 32, 64);
    }
    const float sumr = wave_sum(A.sumr);
    const float sumr2 = wave_sum(A.sumr2);

    // perm position i = lane + 64 rr is block (q + 4 rr), position c: the lane already holds it
    Chol<KB> S;
    float csrow[NR];
    int colrow[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        float bsel = 0.f, csel = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            constexpr int dummy = 0; (void)dummy;
            const int b = (4 * rr + e < KB) ? 4 * rr + e : KB - 1;
            if (4 * rr + e < KB) {
                bsel = (q == e) ? A.rhs[b] : bsel;
                csel = (q == e) ? A.cs[b] : csel;
            }
        }
        S.b[rr] = bsel; S.di[rr] = 0.f; S.y[rr] = 0.f;
        csrow[rr] = csel;
        colrow[rr] = perm_to_col<KB>(min(lane + 64 * rr, KP - 1));
    }

    if (P.gram_out) {       // F^T F without lambda, perm space, lower 16x16 blocks
        float* G = P.gram_out + r64 * KP * KP;
#pragma unroll
        for (int I = 0; I < KB; ++I)
#pragma unroll
            for (int K = 0; K <= I; ++K)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    G[(16 * I + 4 * q + r) * KP + 16 * K + c] = A.acc[blk_idx(I, K)][r];
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        if (i < KP) {
            if (P.rhs_out) P.rhs_out[r64 * KP + i] = S.b[rr];
            if (P.colsum_out) P.colsum_out[r64 * KP + i] = csrow[rr];
        }
    }
    if (P.sumr_out && lane == 0) P.sumr_out[row] = sumr;
    if (P.sumr2_out && lane == 0) P.sumr2_out[row] = sumr2;

    // regulariser on the diagonal (C/D layout: diagonal where 4q + r == c); padded columns get 1
    const float lam = (P.lambda_row ? P.lambda_row[row] : P.lambda_scalar) + ALS_EPS
                    + (P.diag_extra ? P.diag_extra[row] : 0.f);
    float mean_eig = 0.f, short_row_bound = 0.f;       // (condition estimate of solve_dtype "auto" only)
    if (P.cond_limit > 0.f) {
        const int64_t nrat = P.indptr[row + 1] - P.indptr[row];
        mean_eig = gram_trace<KB>(A, lane) / fmaxf((float)min((int64_t)P.k, nrat), 1.f) + lam;
        short_row_bound = (nrat < 4 * (int64_t)P.k) ? mean_eig / lam : 0.f;
    }
#pragma unroll
    for (int J = 0; J < KB; ++J) {
        const float dv = (perm_to_col<KB>(16 * J + c) < P.k) ? lam : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) A.acc[blk_idx(J, J)][r] += (4 * q + r == c) ? dv : 0.f;
    }

    if (P.factor_out) {
        chol_panels<KB, 0, false>(A, S, Ls, lane);
        const bool spd = chol_spd<KB>(S, lane);
        if (P.cond_limit > 0.f) {
            if (row_needs_f64<KB>(S, P, row, spd, mean_eig, short_row_bound, lane)) return;
        } else if (!spd && lane == 0) atomicMax(P.status, row + 1);
        // symmetric completion of L with 1/L_ii on the diagonal, perm space:
        // M[p][i] = L[i][p] (p < i), L[p][i] (p > i)
        float* M = P.factor_out + r64 * KP * KP;
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
            const int i = lane + 64 * rr;
            const int ic = min(i, KP - 1);
            const int colb = C::lcol_off_rt(ic >> 4) - 16 * (ic >> 4) * 16 + (c & 3);
#pragma unroll 8
            for (int p = 0; p < KP; ++p) {      // (unrolled: the LDS reads of 8 rows in flight - one wave per SIMD at k > 96)
                const int Jp = p >> 4;
                const int rowv = max(ic, 16 * Jp);                                   // row i inside block column Jp
                const float lrow = Ls[C::lcol_off_rt(Jp) + (rowv - 16 * Jp) * 16 +
                                      ((((p >> 2) & 3) ^ ((rowv >> 2) & 3)) << 2) + (p & 3)];
                const int rsafe = max(p, ic & ~15);
                const float lcolv = Ls[colb + rsafe * 16 + (((c >> 2) ^ ((rsafe >> 2) & 3)) << 2)];
                if (i < KP) __builtin_nontemporal_store((p == i) ? S.di[rr] : (p < i ? lrow : lcolv), M + p * KP + i);
            }
        }
        return;
    }

    float rhs0[NR];                         // F^T r before any extra right-hand side (statistics)
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) rhs0[rr] = S.b[rr];
    if (P.rhs_extra) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
            if (lane + 64 * rr < KP) S.b[rr] += P.rhs_extra[r64 * P.ld + colrow[rr]];
    }
    if (!(P.reserved0 & 2)) {
        chol_panels<KB, 0, true>(A, S, Ls, lane);
        const bool spd = chol_spd<KB>(S, lane);
        if (P.cond_limit > 0.f) {
            if (row_needs_f64<KB>(S, P, row, spd, mean_eig, short_row_bound, lane)) return;
        } else if (!spd && lane == 0) atomicMax(P.status, row + 1);
    }
    float x[NR];
    if (!(P.reserved0 & 4)) backward_solve<KB>(Ls, S, x, lane);
    else {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) x[rr] = S.y[rr];
    }

    float dot = 0.f, xr = 0.f, yy = 0.f, xx = 0.f;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (lane + 64 * rr < KP) {
            dot = fmaf(csrow[rr], x[rr], dot);
            xr = fmaf(rhs0[rr], x[rr], xr);
            yy = fmaf(S.y[rr], S.y[rr], yy);
            xx = fmaf(x[rr], x[rr], xx);
        }
    }
    dot = wave_sum(dot);
    const float nnz = (float)(P.indptr[row + 1] - P.indptr[row]);
    const float lb = P.lambda_bias_row ? P.lambda_bias_row[row] : P.lambda_bias_scalar;
    const float bnew = (sumr - dot) / (nnz + lb + ALS_EPS);
    const float bold = P.bias_self[row];            // read before the store: bias_out may alias bias_self
    float st1 = 0.f, st2 = 0.f;
    if (P.stat_out) {
        // residuals of this row with the new x and bias, in closed form (DESIGN.md "Statistics"):
        //   sum d   = sum rho - nnz b - (F^T 1).x
        //   sum d^2 = sum (rho - b)^2 - 2 x.F^T(rho - b) + x^T G x,   x^T G x = |y|^2 - lambda |x|^2
        xr = wave_sum(xr); yy = wave_sum(yy); xx = wave_sum(xx);
        // rhs0 was formed with the old bias (bold).  The combination is a difference of large sums: in double, but
        // its inputs are fp32 sums (1e-7 relative each) - once the residuals are so small that less than three of
        // those seven digits survive the cancellation (train RMSE below ~3 % of the ratings' spread: the over-fitted
        // small-lambda corner) the row goes to the fp64 kernel as well when the call allows it (cond_limit > 0)
        const double bn = bnew, dt = dot;
        const double s1 = (double)sumr - (double)nnz * bn;
        const double s2 = (double)sumr2 - 2.0 * bn * (double)sumr + (double)nnz * bn * bn;
        const double cross = (double)xr + ((double)bold - bn) * dt;
        const double quad = (double)yy - (double)lam * (double)xx;
        const double sd2 = s2 - 2.0 * cross + quad;
        if (P.cond_limit > 0.f && !(sd2 >= 1e-3 * s2)) {
            if (lane == 0) P.redo_rows[atomicAdd(P.redo_count, 1)] = row;
            return;
        }
        st1 = (float)(s1 - dt);
        st2 = (float)sd2;
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
        if (lane + 64 * rr < KP) P.X_out[r64 * P.ld + colrow[rr]] = x[rr];
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) {
        P.bias_out[row] = bnew;
        if (P.stat_out) { P.stat_out[2 * r64] = st1; P.stat_out[2 * r64 + 1] = st2; }
    }
}

template <int KB, int MODE>
__global__ __launch_bounds__(64 * KCfg<KB>::WPW, KCfg<KB>::MINW)
void k_row_tasks(const als_row_solve_params P) {
    using C = KCfg<KB>;
    __shared__ __attribute__((aligned(16))) float lds_all[C::WPW * C::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // everything that describes the task is wave-uniform: keep it in SGPRs (scalar loads,
    // scalar branches in the chunk loop)
    const int wave_u = wave;
    const int64_t tid = (int64_t)blockIdx.x * C::WPW + wave_u;
    if (tid >= P.ntasks) return;
    const als_task t = P.tasks[tid];
    const int row = t.row;
    const int seg = t.seg;
    const int slot = t.slot;
    const int64_t rbeg = P.indptr[row], rend = P.indptr[row + 1];
    const int64_t beg = rbeg + (int64_t)seg * ALS_SPLIT_CHUNK;
    const int len = (int)min((int64_t)ALS_SPLIT_CHUNK, rend - beg);
    const float mu = (float)(*P.mu);
    const float bself = P.bias_self[row];

    const float S = (MODE != 0) ? P.F_scale[0] : 1.f;

    RowAcc<KB> A;
    A.zero();
    if (!(P.reserved0 & 1))        // reserved0: ablation flags for profiling builds, 0 in production
        gram_accumulate<KB, MODE>(A, P.indices + beg, P.vals + beg, len, P.F, P.ld, P.F_zero_row,
                                  P.bias_other, mu, bself, lane, lds_all + wave * C::LDS_FLOATS, S,
                                  (const uint32_t*)P.F_planes);
    if (slot >= 0) {
        store_partial<KB>(A, (float*)P.workspace + (size_t)slot * C::SLOT_ITEMS * 64, lane);
        return;
    }
    finish_row<KB>(A, P, row, lds_all + wave_u * C::LDS_FLOATS, lane);
}

// ---------------------------------------------------------------------------
// Dual form for short rows (at most 64 ratings, fewer than k), solve mode only.
//
//   (F^T F + l I) x = F^T r   <=>   x = F^T w,  (F F^T + l I) w = r        (l = lambda + 1e-10)
//
// The n x n system (n <= 64) is built on the fp16 matrix cores exactly like the Gram - ratings on both
// MFMA axes, the k factor columns as the contraction index, same 2-way split - and solved by the
// k = 64 machinery above (blocked Cholesky of a 64 x 64 matrix held in MFMA accumulators; rows past n are
// identity).  Everything the row needs follows from w in closed form, since F x = F F^T w = r - l w:
//   bias_new = (n b_old + l sum w) / (n + lambda_b + 1e-10)
//   d_t      = b_old - bias_new + l w_t                      (residual with the new x and bias)
// One wave per row, 4 per workgroup, 3 waves per SIMD (the primal k = 128 kernel needs 500 registers and
// k^3/3 Cholesky flops per row: one wave per SIMD).  The caller decides which rows qualify (`ndual_tail`):
// all rows of at most 64 ratings for k > 64; for k <= 64 the rows whose 16-rating blocks are fewer than k/16,
// i.e. a strictly smaller system (k = 64: n <= 48).
// ---------------------------------------------------------------------------
// NB = 16-rating blocks of the row (n <= 16 NB): the system is 16 NB x 16 NB and runs on the k = 16 NB
// machinery; rating t of the row lives in lane t & 63, slot t >> 6 (two slots for NB > 4).
template <int KB, int NB>
__device__ __forceinline__ void row_dual(const als_row_solve_params& P, int row, int64_t beg, int len,
                                         float* __restrict__ Ls, int lane) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, NSLAB = (KP + 31) / 32;
    constexpr int NS = KCfg<NB>::NR;                 // rating slots per lane
    const int c = lane & 15, q = lane >> 4;
    const float bold = P.bias_self[row];
    const float mu = (float)*P.mu;
    bool ok[NS];
    int idx_l[NS];
    float r_l[NS];
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
        const int t = lane + 64 * ss;
        ok[ss] = t < len;
        idx_l[ss] = ok[ss] ? P.indices[beg + t] : P.F_zero_row;
        r_l[ss] = ok[ss] ? (P.vals[beg + t] - mu - P.bias_other[idx_l[ss]]) - bold : 0.f;
    }
    const int nblk = (len + 15) >> 4;               // 16-rating blocks in use (wave-uniform)
    int off[NB];
#pragma unroll
    for (int I = 0; I < NB; ++I)                     // lanes past len: the zero row
        off[I] = bperm_i(idx_l[(16 * I) >> 6], (16 * I + c) & 63) * P.ld;

    // K = F F^T: rating block I on the M axis, J <= I on the N axis, 32 factor columns per MFMA
    const float Sc = P.F_scale[0];
    RowAcc<NB> A;
    A.zero();
    for (int s = 0; s < NSLAB; ++s) {
        const bool in = 32 * s + 8 * q < KP;         // a lane's 8 columns are all inside or all outside
        i32x4 H[NB], L[NB];
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            float f[8];
            if (I < nblk && in) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(P.F + (uint32_t)off[I] + 32 * s + 8 * q);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(P.F + (uint32_t)off[I] + 32 * s + 8 * q + 4);
                f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                int hw, lw;
                split2(f[e], f[e + 1], Sc, hw, lw);
                H[I][e >> 1] = hw; L[I][e >> 1] = lw;
            }
        }
#pragma unroll
        for (int bi = 0; bi < NB; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) {
                if (bi >= nblk) continue;
                f32x4 acc = A.acc[blk_idx(bi, bj)];
                const h16x8 hi = __builtin_bit_cast(h16x8, H[bi]), hj = __builtin_bit_cast(h16x8, H[bj]);
                const h16x8 li = __builtin_bit_cast(h16x8, L[bi]), lj = __builtin_bit_cast(h16x8, L[bj]);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(li, hj, acc, 0, 0, 0);     // smallest terms first
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, lj, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, hj, acc, 0, 0, 0);
                A.acc[blk_idx(bi, bj)] = acc;
            }
    }
    mfma_results_settle();
    {
        const float un = P.F_scale[1];               // S^2 K -> K
#pragma unroll
        for (int a = 0; a < KCfg<NB>::NACC; ++a) A.acc[a] *= un;
    }
    // + l I on the rows in use, identity on the padding rows (their rows / columns of K are zero)
    const float lam = (P.lambda_row ? P.lambda_row[row] : P.lambda_scalar) + ALS_EPS;
    const float dual_mean_eig = (P.cond_limit > 0.f) ? gram_trace<NB>(A, lane) / (float)max(min(len, P.k), 1) + lam : 0.f;
#pragma unroll
    for (int J = 0; J < NB; ++J) {
        const float dv = (16 * J + c < len) ? lam : 1.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) A.acc[blk_idx(J, J)][r] += (4 * q + r == c) ? dv : 0.f;
    }
    Chol<NB> S;
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) { S.b[ss] = r_l[ss]; S.di[ss] = 0.f; S.y[ss] = 0.f; }
    chol_panels<NB, 0, true>(A, S, Ls, lane);
    {
        // (F F^T + l I and F^T F + l I share their spectrum up to the multiplicity of l: the same condition estimate)
        const bool spd = chol_spd<NB>(S, lane);
        if (P.cond_limit > 0.f) {
            if (row_needs_f64<NB>(S, P, row, spd, dual_mean_eig, dual_mean_eig / lam, lane)) return;
        } else if (!spd && lane == 0) atomicMax(P.status, row + 1);
    }
    float w[NS];
    backward_solve<NB>(Ls, S, w, lane);               // w_t in lane t & 63, slot t >> 6 (0 past len)

    // x = F^T w: lane (+64 rr) owns factor column lane + 64 rr (storage order); 8 rating rows in flight
    float x[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) x[rr] = 0.f;
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) {
        const int lim = min(len - 64 * ss, 64);
        for (int i0 = 0; i0 < lim; i0 += 8) {
            float fv[8][NR], wi[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = min(i0 + u, 63);
                wi[u] = (i0 + u < lim) ? readlane_f(w[ss], i) : 0.f;
                const int o = __builtin_amdgcn_readlane(idx_l[ss], i) * P.ld;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) fv[u][rr] = P.F[(uint32_t)o + min(lane + 64 * rr, KP - 1)];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) x[rr] = fmaf(fv[u][rr], wi[u], x[rr]);
        }
    }
    const int64_t r64 = row;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
        if (lane + 64 * rr < KP) P.X_out[r64 * P.ld + lane + 64 * rr] = x[rr];
    const float nnz = (float)len;
    const float lb = P.lambda_bias_row ? P.lambda_bias_row[row] : P.lambda_bias_scalar;
    float wsum = 0.f;
#pragma unroll
    for (int ss = 0; ss < NS; ++ss) wsum += ok[ss] ? w[ss] : 0.f;
    const float sw = wave_sum(wsum);
    const float bnew = (nnz * bold + lam * sw) / (nnz + lb + ALS_EPS);
    if (lane == 0) P.bias_out[row] = bnew;
    if (P.stat_out) {
        float e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            const float e = ok[ss] ? (bold - bnew) + lam * w[ss] : 0.f;
            e1 += e;
            e2 = fmaf(e, e, e2);
        }
        const float s1 = wave_sum(e1), s2 = wave_sum(e2);
        if (lane == 0) { P.stat_out[2 * r64] = s1; P.stat_out[2 * r64 + 1] = s2; }
    }
}

template <int KB>
__global__ __launch_bounds__(256, 3)
void k_row_dual(const als_row_solve_params P, int64_t task0, int64_t ntail) {
    using C4 = KCfg<4>;
    __shared__ __attribute__((aligned(16))) float lds_all[4 * C4::LDS_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * 4 + wave;
    if (tid >= ntail) return;
    float* Ls = lds_all + wave * C4::LDS_FLOATS;
    const int row = P.tasks[task0 + tid].row;
    const int64_t beg = P.indptr[row];
    const int len = (int)(P.indptr[row + 1] - beg);
    if (len > 64 || len < 1) {                       // not a task for this kernel: host bug, fail loudly
        if (lane == 0) atomicMax(P.status, row + 1);
        return;
    }
    // the tail is sorted by length, so neighbouring waves take the same branch
    if (len <= 16) row_dual<KB, 1>(P, row, beg, len, Ls, lane);
    else if (len <= 32) row_dual<KB, 2>(P, row, beg, len, Ls, lane);
    else if (len <= 48) row_dual<KB, 3>(P, row, beg, len, Ls, lane);
    else row_dual<KB, 4>(P, row, beg, len, Ls, lane);
}

// rows of 65 ... 96 ratings of models with k > 96: an 80- or 96-size system on the k = 80 / 96 machinery
// (two matrix rows per lane, two waves per workgroup and per SIMD)
template <int KB>
__global__ __launch_bounds__(128, 2)
void k_row_dual_mid(const als_row_solve_params P, int64_t task0, int64_t nmid) {
    using C6 = KCfg<6>;
    __shared__ __attribute__((aligned(16))) float lds_all[2 * C6::LDS_FLOATS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * 2 + wave;
    if (tid >= nmid) return;
    float* Ls = lds_all + wave * C6::LDS_FLOATS;
    const int row = P.tasks[task0 + tid].row;
    const int64_t beg = P.indptr[row];
    const int len = (int)(P.indptr[row + 1] - beg);
    if (len > 96 || len <= 64) {                     // not a task for this kernel: host bug, fail loudly
        if (lane == 0) atomicMax(P.status, row + 1);
        return;
    }
    if (len <= 80) row_dual<KB, 5>(P, row, beg, len, Ls, lane);
    else row_dual<KB, 6>(P, row, beg, len, Ls, lane);
}

template <int KB>
__global__ __launch_bounds__(64 * KCfg<KB>::WPW, KCfg<KB>::MINW)
void k_row_long(const als_row_solve_params P) {
    using C = KCfg<KB>;
    __shared__ __attribute__((aligned(16))) float lds_all[C::WPW * C::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tid = (int64_t)blockIdx.x * C::WPW + wave;
    if (tid >= P.nlong) return;
    const als_long_row lr = P.long_rows[tid];
    RowAcc<KB> A;
    A.zero();
    // k_sum_slots has already folded the row's nslots partials into its first slot
    add_partial<KB>(A, (const float*)P.workspace + (size_t)lr.slot0 * C::SLOT_ITEMS * 64, lane);
    finish_row<KB>(A, P, lr.row, lds_all + wave * C::LDS_FLOATS, lane);
}

// Partial normal equations of a split row, summed into the row's first slot: one thread per element, the
// slots in ascending order (the order a single wave adding slot after slot would use - bitwise the same
// sums), all elements of all long rows in parallel.  A single wave walking 40 slots of 41 KB (k = 128) with
// a handful of loads in flight took milliseconds.
template <int KB>
__global__ __launch_bounds__(256)
void k_sum_slots(const als_long_row* __restrict__ long_rows, float* __restrict__ workspace) {
    constexpr int N = KCfg<KB>::SLOT_ITEMS * 64;
    const als_long_row lr = long_rows[blockIdx.x];
    const int e = blockIdx.y * 256 + threadIdx.x;
    if (e >= N || lr.nslots < 2) return;
    float* w0 = workspace + (size_t)lr.slot0 * N + e;
    float acc = 0.f + w0[0];
    for (int s = 1; s < lr.nslots; ++s) acc += w0[(size_t)s * N];
    w0[0] = acc;
}

// Scale of the f16x2 operand split (split2): S = 2^j with S max|F| in [2^14, 2^15), clamped to 2^-60 ... 2^60, and
// 1 / S^2.  scale[2] (running maximum of the |F| bit patterns) and scale[3] (ticket counter) are zero on entry and
// are left zero by the workgroup that arrives last; a NaN / inf in F has the largest pattern and ends as a NaN Gram,
// which the factorisation reports as "not positive definite" - as without the scaling.
__global__ __launch_bounds__(256)
void k_factor_scale(const float* __restrict__ F, int64_t n4, float* __restrict__ scale, int32_t* __restrict__ redo_count) {
    __shared__ uint32_t wmax[4];
    const f32x4* __restrict__ F4 = reinterpret_cast<const f32x4*>(F);
    uint32_t m = 0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = (i + u * stride < n4) ? F4[i + u * stride] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = max(m, (uint32_t)__float_as_int(v[u][e]) & 0x7FFFFFFFu);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t* w = reinterpret_cast<uint32_t*>(scale);
        atomicMax(&w[2], max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
        __threadfence();
        if (atomicAdd(&w[3], 1u) == gridDim.x - 1) {
            __threadfence();
            const uint32_t mx = atomicMax(&w[2], 0u);
            const int e = (int)(mx >> 23);
            const int se = min(max(268 - e, 67), 187);
            scale[0] = __int_as_float(se << 23);
            scale[1] = __int_as_float((381 - 2 * se) << 23);
            atomicExch(&w[2], 0u);
            atomicExch(&w[3], 0u);
            if (redo_count) *redo_count = 0;        // (the call's list of rows for the fp64 kernel starts empty)
        }
    }
}

int launch_factor_scale(const float* F, int64_t nfloats, float* scale, int32_t* redo_count, hipStream_t st) {
    const int64_t n4 = nfloats / 4;
    const unsigned grid = (unsigned)min((int64_t)512, max((int64_t)1, (n4 + 1023) / 1024));
    hipLaunchKernelGGL(k_factor_scale, dim3(grid), dim3(256), 0, st, F, n4, scale, redo_count);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

__global__ void k_reset_word(int32_t* w) { *w = 0; }

// F [nrows][ld] fp32 -> planes [nrows][ld] words (ld = 16 KB, KB = 4 or 8): per row and 16-lane column group c the
// piece (h pairs)[KB / 2] (l pairs)[KB / 2] of the columns KB c ... KB c + KB - 1, h = fp16(x S), l = fp16(x S - h)
// (split2: the very terms the in-kernel split of ALS_GRAM_F16X2 produces).  One thread per piece.
template <int KB>
__global__ __launch_bounds__(256)
void k_split_planes(const float* __restrict__ F, int64_t npieces, const float* __restrict__ scale,
                    uint32_t* __restrict__ planes) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const float S = scale[0];
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < npieces; e += (int64_t)gridDim.x * 256) {
        uint32_t h[KB / 2], l[KB / 2];
#pragma unroll
        for (int v4 = 0; v4 < KB / 4; ++v4) {
            const f32x4 x = reinterpret_cast<const f32x4*>(F)[e * (KB / 4) + v4];
            int h01, l01, h23, l23;
            split2(x[0], x[1], S, h01, l01);
            split2(x[2], x[3], S, h23, l23);
            h[2 * v4] = (uint32_t)h01; h[2 * v4 + 1] = (uint32_t)h23;
            l[2 * v4] = (uint32_t)l01; l[2 * v4 + 1] = (uint32_t)l23;
        }
        if constexpr (KB == 4) {
            reinterpret_cast<u32x4*>(planes)[e] = u32x4{h[0], h[1], l[0], l[1]};
        } else {
            reinterpret_cast<u32x4*>(planes)[2 * e] = u32x4{h[0], h[1], h[2], h[3]};
            reinterpret_cast<u32x4*>(planes)[2 * e + 1] = u32x4{l[0], l[1], l[2], l[3]};
        }
    }
}

template <int KB>
int launch_row_solve(const als_row_solve_params* p, hipStream_t st) {
    using C = KCfg<KB>;
    // Short rows at the tail of the task list go to the dual-form kernels when the call is a plain solve
    // (see k_row_dual): `ndual_tail` whole rows of at most 64 ratings, preceded - k > 96 only, where it is the
    // smaller system - by `ndual_mid` whole rows of 65 ... 96 ratings.  Any by-product output, extra
    // right-hand side or diagonal, the f32 Gram mode or an ablation flag keeps every row primal.
    const bool plain = p->gram_mode == ALS_GRAM_F16X2 && p->reserved0 == 0 && p->X_out && p->bias_out &&
                       !p->gram_out && !p->factor_out && !p->rhs_out && !p->colsum_out && !p->sumr_out &&
                       !p->sumr2_out && !p->rhs_extra && !p->diag_extra;
    int64_t ntail = 0, nmid = 0;
    if (plain && p->ndual_tail >= 0 && p->ndual_mid >= 0 && (int64_t)p->ndual_tail + p->ndual_mid <= p->ntasks) {
        ntail = p->ndual_tail;
        if constexpr (KB >= 7) nmid = p->ndual_mid;        // (k <= 96: those rows stay with the primal tasks)
    }
    const int64_t nprimal = p->ntasks - ntail - nmid;
    if (nprimal > 0) {
        als_row_solve_params q = *p;
        q.ntasks = nprimal;
        const unsigned grid = (unsigned)((nprimal + C::WPW - 1) / C::WPW);
        bool planes = false;
        if constexpr (KB == 4 || KB == 8)
            planes = p->gram_mode == ALS_GRAM_F16X2 && p->F_planes != nullptr && p->reserved0 == 0;
        if (planes) {
            if constexpr (KB == 4 || KB == 8) {
                const int64_t npieces = ((int64_t)p->F_zero_row + 1) * 16;
                const unsigned sgrid = (unsigned)min((int64_t)4096, (npieces + 255) / 256);
                hipLaunchKernelGGL(k_split_planes<KB>, dim3(sgrid), dim3(256), 0, st, p->F, npieces, p->F_scale,
                                   (uint32_t*)p->F_planes);
                hipLaunchKernelGGL((k_row_tasks<KB, 2>), dim3(grid), dim3(64 * C::WPW), 0, st, q);
            }
        } else if (p->gram_mode == ALS_GRAM_F16X2)
            hipLaunchKernelGGL((k_row_tasks<KB, 1>), dim3(grid), dim3(64 * C::WPW), 0, st, q);
        else
            hipLaunchKernelGGL((k_row_tasks<KB, 0>), dim3(grid), dim3(64 * C::WPW), 0, st, q);
    }
    {
        if (ntail > 0)
            hipLaunchKernelGGL(k_row_dual<KB>, dim3((unsigned)((ntail + 3) / 4)), dim3(256), 0, st, *p, nprimal + nmid, ntail);
    }
    if constexpr (KB >= 7) {
        if (nmid > 0)
            hipLaunchKernelGGL(k_row_dual_mid<KB>, dim3((unsigned)((nmid + 1) / 2)), dim3(128), 0, st, *p, nprimal, nmid);
    }
    if (p->nlong > 0) {
        hipLaunchKernelGGL(k_sum_slots<KB>, dim3((unsigned)p->nlong, (C::SLOT_ITEMS * 64 + 255) / 256), dim3(256), 0, st,
                           p->long_rows, (float*)p->workspace);
        const unsigned grid = (unsigned)((p->nlong + C::WPW - 1) / C::WPW);
        hipLaunchKernelGGL(k_row_long<KB>, dim3(grid), dim3(64 * C::WPW), 0, st, *p);
    }
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

int als_row_solve_f64_dispatch(const als_row_solve_params* p, hipStream_t st);     // row_solve_f64.hip
int als_row_redo_f64_dispatch(const als_row_solve_params* p, hipStream_t st);

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
    return (int64_t)(KB * (KB + 1) / 2 * 4 + 2 * KB + 2) * 64 * sizeof(float);
}

extern "C" int als_factor_scale(const float* F, int64_t nfloats, float* scale, void* stream) {
    if (!F || !scale || nfloats < 0 || (nfloats & 3) || ((uintptr_t)F & 15)) return ALS_E_BADARG;
    return launch_factor_scale(F, nfloats, scale, nullptr, (hipStream_t)stream);
}

extern "C" int als_row_solve(const als_row_solve_params* p, void* stream) {
    if (!p) return ALS_E_BADARG;
    const int ld = als_padded_k(p->k);
    if (ld < 0) return ALS_E_BADK;
    if (p->ld != ld || !p->indptr || !p->indices || !p->vals || !p->F || !p->bias_self ||
        !p->bias_other || !p->mu || !p->status || p->ntasks < 0 || p->nlong < 0 || p->F_zero_row < 0 ||
        (p->gram_mode != ALS_GRAM_F32 && p->gram_mode != ALS_GRAM_F16X2 && p->gram_mode != ALS_GRAM_F64) ||
        (int64_t)p->F_zero_row * ld >= ((int64_t)1 << 31))
        return ALS_E_BADARG;
    if (p->ntasks > 0 && !p->tasks) return ALS_E_BADARG;
    if (p->nlong > 0 && (!p->long_rows || !p->workspace)) return ALS_E_BADARG;
    if (p->factor_out) {
        if (!p->rhs_out || !p->colsum_out || !p->sumr_out) return ALS_E_BADARG;
    } else if (!p->X_out || !p->bias_out) {
        return ALS_E_BADARG;
    }
    hipStream_t st = (hipStream_t)stream;
    if (p->gram_mode == ALS_GRAM_F64) return als_row_solve_f64_dispatch(p, st);
    const bool redo = p->cond_limit > 0.f;
    if (redo && (!p->redo_count || !p->redo_rows)) return ALS_E_BADARG;
    bool counter_reset = false;
    if (p->gram_mode == ALS_GRAM_F16X2) {
        if (!p->F_scale) return ALS_E_BADARG;
        if (!p->F_scale_ready) {
            const int rc = launch_factor_scale(p->F, ((int64_t)p->F_zero_row + 1) * ld, p->F_scale,
                                               redo ? p->redo_count : nullptr, st);
            if (rc != 0) return rc;
            counter_reset = true;
        }
    }
    if (redo && !counter_reset) hipLaunchKernelGGL(k_reset_word, dim3(1), dim3(1), 0, st, p->redo_count);
    int rc = ALS_E_BADK;
#ifdef ALS_KB_ONLY      // development builds (profiles/ab_builds.sh): one model width only, a tenth of the compile time
    if (ld / 16 == ALS_KB_ONLY) rc = launch_row_solve<ALS_KB_ONLY>(p, st);
#else
    switch (ld / 16) {
        case 1: rc = launch_row_solve<1>(p, st); break;
        case 2: rc = launch_row_solve<2>(p, st); break;
        case 3: rc = launch_row_solve<3>(p, st); break;
        case 4: rc = launch_row_solve<4>(p, st); break;
        case 5: rc = launch_row_solve<5>(p, st); break;
        case 6: rc = launch_row_solve<6>(p, st); break;
        case 7: rc = launch_row_solve<7>(p, st); break;
        case 8: rc = launch_row_solve<8>(p, st); break;
        case 9: rc = launch_row_solve<9>(p, st); break;
        case 10: rc = launch_row_solve<10>(p, st); break;
    }
#endif
    // rows the fp32 kernels flagged (condition estimate above the limit, or a broken-down factorisation): fp64
    if (rc == 0 && redo) rc = als_row_redo_f64_dispatch(p, st);
    return rc;
}
