// K1 for k = 112 / 128 (KB = 7, 8) with TWO wavefronts per row.
//
// Same contract as k_row_tasks (row_solve.hip; reference scripts/als.py:414-433, 436-466, scripts/helpers.py:5-20).
// Why a second kernel: at k = 128 the one-wave kernel holds 36 accumulator blocks + the split operands of a
// 32-rating group in ~500 registers (one wave per SIMD) and its L image takes 36 KB of LDS (four rows per CU
// whatever the register count).  A wave alone issues one vector instruction every ~4-8 cycles and exposes every
// gather latency, so the CU idles most of the time.  Two waves per row is the only way to a second wave per SIMD
// at four rows per CU.
//
// Workgroup = 2 waves (A, B) = 1 row at a time; persistent grid (4 workgroups per CU), tasks dealt round-robin
// from the longest-first task list.
//
//   Gram.   A owns factor-column blocks 0..3, B owns 4..KB-1: each wave gathers, sums (F^T r, F^T 1) and splits
//           (exact 3-way bf16 split) only its own columns - half the bytes and half the VALU work of the row.
//           The Gram's lower 16x16 blocks are dealt 18 / 18 (KB = 8; 14 / 14 at KB = 7): A accumulates the blocks
//           among its own columns and the cross blocks of B's first XA block rows, B the blocks among its own
//           columns and the remaining cross blocks.  Cross blocks need the other wave's split operands: A
//           publishes its 4 blocks, B its first XA, per 32-rating group through LDS (the idle L image, double
//           buffered, ONE workgroup barrier per group; the own-column MFMAs are issued before the barrier).
//           Long rows: every 512 ratings the accumulators are added into fp32 totals (see flush_acc in
//           row_solve.hip for why); here the totals live in a per-workgroup scratch in global memory (the LDS
//           holds the exchange buffers), which stays cache resident because the same workgroup reuses it.
//   Cholesky.  Both waves dump their blocks into the LDS image (the layout of KCfg<KB>: block columns, rows
//           x 16, swizzled) and the trailing matrix STAYS there.  Per 16-column panel: every lane takes one row of
//           the block column into registers (lanes 0..15 of BOTH waves take the 16 rows of the diagonal block -
//           the pivots and multipliers are computed redundantly, bitwise equal, so no per-pivot exchange is
//           needed - lanes 16..63 take 48 of the rows below it each), the panel is eliminated with the same DPP
//           row_newbcast scheme as the one-wave kernel, written back, and after a barrier the rank-16 trailing
//           update of the remaining blocks runs on the f32 matrix cores, blocks dealt alternately to the two
//           waves (C read from / written to the image as one b128 per lane by computing the transposed product).
//           The forward substitution rides along in the panel.
//   Solve.  Wave A runs the transposed solve from the image (backward_solve of the one-wave kernel, two unknowns
//           per lane), the bias update and the closed-form statistics; in factor-only mode both waves write half
//           of the symmetric completion each.
#include "als_device.hpp"
#include "als_hip.h"
#include "row_common.hpp"

namespace {

// Phase timing for profiling builds (-DPAIR_PROFILE, profiles/pair_phase_profile.py): per wave role, shader-clock
// cycles summed over all workgroups; read and reset with als_pair_profile_read().  Not compiled into the product.
#ifdef PAIR_PROFILE
__device__ unsigned long long g_pair_prof[2][16];
#define PAIR_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                           prof_[i] += now_ - last_; last_ = now_; } while (0)
#else
#define PAIR_STAMP(i) do { } while (0)
#endif

constexpr int PAIR_MAX_WG = 1024;       // persistent workgroups (256 CUs x 4); sizes the flush scratch
constexpr int PAIR_FLUSH_CHUNKS = 8;    // 64-rating chunks between flushes (512 ratings, as FLUSH_GROUPS = 16)

template <int KB>
struct PairCfg {
    using C = KCfg<KB>;
    static constexpr int KP = C::KP;
    static constexpr int XA = (KB == 8) ? 2 : 1;            // B's block rows whose cross blocks wave A computes
    static constexpr int NSLOT = 4 + XA;                    // blocks published per group (A: 4, B: XA)
    static constexpr int XBUF = NSLOT * 3 * 256;            // dwords of one exchange buffer (H, M, L planes)
    static constexpr int IMG = C::LDS_FLOATS;
    static constexpr int MAIN = (2 * XBUF > IMG) ? 2 * XBUF : IMG;
    static constexpr int LDS_FLOATS = MAIN + 4 * 128;       // + running rhs, F^T r, F^T 1, y
    static constexpr int NACC_A = 10 + 4 * XA;
    static constexpr int NACC_B = (KB - 4) * (KB - 3) / 2 + 4 * (KB - 4 - XA);
    static constexpr int NACC_W = NACC_A > NACC_B ? NACC_A : NACC_B;
    static constexpr int SCRATCH_FLOATS = 2 * NACC_W * 256; // flush totals of one workgroup
};

// what wave W (0 = A, 1 = B) owns
template <int KB, int W>
struct Own {
    static constexpr int XA = PairCfg<KB>::XA;
    static constexpr int NO = W == 0 ? 4 : KB - 4;          // own column blocks
    static constexpr int B0 = W == 0 ? 0 : 4;               // first of them
    static constexpr int NOWN = NO * (NO + 1) / 2;          // lower blocks among the own columns
    static constexpr int NX = W == 0 ? XA : KB - 4 - XA;    // cross block rows (block row >= 4, block column < 4)
    static constexpr int X0 = W == 0 ? 4 : 4 + XA;          // first of them
    static constexpr int NACC = NOWN + 4 * NX;
    static constexpr int own_idx(int I, int K) { return (I - B0) * (I - B0 + 1) / 2 + (K - B0); }
    static constexpr int cross_idx(int I, int K) { return NOWN + (I - X0) * 4 + K; }
    // global lower block (I, K), K <= I: owned by this wave?  local accumulator index
    static constexpr bool owns(int I, int K) {
        if (I < 4) return W == 0;
        if (K >= 4) return W == 1;
        return W == 0 ? (I < 4 + XA) : (I >= 4 + XA);
    }
    static constexpr int idx(int I, int K) {
        return !owns(I, K) ? 0 : ((K >= 4 || I < 4) ? own_idx(I, K) : cross_idx(I, K));
    }
};

// An element offset the optimiser knows nothing about: address arithmetic based on it cannot be hoisted out of the
// (rarely taken, fully unrolled) block that uses it - hoisted, the 72 addresses of a flush alone take 144 registers
// for the whole kernel and everything else spills.  (Laundering the pointer itself would lose its address space:
// flat instead of global instructions.)
__device__ __forceinline__ int64_t opaque(int64_t off) {
    asm volatile("" : "+v"(off));
    return off;
}

// one Gram block += six cross products of the split operands, smallest terms first (as process_chunk_bf16x3)
__device__ __forceinline__ f32x4 mfma6(f32x4 acc, const i32x4& Hi, const i32x4& Mi, const i32x4& Li,
                                       const i32x4& Hj, const i32x4& Mj, const i32x4& Lj) {
    const bf16x8 hi = __builtin_bit_cast(bf16x8, Hi), hj = __builtin_bit_cast(bf16x8, Hj);
    const bf16x8 mi = __builtin_bit_cast(bf16x8, Mi), mj = __builtin_bit_cast(bf16x8, Mj);
    const bf16x8 li = __builtin_bit_cast(bf16x8, Li), lj = __builtin_bit_cast(bf16x8, Lj);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(li, hj, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, lj, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mi, mj, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mi, hj, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, mj, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi, hj, acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------------------------
// one 32-rating group: lane (c, q) takes ratings 8q .. 8q+7 of the group and its own column blocks.
// Contains ONE workgroup barrier: both waves call it the same number of times.
// ---------------------------------------------------------------------------------------------------------
// gather of one 32-rating group: lane (c, q) fetches its NO own floats of ratings 8q .. 8q+7 of group g of the
// chunk whose row offsets are off_l (lane t = rating t).  Issued one group AHEAD of its use (the wave has the
// registers for it - the one-wave kernel does not), so the gather latency hides behind the previous group.
template <int KB, int NO>
__device__ __forceinline__ void pair_load(float (&f)[8][NO], int off_l, int g, const float* __restrict__ Fc, int q) {
    int off_t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) off_t[j] = bperm_i(off_l, 32 * g + 8 * q + j);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* src = Fc + (uint32_t)off_t[j];
        if constexpr (KB % 4 == 0 && NO == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            f[j][0] = v.x; f[j][1] = v.y; f[j][2] = v.z; f[j][3] = v.w;
        } else {
#pragma unroll
            for (int b = 0; b < NO; ++b) f[j][b] = src[b];
        }
    }
}

template <int KB, int W, int NACC, int NO>
__device__ __forceinline__ void pair_group(f32x4 (&acc)[NACC], float (&rhs)[NO], float (&cs)[NO],
                                           const float (&f)[8][NO], float r_l, int g, int q, int lane,
                                           i32x4* __restrict__ xb) {
    using O = Own<KB, W>;
    static_assert(NACC == O::NACC && NO == O::NO, "accumulator set of wave W");
    constexpr int XA = O::XA;
    float r_t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r_t[j] = bperm_f(r_l, 32 * g + 8 * q + j);
    i32x4 H[NO], M[NO], L[NO];
#pragma unroll
    for (int b = 0; b < NO; ++b) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const float x0 = f[j][b], x1 = f[j + 1][b];
            rhs[b] = fmaf(x0, r_t[j], rhs[b]);
            rhs[b] = fmaf(x1, r_t[j + 1], rhs[b]);
            cs[b] += x0 + x1;
            int hw, mw, lw;
            split3(x0, x1, hw, mw, lw);
            H[b][j >> 1] = hw; M[b][j >> 1] = mw; L[b][j >> 1] = lw;
        }
    }
    // publish: A its four blocks (slots 0..3), B its first XA (slots 4..)
    constexpr int NPUB = W == 0 ? 4 : XA, S0 = W == 0 ? 0 : 4;
#pragma unroll
    for (int b = 0; b < NPUB; ++b) {
        xb[((S0 + b) * 3 + 0) * 64 + lane] = H[b];
        xb[((S0 + b) * 3 + 1) * 64 + lane] = M[b];
        xb[((S0 + b) * 3 + 2) * 64 + lane] = L[b];
    }
    // blocks among the own columns: no foreign operand, issued in front of the barrier
#pragma unroll
    for (int bi = 0; bi < NO; ++bi)
#pragma unroll
        for (int bj = 0; bj <= bi; ++bj)
            acc[bi * (bi + 1) / 2 + bj] = mfma6(acc[bi * (bi + 1) / 2 + bj], H[bi], M[bi], L[bi], H[bj], M[bj], L[bj]);
    __syncthreads();
    if constexpr (W == 0) {
        // block rows 4 .. 4+XA-1 (B's columns, row operand foreign) x own columns 0..3
#pragma unroll
        for (int xi = 0; xi < XA; ++xi) {
            const i32x4 fh = xb[((4 + xi) * 3 + 0) * 64 + lane];
            const i32x4 fm = xb[((4 + xi) * 3 + 1) * 64 + lane];
            const i32x4 fl = xb[((4 + xi) * 3 + 2) * 64 + lane];
#pragma unroll
            for (int K = 0; K < 4; ++K)
                acc[O::NOWN + xi * 4 + K] = mfma6(acc[O::NOWN + xi * 4 + K], fh, fm, fl, H[K], M[K], L[K]);
        }
    } else {
        // own block rows 4+XA .. KB-1 (local XA ..) x A's columns 0..3 (column operand foreign)
#pragma unroll
        for (int K = 0; K < 4; ++K) {
            const i32x4 fh = xb[(K * 3 + 0) * 64 + lane];
            const i32x4 fm = xb[(K * 3 + 1) * 64 + lane];
            const i32x4 fl = xb[(K * 3 + 2) * 64 + lane];
#pragma unroll
            for (int xi = 0; xi < O::NX; ++xi)
                acc[O::NOWN + xi * 4 + K] = mfma6(acc[O::NOWN + xi * 4 + K], H[XA + xi], M[XA + xi], L[XA + xi], fh, fm, fl);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Cholesky panel on the LDS-resident trailing matrix.  NS row sets per lane: set 0 holds the diagonal block's
// rows in lanes 0..15 (both waves) and rows of the block column in lanes 16..63; a second set (wave B, first
// panel of KB = 8 only: 112 rows below the diagonal block) holds 48 more rows in lanes 16..63.
// ---------------------------------------------------------------------------------------------------------
template <int NS, int T2>
__device__ __forceinline__ void pair_trailing(float (&p)[NS][16], const float (&l)[NS], float lrep) {
    if constexpr (T2 < 16) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                : "+v"(p[s][T2]) : "v"(lrep), "v"(l[s]), "n"(T2));
        pair_trailing<NS, T2 + 1>(p, l, lrep);
    }
}

// pivots T .. 15 of the panel.  prep: unscaled column T of the diagonal block (lanes 0..15 of set 0) replicated into
// every 16-lane row; inv = 1 / sqrt(pivot T).  Column T + 1 is updated FIRST and the next pivot's cross-lane round
// trip (~100 cycles) and rsq are issued right behind it, underneath the updates of columns T + 2 .. 15.
template <int NS, int T>
__device__ __forceinline__ void pair_pivots(float (&p)[NS][16], float (&b)[NS], float& ysel, int lane, float prep,
                                            float inv) {
    float l[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { l[s] = p[s][T] * inv; p[s][T] = l[s]; }
    float lrep = -(prep * inv);
    asm("s_nop 1" : "+v"(lrep));        // VALU write -> DPP read: 2 wait states, invisible to the hazard recogniser
    const float yt = readlane_f(b[0], T) * inv;
    if constexpr (T + 1 < 16) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
            asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                : "+v"(p[s][T + 1]) : "v"(lrep), "v"(l[s]), "n"(T + 1));
        const float prep_n = bperm_f(p[0][T + 1], lane & 15);
        const float inv_n = __builtin_amdgcn_rsqf(readlane_f(p[0][T + 1], T + 1));
        pair_trailing<NS, T + 2>(p, l, lrep);
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s] = fmaf(-l[s], yt, b[s]);
        ysel = select_lanes<1ull << T>(yt, ysel);
        pair_pivots<NS, T + 1>(p, b, ysel, lane, prep_n, inv_n);
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) b[s] = fmaf(-l[s], yt, b[s]);
        ysel = select_lanes<1ull << T>(yt, ysel);
    }
}

// factorise block column J.  Rows below the diagonal block are written back here (the trailing update reads
// them after the caller's barrier); the diagonal block's rows stay in pd (lanes 0..15): wave A stores them AFTER
// the barrier, because wave B reads the unfactorised diagonal block at the start of its own panel.
template <int KB, int W, int NS>
__device__ __forceinline__ void pair_panel(float* __restrict__ img, float* __restrict__ bvec, int J, int lane,
                                           float (&pd)[16], float& ysel) {
    using C = KCfg<KB>;
    const int OFF = C::lcol_off_rt(J);
    const int live = C::KP - 16 * J;               // rows of block column J
    float p[NS][16], b[NS];
    int rowrel[NS];
    bool store[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int u = W + s;                       // 48-row unit: (A, set 0) = 0, (B, set 0) = 1, (B, set 1) = 2
        const bool diag = (s == 0) && lane < 16;
        const int rr = diag ? lane : 16 + 48 * u + (lane - 16);
        const bool ok = diag || (lane >= 16 && rr < live);
        store[s] = ok && !diag;
        rowrel[s] = ok ? rr : live - 1;            // idle lanes read the last row and store nothing
        const int si = (rowrel[s] >> 2) & 3;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(img + OFF + rowrel[s] * 16 + ((g ^ si) << 2));
            p[s][4 * g] = v.x; p[s][4 * g + 1] = v.y; p[s][4 * g + 2] = v.z; p[s][4 * g + 3] = v.w;
        }
        b[s] = bvec[16 * J + rowrel[s]];
    }
    wave_lds_sync();
    pair_pivots<NS, 0>(p, b, ysel, lane, bperm_f(p[0][0], lane & 15), __builtin_amdgcn_rsqf(readlane_f(p[0][0], 0)));
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (store[s]) {
            const int si = (rowrel[s] >> 2) & 3;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 v = {p[s][4 * g], p[s][4 * g + 1], p[s][4 * g + 2], p[s][4 * g + 3]};
                *reinterpret_cast<f32x4*>(img + OFF + rowrel[s] * 16 + ((g ^ si) << 2)) = v;
            }
            bvec[16 * J + rowrel[s]] = b[s];
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) pd[t] = p[0][t];
}

// rank-16 update of the blocks (I, K), J < K <= I, dealt alternately to the two waves.  Lane (c, q) holds
// elements (row 16I + c, cols 16K + 4q .. 4q+3) of its block: the transposed product, so that C is one b128.
// J is a template parameter (the caller switches on it): block list, LDS offsets and ownership are compile-time
// constants, all operand rows and all of the wave's C blocks are requested up front and the independent MFMA
// chains of the blocks overlap.
template <int KB, int W, int J>
__device__ __forceinline__ void pair_update(float* __restrict__ img, int lane) {
    using C = KCfg<KB>;
    constexpr int N = KB - J - 1;                   // block rows below the diagonal block
    if constexpr (N > 0) {
        const int c = lane & 15, q = lane >> 4;
        float* base = img + c * 16 + ((q ^ ((c >> 2) & 3)) << 2);
        f32x4 op[N], nop[N];
#pragma unroll
        for (int R = 0; R < N; ++R) {
            op[R] = *reinterpret_cast<const f32x4*>(base + C::lcol_off(J) + 16 * (R + 1) * 16);
            nop[R] = -op[R];
            // negated once per row: without this the compiler re-forms the negation in front of every MFMA (a
            // VALU write the MFMA has to wait two states for)
            asm volatile("" : "+v"(nop[R]));
        }
        constexpr int NB = N * (N + 1) / 2, NW = (NB + 1 - W) / 2;      // blocks of this wave: t = W, W + 2, ...
        f32x4 Cv[NW > 0 ? NW : 1];
        {
            int t = 0, u = 0;
#pragma unroll
            for (int K = J + 1; K < KB; ++K)
#pragma unroll
                for (int I = K; I < KB; ++I, ++t)
                    if ((t & 1) == W) Cv[u++] = *reinterpret_cast<const f32x4*>(base + C::lcol_off(K) + 16 * (I - K) * 16);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int t = 0, u = 0;
#pragma unroll
            for (int K = J + 1; K < KB; ++K)
#pragma unroll
                for (int I = K; I < KB; ++I, ++t)
                    if ((t & 1) == W) {
                        Cv[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(nop[K - J - 1][e], op[I - J - 1][e], Cv[u], 0, 0, 0);
                        ++u;
                    }
        }
        {
            int t = 0, u = 0;
#pragma unroll
            for (int K = J + 1; K < KB; ++K)
#pragma unroll
                for (int I = K; I < KB; ++I, ++t)
                    if ((t & 1) == W) *reinterpret_cast<f32x4*>(base + C::lcol_off(K) + 16 * (I - K) * 16) = Cv[u++];
        }
    }
}

template <int KB, int W, int J0>
__device__ __forceinline__ void pair_update_dispatch(float* __restrict__ img, int J, int lane) {
    if (J == J0) pair_update<KB, W, J0>(img, lane);
    else if constexpr (J0 + 2 < KB) pair_update_dispatch<KB, W, J0 + 1>(img, J, lane);
}

// ---------------------------------------------------------------------------------------------------------
// the wave's program
// ---------------------------------------------------------------------------------------------------------
template <int KB, int W>
__device__ __forceinline__ void pair_body(const als_row_solve_params& P, float* __restrict__ lds, int lane) {
    using PC = PairCfg<KB>;
    using O = Own<KB, W>;
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NO = O::NO, B0 = O::B0, NACC = O::NACC;
    const int c = lane & 15, q = lane >> 4;
    float* img = lds;
    i32x4* xbuf = reinterpret_cast<i32x4*>(lds);
    float* bvec = lds + PC::MAIN;
    float* rhs0vec = bvec + 128;
    float* csvec = bvec + 256;
    float* yvec = bvec + 384;
    float* totals = (float*)P.scratch + ((size_t)blockIdx.x * 2 + W) * (size_t)(PC::NACC_W * 256);
    const float mu = (float)(*P.mu);
    const float* Fc = P.F + KB * c + B0;
    const int swzc = (((c >> 2) ^ q) << 2) + (c & 3);
    int parity = 0;                                 // exchange buffer of the next group (same in both waves)
#ifdef PAIR_PROFILE
    unsigned long long prof_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long last_ = __builtin_amdgcn_s_memtime();
#endif

    for (int64_t tid = blockIdx.x; tid < P.ntasks; tid += gridDim.x) {
        __syncthreads();                            // the previous task's image / vectors are no longer read
        PAIR_STAMP(0);
        const als_task t = P.tasks[tid];
        const int row = t.row;
        const int64_t r64 = row;
        const int64_t rbeg = P.indptr[row], rend = P.indptr[row + 1];
        const int64_t beg = rbeg + (int64_t)t.seg * ALS_SPLIT_CHUNK;
        const int len = (int)min((int64_t)ALS_SPLIT_CHUNK, rend - beg);
        const float bself = P.bias_self[row];
        const int32_t* idxp = P.indices + beg;
        const float* valp = P.vals + beg;

        f32x4 acc[NACC];
        float rhs[NO], cs[NO];
        float sumr = 0.f, sumr2 = 0.f;
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < NO; ++b) { rhs[b] = 0.f; cs[b] = 0.f; }

        // ---- Gram: 64-rating chunks, indices two chunks ahead, value + opposite bias one ahead -------------
        int nflush = 0;
        int idx1 = (lane < len) ? __builtin_nontemporal_load(idxp + lane) : P.F_zero_row;
        int idx2 = (64 + lane < len) ? __builtin_nontemporal_load(idxp + 64 + lane) : P.F_zero_row;
        float val1 = (lane < len) ? __builtin_nontemporal_load(valp + lane) : 0.f;
        float bo1 = (lane < len) ? P.bias_other[idx1] : 0.f;
        float f[8][NO];
        pair_load<KB, NO>(f, idx1 * P.ld, 0, Fc, q);          // first group of the row
        for (int base = 0; base < len; base += 64) {
            const int idx0 = idx1;
            const float val0 = val1, bo0 = bo1;
            idx1 = idx2;
            const int t2 = base + 128 + lane;
            idx2 = (t2 < len) ? __builtin_nontemporal_load(idxp + t2) : P.F_zero_row;
            const int t1 = base + 64 + lane;
            val1 = (t1 < len) ? __builtin_nontemporal_load(valp + t1) : 0.f;
            bo1 = (t1 < len) ? P.bias_other[idx1] : 0.f;
            const int nvalid = min(64, len - base);
            const bool ok = lane < nvalid;
            const float rb = val0 - mu - bo0;
            if constexpr (W == 0) {
                sumr += ok ? rb : 0.f;
                sumr2 = ok ? fmaf(rb, rb, sumr2) : sumr2;
            }
            const float r0 = ok ? (rb - bself) : 0.f;
            const int off0 = idx0 * P.ld;
            const int off1 = idx1 * P.ld;                   // next chunk (lanes past the row: the zero row)
            const int ng = nvalid > 32 ? 2 : 1;
#pragma unroll 1
            for (int g = 0; g < ng; ++g) {
                const bool more = g + 1 < ng;
                const bool any = more || base + 64 < len;
                float fn[8][NO];
                if (any) pair_load<KB, NO>(fn, more ? off0 : off1, more ? g + 1 : 0, Fc, q);
                pair_group<KB, W>(acc, rhs, cs, f, r0, g, q, lane, xbuf + parity * (PC::XBUF / 4));
                parity ^= 1;
                if (any) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
#pragma unroll
                        for (int b = 0; b < NO; ++b) f[j][b] = fn[j][b];
                }
            }
            if (((base >> 6) + 1) % PAIR_FLUSH_CHUNKS == 0 && base + 64 < len) {
                float* tl = totals + opaque(lane);
                if (nflush == 0) {
#pragma unroll
                    for (int a = 0; a < NACC; ++a)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { tl[(a * 4 + r) * 64] = acc[a][r]; acc[a][r] = 0.f; }
                } else {
#pragma unroll
                    for (int a = 0; a < NACC; a += 2) {         // 8 loads in flight
                        float tv[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) tv[e] = (a + (e >> 2) < NACC) ? tl[((a + (e >> 2)) * 4 + (e & 3)) * 64] : 0.f;
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (a + (e >> 2) < NACC) {
                                tl[((a + (e >> 2)) * 4 + (e & 3)) * 64] = tv[e] + acc[a + (e >> 2)][e & 3];
                                acc[a + (e >> 2)][e & 3] = 0.f;
                            }
                    }
                }
                ++nflush;
            }
        }
        if (nflush > 0) {
            const float* tl = totals + opaque(lane);
#pragma unroll
            for (int a = 0; a < NACC; a += 2) {
                float tv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) tv[e] = (a + (e >> 2) < NACC) ? tl[((a + (e >> 2)) * 4 + (e & 3)) * 64] : 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (a + (e >> 2) < NACC) acc[a + (e >> 2)][e & 3] += tv[e];
            }
        }

        PAIR_STAMP(1);
        if (t.slot >= 0) {      // segment of a split row: partial in the slot format of store_partial (row_solve.hip)
            float* ws = (float*)P.workspace + opaque((int64_t)t.slot * C::SLOT_ITEMS * 64 + lane);
#pragma unroll
            for (int I = 0; I < KB; ++I)
#pragma unroll
                for (int K = 0; K <= I; ++K)
                    if (O::owns(I, K)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) ws[(blk_idx(I, K) * 4 + r) * 64] = acc[O::idx(I, K)][r];
                    }
#pragma unroll
            for (int b = 0; b < NO; ++b) {
                ws[(C::NACC * 4 + B0 + b) * 64] = rhs[b];
                ws[(C::NACC * 4 + KB + B0 + b) * 64] = cs[b];
            }
            if constexpr (W == 0) {
                ws[(C::NACC * 4 + 2 * KB) * 64] = sumr;
                ws[(C::NACC * 4 + 2 * KB + 1) * 64] = sumr2;
            }
            continue;
        }

        __syncthreads();                            // both waves are done with the exchange buffers
        PAIR_STAMP(2);
        // ---- accumulators -> image (C/D layout: row 4q + r, col c of the block), Gram by-product ----------
        {
            float* imgd = img + 4 * q * 16 + swzc;
#pragma unroll
            for (int I = 0; I < KB; ++I)
#pragma unroll
                for (int K = 0; K <= I; ++K)
                    if (O::owns(I, K)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) imgd[C::lcol_off(K) + (16 * (I - K) + r) * 16] = acc[O::idx(I, K)][r];
                    }
            if (P.gram_out) {
                float* G = P.gram_out + opaque(r64 * KP * KP + 4 * q * KP + c);
#pragma unroll
                for (int I = 0; I < KB; ++I)
#pragma unroll
                    for (int K = 0; K <= I; ++K)
                        if (O::owns(I, K)) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) G[(16 * I + r) * KP + 16 * K] = acc[O::idx(I, K)][r];
                        }
            }
        }
        // ---- right-hand side / column sums: perm position 16 (B0 + q) + c = 64 W + lane -------------------
#pragma unroll
        for (int b = 0; b < NO; ++b) {
            rhs[b] += __shfl_xor(rhs[b], 16, 64); rhs[b] += __shfl_xor(rhs[b], 32, 64);
            cs[b] += __shfl_xor(cs[b], 16, 64);   cs[b] += __shfl_xor(cs[b], 32, 64);
        }
        float bsel = 0.f, csel = 0.f;
#pragma unroll
        for (int e = 0; e < NO; ++e) {
            bsel = (q == e) ? rhs[e] : bsel;
            csel = (q == e) ? cs[e] : csel;
        }
        const int pos = 64 * W + lane;
        const bool mine = q < NO;                    // KB = 7: the last quarter of wave B has no position
        const float lam = (P.lambda_row ? P.lambda_row[row] : P.lambda_scalar) + ALS_EPS
                        + (P.diag_extra ? P.diag_extra[row] : 0.f);
        float sumr_w = 0.f, sumr2_w = 0.f;
        if constexpr (W == 0) { sumr_w = wave_sum(sumr); sumr2_w = wave_sum(sumr2); }
        if (mine) {
            const int col = perm_to_col<KB>(pos);
            rhs0vec[pos] = bsel;
            csvec[pos] = csel;
            bvec[pos] = bsel + ((P.rhs_extra && !P.factor_out) ? P.rhs_extra[r64 * P.ld + col] : 0.f);
            if (P.rhs_out) P.rhs_out[r64 * KP + pos] = bsel;
            if (P.colsum_out) P.colsum_out[r64 * KP + pos] = csel;
            // regulariser on the diagonal of the blocks this wave has just written; padded columns get 1
            img[C::lcol_off_rt(pos >> 4) + c * 16 + (c & 3)] += (col < P.k) ? lam : 1.0f;
        }
        if constexpr (W == 0) {
            if (P.sumr_out && lane == 0) P.sumr_out[row] = sumr_w;
            if (P.sumr2_out && lane == 0) P.sumr2_out[row] = sumr2_w;
        }
        PAIR_STAMP(3);
        __syncthreads();
        PAIR_STAMP(4);

        // ---- blocked Cholesky on the image --------------------------------------------------------------
#pragma unroll 1
        for (int J = 0; J < KB; ++J) {
            float pd[16];
            float ysel = 0.f;
            if (W == 1 && KB == 8 && J == 0) pair_panel<KB, W, 2>(img, bvec, J, lane, pd, ysel);
            else                             pair_panel<KB, W, 1>(img, bvec, J, lane, pd, ysel);
            PAIR_STAMP(5);
            __syncthreads();
            PAIR_STAMP(6);
            if constexpr (W == 0) {
                if (lane < 16) {                    // L of the diagonal block (its upper part is never read)
                    const int OFF = C::lcol_off_rt(J);
                    const int si = (lane >> 2) & 3;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = {pd[4 * g], pd[4 * g + 1], pd[4 * g + 2], pd[4 * g + 3]};
                        *reinterpret_cast<f32x4*>(img + OFF + lane * 16 + ((g ^ si) << 2)) = v;
                    }
                    yvec[16 * J + lane] = ysel;
                }
            }
            pair_update_dispatch<KB, W, 0>(img, J, lane);
            PAIR_STAMP(7);
            __syncthreads();
            PAIR_STAMP(8);
        }

        // ---- factor-only mode: symmetric completion of L, 1 / L_ii on the diagonal, perm space -------------
        if (P.factor_out) {
            const int i = pos, ic = min(i, KP - 1);
            const float di = __builtin_amdgcn_rcpf(img[C::lcol_off_rt(ic >> 4) + c * 16 + (c & 3)]);
            const bool bad = (i < KP) && !(di > 0.f && di < __builtin_inff());
            if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicMax(P.status, row + 1);
            float* M = P.factor_out + r64 * KP * KP;
            const int colb = C::lcol_off_rt(ic >> 4) - 16 * (ic >> 4) * 16 + (c & 3);
#pragma unroll 8
            for (int p = 0; p < KP; ++p) {
                const int Jp = p >> 4;
                const int rowv = max(ic, 16 * Jp);                                   // row i inside block column Jp
                const float lrow = img[C::lcol_off_rt(Jp) + (rowv - 16 * Jp) * 16 +
                                       ((((p >> 2) & 3) ^ ((rowv >> 2) & 3)) << 2) + (p & 3)];
                const int rsafe = max(p, ic & ~15);
                const float lcolv = img[colb + rsafe * 16 + (((c >> 2) ^ ((rsafe >> 2) & 3)) << 2)];
                if (i < KP) __builtin_nontemporal_store((p == i) ? di : (p < i ? lrow : lcolv), M + p * KP + i);
            }
            PAIR_STAMP(9);
            continue;
        }

        // ---- transposed solve, bias, statistics: wave A -------------------------------------------------
        if constexpr (W == 0) {
            constexpr int NR = C::NR;
            Chol<KB> S;
            float csrow[NR], rhs0[NR];
            int colrow[NR];
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const int i = min(lane + 64 * rr, KP - 1);
                S.y[rr] = yvec[i];
                S.b[rr] = 0.f;
                S.di[rr] = __builtin_amdgcn_rcpf(img[C::lcol_off_rt(i >> 4) + c * 16 + (c & 3)]);
                csrow[rr] = csvec[i];
                rhs0[rr] = rhs0vec[i];
                colrow[rr] = perm_to_col<KB>(i);
            }
            if (!chol_spd<KB>(S, lane) && lane == 0) atomicMax(P.status, row + 1);
            float x[NR];
            backward_solve<KB>(img, S, x, lane);
            PAIR_STAMP(11);
            float dot = 0.f, xr = 0.f, yy = 0.f, xx = 0.f;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if (lane + 64 * rr < KP) {
                    P.X_out[r64 * P.ld + colrow[rr]] = x[rr];
                    dot = fmaf(csrow[rr], x[rr], dot);
                    xr = fmaf(rhs0[rr], x[rr], xr);
                    yy = fmaf(S.y[rr], S.y[rr], yy);
                    xx = fmaf(x[rr], x[rr], xx);
                }
            }
            dot = wave_sum(dot);
            const float nnz = (float)(rend - rbeg);
            const float lb = P.lambda_bias_row ? P.lambda_bias_row[row] : P.lambda_bias_scalar;
            const float bnew = (sumr_w - dot) / (nnz + lb + ALS_EPS);
            if (lane == 0) P.bias_out[row] = bnew;      // bself was read at the top of the task
            if (P.stat_out) {       // closed-form residual sums, as finish_row (row_solve.hip)
                xr = wave_sum(xr); yy = wave_sum(yy); xx = wave_sum(xx);
                if (lane == 0) {
                    const double bn = bnew, dt = dot;
                    const double s1 = (double)sumr_w - (double)nnz * bn;
                    const double s2 = (double)sumr2_w - 2.0 * bn * (double)sumr_w + (double)nnz * bn * bn;
                    const double cross = (double)xr + ((double)bself - bn) * dt;
                    const double quad = (double)yy - (double)lam * (double)xx;
                    P.stat_out[2 * r64] = (float)(s1 - dt);
                    P.stat_out[2 * r64 + 1] = (float)(s2 - 2.0 * cross + quad);
                }
            }
        }
        PAIR_STAMP(10);
    }
#ifdef PAIR_PROFILE
    if (lane == 0) {
        for (int i = 0; i < 16; ++i) atomicAdd(&g_pair_prof[W][i], prof_[i]);
    }
#endif
}

template <int KB>
__global__ __launch_bounds__(128, 2)
void k_row_pair(const als_row_solve_params P) {
    __shared__ __attribute__((aligned(16))) float lds[PairCfg<KB>::LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0) pair_body<KB, 0>(P, lds, lane);
    else                                                        pair_body<KB, 1>(P, lds, lane);
}

template <int KB>
int launch_pair(const als_row_solve_params* p, hipStream_t st) {
    static int wg_per_cu[64];           // per device, 0 = not queried yet
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return ALS_E_LAUNCH;
    static int ncu[64];
    if (wg_per_cu[dev] == 0) {
        int occ = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_row_pair<KB>, 128, 0) != hipSuccess || occ < 1 ||
            hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return ALS_E_LAUNCH;
        wg_per_cu[dev] = occ;
        ncu[dev] = prop.multiProcessorCount;
    }
    int64_t grid = (int64_t)wg_per_cu[dev] * ncu[dev];
    if (grid > PAIR_MAX_WG) grid = PAIR_MAX_WG;
    if (grid > p->ntasks) grid = p->ntasks;
    hipLaunchKernelGGL(k_row_pair<KB>, dim3((unsigned)grid), dim3(128), 0, st, *p);
    return hipGetLastError() == hipSuccess ? 0 : ALS_E_LAUNCH;
}

}  // namespace

#ifdef PAIR_PROFILE
// out[2][16] (host): cycles per phase and wave role since the last call; synchronises the device
extern "C" int als_pair_profile_read(unsigned long long* out) {
    if (hipDeviceSynchronize() != hipSuccess) return ALS_E_LAUNCH;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_prof), sizeof(unsigned long long) * 32) != hipSuccess) return ALS_E_LAUNCH;
    unsigned long long zero[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pair_prof), zero, sizeof(zero)) == hipSuccess ? 0 : ALS_E_LAUNCH;
}
#endif

// bytes of als_row_solve_params::scratch for k factors (0: the two-wave kernel does not serve this k)
extern "C" int64_t als_row_solve_scratch_bytes(int k) {
    const int ld = als_padded_k(k);
    if (ld < 0) return ALS_E_BADK;
    switch (ld / 16) {
        case 7: return (int64_t)PAIR_MAX_WG * PairCfg<7>::SCRATCH_FLOATS * (int64_t)sizeof(float);
        case 8: return (int64_t)PAIR_MAX_WG * PairCfg<8>::SCRATCH_FLOATS * (int64_t)sizeof(float);
    }
    return 0;
}

// called by als_row_solve (row_solve.hip) for the primal tasks of a bf16x3 call with p->scratch set
int als_row_pair_dispatch(const als_row_solve_params* p, hipStream_t st) {
    if (p->ntasks <= 0) return 0;
    switch (p->ld / 16) {
        case 7: return launch_pair<7>(p, st);
        case 8: return launch_pair<8>(p, st);
    }
    return ALS_E_BADK;
}
