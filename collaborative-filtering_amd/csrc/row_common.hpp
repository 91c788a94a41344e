// Pieces of the row-solve kernels (row_solve.hip): the fp16 operand split, the per-wave Cholesky state, the DPP
// panel update and the transposed solve from the L image in LDS.
#pragma once
#include <type_traits>
#include "als_device.hpp"

namespace {

__host__ __device__ constexpr int blk_idx(int I, int K) { return I * (I + 1) / 2 + K; }

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// 2-way fp16 split of two floats, fp32-equivalent (see process_chunk_f16x2): with S an exact power-of-two scale
// that puts the largest |x| of the whole factor matrix just below 2^15 (als_factor_scale),
//     x S = h + l + e,   h = fp16(x S),  l = fp16(x S - h)  (both round-to-nearest),  |e| <= 2^-23 |x S|:
// h carries 11 significant bits, the remainder is exact in fp32 and at most half an ulp of h, so its own 11 bits
// (plus its sign) reach the 2^-23 level - the fp32 representation error is 2^-24.  The scale keeps l out of the
// fp16 subnormal range for every element within 2^-14 of the matrix maximum (smaller elements lose relative,
// not absolute, precision: below 2^-38 of the largest Gram term).  6 VALU per pair: 2 v_mul, 2 v_cvt_pk_f16_f32,
// 2 v_fma_mix_f32 (the bf16 3-way split by truncation of rounds 1-2 took 11 and twice the matrix instructions).
__device__ __forceinline__ void split2(float x0, float x1, float S, int& H, int& L) {
    const float y0 = x0 * S, y1 = x1 * S;
    const h16x2 h = __builtin_convertvector(f32x2{y0, y1}, h16x2);
    const float d0 = fmaf((float)h[0], -1.0f, y0), d1 = fmaf((float)h[1], -1.0f, y1);        // exact
    const h16x2 l = __builtin_convertvector(f32x2{d0, d1}, h16x2);
    H = __builtin_bit_cast(int, h);
    L = __builtin_bit_cast(int, l);
}

// Between the last v_mfma_f32_16x16x32_f16 of a Gram loop and the first vector instruction that reads the
// accumulators.  hipcc (ROCm 7.2) pads this matrix-result -> vector-read hazard (an 8-pass matrix instruction: 11
// wait states) too little on one path of k_row_dual - when the branch around the optional lambda_row load is taken
// the unscaling v_pk_mul_f32 follows the last MFMAs within a handful of instructions: a few rows in 10^5 came out as
// garbage / NaN, differently from run to run (cfg5-small, rows of 33 ... 48 ratings; found with
// profiles/debug/dual_probe.py, round 3).  The wait states are therefore spelled out - 20 idle cycles per row -
// and tests/test_isa_hazards_cpu.py checks the distance in the built code object.
__device__ __forceinline__ void mfma_results_settle() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------
// blocked Cholesky state of one wave
// ---------------------------------------------------------------------------
template <int KB>
struct Chol {
    static constexpr int NR = KCfg<KB>::NR;
    float di[NR];       // 1 / L[i][i] of the lane's rows i = lane + 64*rr
    float y[NR];        // forward-solved right-hand side (when SOLVE)
    float b[NR];        // running right-hand side
};

// positive definiteness, checked once after the factorisation: pivot d <= 0 (or NaN) leaves rsq(d) = inf / NaN
// in di, and everything computed after it is NaN as well - no per-pivot compare needed
template <int KB>
__device__ __forceinline__ bool chol_spd(const Chol<KB>& S, int lane) {
    bool bad = false;
#pragma unroll
    for (int rr = 0; rr < KCfg<KB>::NR; ++rr) {
        const float v = S.di[rr];
        bad = bad || ((lane + 64 * rr < KCfg<KB>::KP) && !(v > 0.f && v < __builtin_inff()));
    }
    return __builtin_amdgcn_ballot_w64(bad) == 0;
}

// one pivot of panel J (column 16J + T): scale the column, ride the forward
// substitution along, update the remaining columns of the panel
// p[.][t2] += l * (-L[16J+t2][T]) for t2 = T2 ... 15; the negated multiplier is lane t2 of every 16-lane row of
// lrep and is picked up by the FMA itself (v_fmac_f32_dpp row_newbcast:t2).  The compiler only emits DPP on
// v_mov here (its DPP combiner has no VOP3 v_fma form on gfx9), so the VOP2 form is written out; the caller
// orders an s_nop between the write of lrep and these reads (wait states the hazard recogniser cannot see).
template <int KB, int T2>
__device__ __forceinline__ void panel_trailing(float (&p)[KCfg<KB>::NR][16], const float (&l)[KCfg<KB>::NR], float lrep) {
    if constexpr (T2 < 16) {
#pragma unroll
        for (int rr = 0; rr < KCfg<KB>::NR; ++rr)
            asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
                : "+v"(p[rr][T2]) : "v"(lrep), "v"(l[rr]), "n"(T2));
        panel_trailing<KB, T2 + 1>(p, l, lrep);
    }
}

// coefficients of one 16-step block of the transposed solve: cf[rr][t] = L[16 pb + t][i] / L[i][i] for the
// lanes i = lane + 64 rr below the pivot row, 0 elsewhere (compile-time lane masks)
template <int KB, int RR, int T, int PB>
__device__ __forceinline__ void bwd_coeff_one(const float* __restrict__ Ls, const int (&colbase)[KCfg<KB>::NR][4],
                                              const Chol<KB>& S, float (&cf)[KCfg<KB>::NR][16]) {
    constexpr int prow = 16 * PB + T;
    const float v = Ls[colbase[RR][(T >> 2) & 3] + prow * 16];
    cf[RR][T] = select_lanes<lanes_below<prow, RR>()>(v * S.di[RR], 0.f);
    if constexpr (T + 1 < 16) bwd_coeff_one<KB, RR, T + 1, PB>(Ls, colbase, S, cf);
    else if constexpr (RR + 1 < KCfg<KB>::NR) bwd_coeff_one<KB, RR + 1, 0, PB>(Ls, colbase, S, cf);
}
template <int KB, int PB0>
__device__ __forceinline__ void bwd_coeffs(const float* __restrict__ Ls, const int (&colbase)[KCfg<KB>::NR][4],
                                           const Chol<KB>& S, float (&cf)[KCfg<KB>::NR][16],
                                           std::integral_constant<int, PB0>, int pb) {
    // pb is a compile-time constant at every call site (unrolled loop); dispatch it to the template
    if (pb == PB0) bwd_coeff_one<KB, 0, 0, PB0>(Ls, colbase, S, cf);
    else if constexpr (PB0 + 1 < KB) bwd_coeffs<KB>(Ls, colbase, S, cf, std::integral_constant<int, PB0 + 1>{}, pb);
}

// L^T x = y with L in LDS (block columns).  Lane (+64 rr) owns unknown i = lane + 64 rr and
// reads its column L[p][i], p > i, 16 rows at a time.
template <int KB>
__device__ __forceinline__ void backward_solve(const float* __restrict__ Ls, const Chol<KB>& S,
                                               float (&x)[KCfg<KB>::NR], int lane) {
    using C = KCfg<KB>;
    constexpr int KP = C::KP, NR = C::NR;
    const int c = lane & 15;
    float rs[NR];
    int colbase[NR][4];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = min(lane + 64 * rr, KP - 1);
        const int Ji = i >> 4;
        rs[rr] = S.y[rr] * S.di[rr];
        // element (row, lane's column) of the lane's block column sits at base[g] + row * 16, g = (row >> 2) & 3
        // selecting the swizzled 4-float group.  Rows above the block column (row < 16 Ji) are masked
        // below; their addresses stay inside this wave's image (lcol_off(J) >= 256 J), so the read needs
        // no clamp and every address is one of four per-lane bases plus a compile-time offset.
#pragma unroll
        for (int g = 0; g < 4; ++g)
            colbase[rr][g] = C::lcol_off_rt(Ji) - 16 * Ji * 16 + (c & 3) + (((c >> 2) ^ g) << 2);
    }
#pragma unroll
    for (int pb = KB - 1; pb >= 0; --pb) {
        float cf[NR][16];
        bwd_coeffs<KB>(Ls, colbase, S, cf, std::integral_constant<int, 0>{}, pb);
#pragma unroll
        for (int t = 15; t >= 0; --t) {
            const int prow = 16 * pb + t;
            const float xi = readlane_f(rs[prow >> 6], prow & 63);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) rs[rr] = fmaf(-cf[rr][t], xi, rs[rr]);
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) x[rr] = rs[rr];
}

}  // namespace
