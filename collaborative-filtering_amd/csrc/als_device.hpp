// Device-side helpers shared by the gfx950 ALS kernels.
//
// Layout vocabulary ("Data layout"):
//   KB    number of 16-column factor blocks, ld = KP = 16*KB
//   lane  = (c, q): c = lane & 15 (position in a block), q = lane >> 4
//   perm  factor column `col` sits at perm position 16*(col % KB) + col / KB,
//         so that lane c's contiguous KB floats F[row][KB*c .. KB*c+KB-1] are
//         exactly "position c of every block": one coalesced vector load per
//         lane feeds all MFMA operands with no LDS staging and no shuffles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define ALS_EPS 1e-10f

__device__ __forceinline__ float readlane_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane));
}
__device__ __forceinline__ float bperm_f(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int bperm_i(int v, int src_lane) {
    return __builtin_amdgcn_ds_bpermute(src_lane << 2, v);
}
// (lane in MASK) ? a : b with a compile-time lane mask: the inverse ballot of a literal turns into one
// v_cndmask on an SGPR-pair constant instead of v_cmp (lane, literal) + v_cndmask - the compiler cannot know
// that a comparison of the lane id with a literal is a literal mask.  (No inline asm: the hazard recogniser
// does not see into it, and a VALU write next to MFMAs needs its wait states.)
template <unsigned long long MASK>
__device__ __forceinline__ float select_lanes(float a, float b) {
    return __builtin_amdgcn_inverse_ballot_w64(MASK) ? a : b;
}
// lanes with lane + 64 * RR < BOUND
template <int BOUND, int RR>
constexpr unsigned long long lanes_below() {
    constexpr int n = BOUND - 64 * RR;
    return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull));
}

// Sum over the 64 lanes, result in every lane (wave-uniform).  DPP adds inside the 16-lane rows (xor 1, xor 2,
// half-row mirror, row mirror - a sum does not care which partner), row_bcast15 / row_bcast31 across the rows,
// one v_readlane of lane 63: 6 VALU + 1 readlane instead of 6 x (ds_bpermute + index arithmetic + add).
// The order of the additions is fixed, so the result is reproducible.
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp = [](float x, auto ctrl, auto rows) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value,
                                                          decltype(rows)::value, 0xf, false));
    };
    using std::integral_constant;
    v += dpp(v, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});     // quad_perm [1,0,3,2]
    v += dpp(v, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});     // quad_perm [2,3,0,1]
    v += dpp(v, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{});    // row_half_mirror
    v += dpp(v, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{});    // row_mirror: every lane = row sum
    v += dpp(v, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{});    // row_bcast15 into rows 1, 3
    v += dpp(v, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{});    // row_bcast31 into rows 2, 3
    return readlane_f(v, 63);
}
// Maximum over the 64 lanes of NON-NEGATIVE floats (their bit patterns order like unsigned integers), result in
// every lane: the DPP ladder of wave_sum with v_max_u32.
__device__ __forceinline__ float wave_max_nonneg(float v) {
    auto dpp = [](uint32_t x, auto ctrl, auto rows) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
    };
    using std::integral_constant;
    uint32_t u = (uint32_t)__float_as_int(v);
    u = max(u, dpp(u, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{}));
    u = max(u, dpp(u, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{}));
    u = max(u, dpp(u, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{}));
    u = max(u, dpp(u, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{}));
    u = max(u, dpp(u, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}));    // rows 0, 2 keep their own (bound_ctrl off: old value 0)
    u = max(u, dpp(u, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}));
    return __int_as_float(__builtin_amdgcn_readlane((int)u, 63));
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// All LDS traffic of one wave is ordered by the hardware; this stops the
// compiler from moving LDS accesses across a cross-lane hand-off and waits for
// outstanding DS operations.  Waves of a workgroup never exchange data, so no
// s_barrier is involved (waves run different trip counts).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// perm position p -> actual factor column
template <int KB>
__device__ __forceinline__ int perm_to_col(int p) { return KB * (p & 15) + (p >> 4); }

// Load the KB contiguous floats of lane c (16-byte aligned when KB % 4 == 0).
template <int KB>
__device__ __forceinline__ void load_frow(const float* __restrict__ p, float (&f)[KB]) {
    if constexpr (KB % 4 == 0) {
#pragma unroll
        for (int j = 0; j < KB / 4; ++j) {
            f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * j);
            f[4 * j] = v.x; f[4 * j + 1] = v.y; f[4 * j + 2] = v.z; f[4 * j + 3] = v.w;
        }
    } else if constexpr (KB % 2 == 0) {
#pragma unroll
        for (int j = 0; j < KB / 2; ++j) {
            f32x2 v = *reinterpret_cast<const f32x2*>(p + 2 * j);
            f[2 * j] = v.x; f[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < KB; ++j) f[j] = p[j];
    }
}

template <int KB>
struct KCfg {
    static constexpr int KP = 16 * KB;
    static constexpr int NACC = KB * (KB + 1) / 2;    // lower 16x16 blocks of the Gram / trailing matrix
    static constexpr int NR = (KP + 63) / 64;         // matrix rows owned per lane
    // L in LDS: block column J holds rows [16J, KP) x 16 columns, row stride 16 floats, the four
    // 4-float groups of a row XOR-swizzled with (row >> 2) & 3 so that b128 row reads by 16
    // consecutive lanes and b128 operand reads by lanes (c, q) are bank-conflict free.
    static constexpr int lcol_off(int J) { return 16 * (J * KP - 8 * J * (J - 1)); }
    __device__ static __forceinline__ int lcol_off_rt(int J) { return 16 * (J * KP - 8 * J * (J - 1)); }
    static constexpr int LDS_FLOATS = lcol_off(KB);   // 10 KB at k = 64
    static_assert(lcol_off(KB) == NACC * 256, "the L image and a full accumulator set have the same size");
    // (Skewing block column J by 16 J dwords makes the transposed solve's column reads - lanes q and q + 1 of a
    // 32-lane group on the same 16 banks today, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.26 in the U-step -
    // conflict-free; measured: no change in time, 6.87 / 6.92 vs 6.92 / 6.84 ms.  The LDS pipe is ~30 % busy and
    // a 2-way conflict on ds_write_b32 / ds_read_b32 hides behind the instruction's own issue cycles.)
    // waves per workgroup / minimum waves per SIMD asked of the register allocator
    // ONE wave per workgroup (round 3; rounds 1-2: 4 / 2 / 1 by model width).  The waves of a workgroup never
    // exchange anything, so a larger workgroup only coarsens the granularity at which the dispatcher refills a CU: all
    // waves of a workgroup must have finished before its registers and LDS go to the next one.  cfg 4, same box:
    // U-step 6.17 -> 6.02 ms, V-step 4.51 -> 4.49 ms, bit-identical (profiles/r03_ab_waves_per_workgroup.txt).
#ifndef ALS_WPW
#define ALS_WPW 1
#endif
    static constexpr int WPW = ALS_WPW;
#ifndef ALS_MINW_LE4
#define ALS_MINW_LE4 3      // (tunable of development builds; 4 = 128 registers spills in K1)
#endif
    static constexpr int MINW = (KB <= 4) ? ALS_MINW_LE4 : (KB <= 6 ? 2 : 1);
    static constexpr int SLOT_ITEMS = NACC * 4 + 2 * KB + 2;  // per-lane floats of a partial
    // gather steps (4 ratings each) staged in registers at a time
    static constexpr int GS = (KB <= 8) ? 8 : 4;
};

// ---------------------------------------------------------------------------
// triangular solves  (L L^T) x = b  for one wave
// ---------------------------------------------------------------------------
// KP <= 64, everything in registers.  Lane i holds column i of the symmetric
// completion of L:  a[p] = L[i][p] (p < i),  a[p] = L[p][i] (p > i), and
// di = 1/L[i][i].  rb is b[i] on entry; returns x[i].  Both sweeps keep the
// right-hand side pre-scaled by di so that each of the 2*KP dependent steps is
// one v_readlane + one v_fma.
template <int KP>
__device__ __forceinline__ float solve_regs(const float (&a)[KP], float di, float rb, int lane,
                                            float* y_out = nullptr) {
    // The lane masks are made by the SCALAR unit inside the call: a shift of all-ones by (step + an opaque zero).
    // Written as `lane > j` - or as literal masks - the 2 KP mask values are loop invariants of the caller's item loop:
    // the compiler hoists them and, out of SGPRs, spills them into VGPR lanes - two extra v_readlane per step.
    int zero;
    asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
    const unsigned long long ones = ~0ull;
    float rs = rb * di;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const float yj = readlane_f(rs, j);
        const unsigned long long mask = (j >= 63) ? 0ull : (ones << (j + 1 + zero));       // lanes > j
        const float cf = __builtin_amdgcn_inverse_ballot_w64(mask) ? a[j] * di : 0.f;
        rs = fmaf(-cf, yj, rs);
    }
    if (y_out) *y_out = rs;   // lane j holds y_j of L y = b
    rs *= di;   // rescale for the transposed sweep
#pragma unroll
    for (int i = KP - 1; i >= 0; --i) {
        const float xi = readlane_f(rs, i);
        const unsigned long long mask = (i <= 0) ? 0ull : ~(ones << (i + zero));           // lanes < i
        const float cf = __builtin_amdgcn_inverse_ballot_w64(mask) ? a[i] * di : 0.f;
        rs = fmaf(-cf, xi, rs);
    }
    return rs;
}

// any KP, coefficients streamed from the factor image in global memory (the symmetric completion
// M of L written by als_row_solve: M[p][i] = L[i][p] for p < i, L[p][i] for p > i, 1/L[i][i] on
// the diagonal).  Lane (+64 rr) owns unknown i = lane + 64 rr and needs exactly column i of M,
// i.e. M[p*KP + i] - coalesced across lanes for every p.  16 steps are loaded at a time, the next
// block is requested before the current one is consumed, so registers stay at 2*16*NR for any k.
// rb enters as b, leaves as x; y_out (optional) receives the forward-solved vector.
template <int KB>
__device__ __forceinline__ void solve_stream_plain(const float* __restrict__ M, float (&rb)[KCfg<KB>::NR], int lane,
                                             float* y_out = nullptr) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR;
    int ic[NR];
    float di[NR], rs[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        ic[rr] = min(lane + 64 * rr, KP - 1);
        di[rr] = M[ic[rr] * KP + ic[rr]];
        rs[rr] = rb[rr] * di[rr];
    }
    float cur[NR][16], nxt[NR][16];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int t = 0; t < 16; ++t) cur[rr][t] = M[t * KP + ic[rr]];
    // L y = b, blocks ascending
#pragma unroll
    for (int pb = 0; pb < KB; ++pb) {
        if (pb + 1 < KB) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int t = 0; t < 16; ++t) nxt[rr][t] = M[(16 * (pb + 1) + t) * KP + ic[rr]];
        } else {    // first block of the transposed sweep
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int t = 0; t < 16; ++t) nxt[rr][t] = M[(16 * (KB - 1) + t) * KP + ic[rr]];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = 16 * pb + t;
            const float yj = readlane_f(rs[j >> 6], j & 63);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const float cf = (lane + 64 * rr > j) ? cur[rr][t] * di[rr] : 0.f;
                rs[rr] = fmaf(-cf, yj, rs[rr]);
            }
        }
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
#pragma unroll
            for (int t = 0; t < 16; ++t) cur[rr][t] = nxt[rr][t];
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (y_out) y_out[rr] = rs[rr];      // lane holds y_i
        rs[rr] *= di[rr];
    }
    // L^T x = y, blocks descending (cur already holds the last block)
#pragma unroll
    for (int pb = KB - 1; pb >= 0; --pb) {
        if (pb > 0) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int t = 0; t < 16; ++t) nxt[rr][t] = M[(16 * (pb - 1) + t) * KP + ic[rr]];
        }
#pragma unroll
        for (int t = 15; t >= 0; --t) {
            const int j = 16 * pb + t;
            const float xj = readlane_f(rs[j >> 6], j & 63);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                const float cf = (lane + 64 * rr < j) ? cur[rr][t] * di[rr] : 0.f;
                rs[rr] = fmaf(-cf, xj, rs[rr]);
            }
        }
        if (pb > 0) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int t = 0; t < 16; ++t) cur[rr][t] = nxt[rr][t];
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) rb[rr] = rs[rr];
}

// k <= 128 (round 3): every step index is a compile-time constant in the unrolled code, so a 64-lane group that a
// step cannot touch costs nothing, a group that it touches as a whole needs no mask, and the pivot's own group takes a
// LITERAL lane mask (the compares used to be hoisted out of the item loop and their SGPR pairs spilled into VGPR lanes:
// two extra v_readlane per step); the next block's loads are pinned in front of the current block's chain.
template <int KB>
__device__ __forceinline__ void solve_stream(const float* __restrict__ M, float (&rb)[KCfg<KB>::NR], int lane,
                                             float* y_out = nullptr) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR;
    if constexpr (NR > 2) {
        solve_stream_plain<KB>(M, rb, lane, y_out);          // (three groups: see lds_fwd_group_plain)
        return;
    } else {
    int ic[NR];
    float di[NR], rs[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        ic[rr] = min(lane + 64 * rr, KP - 1);
        di[rr] = M[ic[rr] * KP + ic[rr]];
        rs[rr] = rb[rr] * di[rr];
    }
    float cur[NR][16], nxt[NR][16];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr)
#pragma unroll
        for (int t = 0; t < 16; ++t) cur[rr][t] = M[t * KP + ic[rr]];
    // L y = b, blocks ascending
#pragma unroll
    for (int pb = 0; pb < KB; ++pb) {
        const int nb = (pb + 1 < KB) ? pb + 1 : KB - 1;     // (last: first block of the transposed sweep)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
#pragma unroll
            for (int t = 0; t < 16; ++t) nxt[rr][t] = M[(16 * nb + t) * KP + ic[rr]];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int j = 16 * pb + t, jb = j >> 6, t6 = j & 63;
            const float yj = readlane_f(rs[jb], t6);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if (rr < jb) continue;                       // that group's unknowns are final
                float cf = cur[rr][t] * di[rr];
                if (rr == jb) {
                    const unsigned long long mask = (t6 >= 63) ? 0ull : (~0ull << (t6 + 1));     // lanes > t6
                    cf = __builtin_amdgcn_inverse_ballot_w64(mask) ? cf : 0.f;
                }
                rs[rr] = fmaf(-cf, yj, rs[rr]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
#pragma unroll
            for (int t = 0; t < 16; ++t) cur[rr][t] = nxt[rr][t];
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        if (y_out) y_out[rr] = rs[rr];      // lane holds y_i
        rs[rr] *= di[rr];
    }
    // L^T x = y, blocks descending (cur already holds the last block)
#pragma unroll
    for (int pb = KB - 1; pb >= 0; --pb) {
        if (pb > 0) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int t = 0; t < 16; ++t) nxt[rr][t] = M[(16 * (pb - 1) + t) * KP + ic[rr]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 15; t >= 0; --t) {
            const int j = 16 * pb + t, jb = j >> 6, t6 = j & 63;
            const float xj = readlane_f(rs[jb], t6);
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                if (rr > jb) continue;                       // rows below the pivot's group are not touched
                float cf = cur[rr][t] * di[rr];
                if (rr == jb) {
                    const unsigned long long mask = (t6 <= 0) ? 0ull : ((1ull << t6) - 1ull);     // lanes < t6
                    cf = __builtin_amdgcn_inverse_ballot_w64(mask) ? cf : 0.f;
                }
                rs[rr] = fmaf(-cf, xj, rs[rr]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (pb > 0) {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
#pragma unroll
                for (int t = 0; t < 16; ++t) cur[rr][t] = nxt[rr][t];
        }
    }
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) rb[rr] = rs[rr];
    }
}

// any KP, L in LDS: Al[j*LD + i] = L[i][j] (i > j), dinv[j] = 1/L[j][j].
// vec[] (LDS, perm space) holds b on entry and x on exit.
//
// Both substitutions run in blocks of 16 steps whose LDS operands are fetched one block ahead (the reads
// do not depend on the running right-hand side, only the FMA chain does).  The running vector is kept
// unscaled: lane j's entry is final once step j has passed, y = rb * dinv is applied at the hand-over and
// at the end, which removes the per-step "lane == j" select.  JB / IB (the 64-lane group of the pivot)
// is a template parameter, so groups that a step cannot touch cost no instruction and only the pivot's
// own group needs the i > j (j < ii) mask.
// k > 128 (three 64-lane groups): the loop form of rounds 1-2.  The unrolled form below is WRONG there inside the
// persistent dataflow kernel (k = 130 ... 160: fits differ in the 4th digit; the same functions are right in the per-level
// kernel and at k <= 128 - 395 VGPRs + 139 AGPRs in that kernel, not tracked down), so those widths keep this one.
template <int KB, int LD, int JB>
__device__ __forceinline__ void lds_fwd_group_plain(const float* __restrict__ Al, const float (&di)[KCfg<KB>::NR],
                                              float (&rb)[KCfg<KB>::NR], const int (&ci)[KCfg<KB>::NR],
                                              float (&cur)[16][KCfg<KB>::NR], int lane) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, SB = 16;
    constexpr int JEND = (64 * JB + 64 < KP) ? 64 * JB + 64 : KP;
    float nxt[SB][NR];
    for (int j0 = 64 * JB; j0 < JEND; j0 += SB) {
        if (j0 + SB < KP) {
#pragma unroll
            for (int u = 0; u < SB; ++u)
#pragma unroll
                for (int rr = JB; rr < NR; ++rr) nxt[u][rr] = Al[(j0 + SB + u) * LD + ci[rr]];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int j = j0 + u;
            const float yj = readlane_f(rb[JB] * di[JB], j & 63);
            rb[JB] = fmaf((lane + 64 * JB > j) ? -cur[u][JB] : 0.f, yj, rb[JB]);
#pragma unroll
            for (int rr = JB + 1; rr < NR; ++rr) rb[rr] = fmaf(-cur[u][rr], yj, rb[rr]);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int rr = JB; rr < NR; ++rr) cur[u][rr] = nxt[u][rr];
    }
    if constexpr (JB + 1 < NR) lds_fwd_group_plain<KB, LD, JB + 1>(Al, di, rb, ci, cur, lane);
}

template <int KB, int LD, int IB>
__device__ __forceinline__ void lds_bwd_group_plain(const float* __restrict__ Al, const float (&di)[KCfg<KB>::NR],
                                              float (&rb)[KCfg<KB>::NR], const int (&ci)[KCfg<KB>::NR],
                                              float (&cur)[16][KCfg<KB>::NR], int lane) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, SB = 16;
    constexpr int ITOP = ((64 * IB + 64 < KP) ? 64 * IB + 64 : KP) - 1;
    float nxt[SB][NR];
    for (int i0 = ITOP; i0 >= 64 * IB; i0 -= SB) {
        if (i0 - SB >= 0) {
#pragma unroll
            for (int u = 0; u < SB; ++u)
#pragma unroll
                for (int rr = 0; rr <= IB; ++rr) nxt[u][rr] = Al[ci[rr] * LD + (i0 - SB - u)];
        }
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int ii = i0 - u;
            const float xi = readlane_f(rb[IB] * di[IB], ii & 63);
            rb[IB] = fmaf((lane + 64 * IB < ii) ? -cur[u][IB] : 0.f, xi, rb[IB]);
#pragma unroll
            for (int rr = 0; rr < IB; ++rr) rb[rr] = fmaf(-cur[u][rr], xi, rb[rr]);
        }
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int rr = 0; rr <= IB; ++rr) cur[u][rr] = nxt[u][rr];
    }
    if constexpr (IB > 0) lds_bwd_group_plain<KB, LD, IB - 1>(Al, di, rb, ci, cur, lane);
}

// Round 3: the step loops are fully unrolled with the lane masks as LITERALS (inverse_ballot of a constant; the
// compiler used to hoist 256 `lane < const` compares out of the item loop and spill their SGPR pairs into VGPR lanes:
// two extra v_readlane per step), and the operand block of the NEXT 16 steps is requested - and pinned there by a
// scheduling barrier - before the current block's dependent chain starts (it used to be re-loaded just in time, an
// LDS round trip inside every step).  k = 128: 71 -> ~45 cycles per step.
template <int KB, int LD, int JB>
__device__ __forceinline__ void lds_fwd_group(const float* __restrict__ Al, const float (&di)[KCfg<KB>::NR],
                                              float (&rb)[KCfg<KB>::NR], const int (&ci)[KCfg<KB>::NR],
                                              float (&cur)[16][KCfg<KB>::NR], int lane) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, SB = 16;
    constexpr int JEND = (64 * JB + 64 < KP) ? 64 * JB + 64 : KP;
    float nxt[SB][NR];
#pragma unroll
    for (int j0 = 64 * JB; j0 < JEND; j0 += SB) {
        if (j0 + SB < KP) {
#pragma unroll
            for (int u = 0; u < SB; ++u)
#pragma unroll
                for (int rr = JB; rr < NR; ++rr) nxt[u][rr] = Al[(j0 + SB + u) * LD + ci[rr]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int t = j0 + u - 64 * JB;                                       // lanes > t of group JB take the update
            const unsigned long long mask = (t >= 63) ? 0ull : (~0ull << (t + 1));
            const float yj = readlane_f(rb[JB] * di[JB], t);
            rb[JB] = fmaf(__builtin_amdgcn_inverse_ballot_w64(mask) ? -cur[u][JB] : 0.f, yj, rb[JB]);
#pragma unroll
            for (int rr = JB + 1; rr < NR; ++rr) rb[rr] = fmaf(-cur[u][rr], yj, rb[rr]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int rr = JB; rr < NR; ++rr) cur[u][rr] = nxt[u][rr];
    }
    if constexpr (JB + 1 < NR) lds_fwd_group<KB, LD, JB + 1>(Al, di, rb, ci, cur, lane);
}

template <int KB, int LD, int IB>
__device__ __forceinline__ void lds_bwd_group(const float* __restrict__ Al, const float (&di)[KCfg<KB>::NR],
                                              float (&rb)[KCfg<KB>::NR], const int (&ci)[KCfg<KB>::NR],
                                              float (&cur)[16][KCfg<KB>::NR], int lane) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, SB = 16;
    constexpr int ITOP = ((64 * IB + 64 < KP) ? 64 * IB + 64 : KP) - 1;
    float nxt[SB][NR];
#pragma unroll
    for (int i0 = ITOP; i0 >= 64 * IB; i0 -= SB) {
        if (i0 - SB >= 0) {
#pragma unroll
            for (int u = 0; u < SB; ++u)
#pragma unroll
                for (int rr = 0; rr <= IB; ++rr) nxt[u][rr] = Al[ci[rr] * LD + (i0 - SB - u)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SB; ++u) {
            const int t = i0 - u - 64 * IB;                                       // lanes < t of group IB take the update
            const unsigned long long mask = (t <= 0) ? 0ull : ((1ull << t) - 1ull);
            const float xi = readlane_f(rb[IB] * di[IB], t);
            rb[IB] = fmaf(__builtin_amdgcn_inverse_ballot_w64(mask) ? -cur[u][IB] : 0.f, xi, rb[IB]);
#pragma unroll
            for (int rr = 0; rr < IB; ++rr) rb[rr] = fmaf(-cur[u][rr], xi, rb[rr]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < SB; ++u)
#pragma unroll
            for (int rr = 0; rr <= IB; ++rr) cur[u][rr] = nxt[u][rr];
    }
    if constexpr (IB > 0) lds_bwd_group<KB, LD, IB - 1>(Al, di, rb, ci, cur, lane);
}

template <int KB, int LD>
__device__ __forceinline__ void solve_lds(const float* __restrict__ Al, const float* __restrict__ dinv,
                                          float* __restrict__ vec, int lane, float* y_out = nullptr) {
    constexpr int KP = KCfg<KB>::KP, NR = KCfg<KB>::NR, SB = 16;
    static_assert(KP % SB == 0, "KP is a multiple of 16");
    float rb[NR], di[NR];
    int ci[NR];
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        ci[rr] = min(i, KP - 1);
        rb[rr] = (i < KP) ? vec[i] : 0.f;
        di[rr] = (i < KP) ? dinv[i] : 0.f;
    }
    float cur[SB][NR];
#pragma unroll
    for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) cur[u][rr] = Al[u * LD + ci[rr]];
    if constexpr (NR <= 2) lds_fwd_group<KB, LD, 0>(Al, di, rb, ci, cur, lane);            // L y = b
    else                   lds_fwd_group_plain<KB, LD, 0>(Al, di, rb, ci, cur, lane);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) rb[rr] *= di[rr];                   // y
    if (y_out) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) y_out[rr] = rb[rr];
    }
    // L^T x = y: step ii needs L[ii][j] = Al[j * LD + ii] for the lanes j < ii
#pragma unroll
    for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) cur[u][rr] = Al[ci[rr] * LD + (KP - 1 - u)];
    if constexpr (NR <= 2) lds_bwd_group<KB, LD, NR - 1>(Al, di, rb, ci, cur, lane);
    else                   lds_bwd_group_plain<KB, LD, NR - 1>(Al, di, rb, ci, cur, lane);
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) rb[rr] *= di[rr];                   // x
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        const int i = lane + 64 * rr;
        if (i < KP) vec[i] = rb[rr];
    }
    wave_lds_sync();
}
