/*
 * als_hip.h - C ABI of the MI355X (gfx950) ALS hot path.
 *
 * The reference (zhukovanadezhda/collaborative-filtering) has no FFI: its hot
 * path is the Python method `ALS.fit` (scripts/als.py:300-529) and
 * `ALS.predict` (scripts/als.py:532-574).  This header is the boundary the
 * build's Python mirror of that class (collaborative-filtering_amd/als.py)
 * calls through ctypes; every entry point names the reference lines it
 * replaces.  See INTEGRATION.md for the binding a maintainer would add.
 *
 * Conventions
 *   - plain C: pointers, sizes, scalars; no C++/torch types.
 *   - every pointer is a DEVICE pointer (HBM) unless it says "host".
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*);
 *     nothing synchronises, allocates or frees.
 *   - return value: 0 ok, negative = argument error (ALS_E_*).
 *   - factor matrices are row-major fp32 with leading dimension
 *     ld = als_padded_k(k) (k rounded up to a multiple of 16); the padding
 *     columns must be zero on input and are written as zero.
 *   - rating matrices are CSR-like: int64 row pointers, int32 indices, fp32
 *     values.  The same entry point serves the user step (CSR) and the item
 *     step (CSC); "row" below means a row of whichever orientation is passed.
 *   - "perm space": inside the solver a factor column c lives at position
 *     perm(c) = 16*(c % KB) + c / KB, KB = ld/16.  Outputs documented as
 *     perm-space use that order; als_perm_index() gives the map.
 */
#ifndef ALS_HIP_H
#define ALS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALS_HIP_VERSION 103
#define ALS_NXCD 8             /* XCDs (private L2s) of the MI355X: als_host_row_tasks deals equal-length tasks over them */

#define ALS_E_BADARG   (-1)
#define ALS_E_BADK     (-2)   /* k outside 1..ALS_MAX_K */
#define ALS_E_LAUNCH   (-3)   /* hipGetLastError() != hipSuccess after a launch */

#define ALS_GRAM_F32    0
#define ALS_GRAM_F16X2  1     /* 2-way fp16 split of the scaled floats, three cross products on v_mfma_f32_16x16x32_f16,
                                 fp32 accumulate: fp32-equivalent (default) */
#define ALS_GRAM_F64    2     /* fp64 Gram (v_mfma_f64_16x16x4_f64), fp64 Cholesky and substitutions: the
                                 reference's arithmetic type; x and the bias are rounded to fp32 on store */

#define ALS_MAX_K 160
#define ALS_SPLIT_CHUNK 4096  /* ratings per task segment (see als_task) */

/* One unit of row-solve work: ratings [seg*ALS_SPLIT_CHUNK, +len) of `row`.
 * slot < 0 : the row fits one segment and is solved in place.
 * slot >= 0: partial normal equations go to workspace slot `slot`; the row is
 *            completed by the matching als_long_row entry.                  */
typedef struct als_task {
    int32_t row;
    int32_t seg;
    int32_t slot;
    int32_t reserved;
} als_task;

/* A row split over `nslots` consecutive workspace slots starting at `slot0`. */
typedef struct als_long_row {
    int32_t row;
    int32_t slot0;
    int32_t nslots;
    int32_t reserved;
} als_long_row;

int als_version(void);
int als_padded_k(int k);                       /* ld for k factors            */
int als_perm_index(int k, int c);              /* perm-space position of col c */
/* bytes of one partial slot / of the whole workspace for `nslots` slots */
int64_t als_partial_slot_bytes(int k);
int64_t als_partial_slot_bytes_f64(int k);     /* slot size when gram_mode == ALS_GRAM_F64 */

/* ---------------------------------------------------------------------------
 * als_row_solve - normal-equation build and solve for a set of rows.
 * Replaces the bodies of the user loop (scripts/als.py:414-433) and the item
 * loop (scripts/als.py:436-466) incl. cholesky_solve (scripts/helpers.py:5-20).
 *
 * For every task row with nnz > 0:
 *     F_t   = F[indices[t]]                       (gathered factor rows)
 *     r_t   = vals[t] - (mu + bias_self[row] + bias_other[indices[t]])
 *     A     = F^T F + (lambda(row) + 1e-10 + diag_extra[row]) I
 *     b     = F^T r + rhs_extra[row]              (rhs_extra: actual col order)
 *     x     = A^{-1} b          (Cholesky; x -> X_out[row])
 *     bias  = sum_t(vals[t] - F_t.x - mu - bias_other[indices[t]])
 *             / (nnz + lambda_bias(row) + 1e-10)  (-> bias_out[row])
 * lambda(row) = lambda_row ? lambda_row[row] : lambda_scalar (same for bias).
 * Rows with nnz == 0 must not appear in the task list (they keep their
 * previous X/bias, as in the reference).
 *
 * Optional outputs
 *   gram_out   [nrows][ld][ld]  F^T F (no lambda), perm space, lower 16x16
 *                               blocks (block row >= block col) written; the
 *                               upper blocks are left untouched.
 *   rhs_out    [nrows][ld]    F^T r (perm space, without rhs_extra),
 *   colsum_out [nrows][ld]    sum_t F_t (perm space),
 *   sumr_out   [nrows]        sum_t(vals[t] - mu - bias_other[indices[t]])
 *   sumr2_out  [nrows]        sum_t(vals[t] - mu - bias_other[indices[t]])^2
 *                             (each written when its pointer is non-NULL)
 *   stat_out   [nrows][2]     solve mode only: (sum d, sum d^2) of the row's residuals after the update
 *   factor-only mode (factor_out != NULL): nothing is solved; instead
 *     factor_out [nrows][ld][ld] symmetric completion of the Cholesky factor
 *                               L (perm space) with 1/L_ii on the diagonal
 *   is written for als_gs_sweep, which also needs rhs_out/colsum_out/sumr_out.
 * status: int32[1], must be 0 on entry (host zeroes it); on a non-SPD pivot
 *   the kernel stores (row + 1) with atomicMax.
 * workspace: als_partial_slot_bytes(k) * (number of slots referenced by
 *   tasks) bytes; may be NULL when no task has slot >= 0.
 * ------------------------------------------------------------------------- */
typedef struct als_row_solve_params {
    int32_t k;
    int32_t ld;                 /* = als_padded_k(k); ld of F, X_out, rhs_extra */
    int64_t nrows;              /* rows in this orientation (bounds checks only) */
    int32_t F_zero_row;         /* index of an all-zero row of F (ratings past the end of a
                                   row are pointed at it instead of being masked); F_zero_row*ld
                                   and every indices[t]*ld must be < 2^31 */
    int32_t reserved0;          /* 0 (profiling builds: phase-ablation flags) */
    int32_t gram_mode;          /* ALS_GRAM_F32: v_mfma_f32_16x16x4_f32 on the gathered floats;
                                   ALS_GRAM_F16X2: every float times S (a power of two, als_factor_scale) is split
                                   into two fp16 terms by rounding to nearest (error <= 2^-23), three cross
                                   products on v_mfma_f32_16x16x32_f16, fp32 accumulate (same accuracy
                                   class, the 16-bit matrix cores run beside the VALU); needs F_scale;
                                   ALS_GRAM_F64: everything in fp64 (row_solve_f64.hip) - for lambda << 1 with
                                   rank-deficient rows, where cond(A) ~ 1/lambda amplifies the fp32 rounding of
                                   the Gram; workspace slots are als_partial_slot_bytes_f64(k) bytes; the
                                   dual-form classes are ignored */
    int32_t ndual_tail;         /* number of TRAILING tasks that are whole rows (slot < 0) of at most 64 ratings
                                   to be solved in the dual form (n x n instead of k x k system, same
                                   solution); honoured in plain solve calls (no by-product outputs, no
                                   rhs_extra / diag_extra, f16x2 Gram).  Pays when the n x n system is much
                                   the smaller one (k > 64: 5x per row at k = 128; at k = 64, n <= 48 it measured
                                   equal to the primal kernel); 0 = never */
    int32_t ndual_mid;          /* number of tasks just BEFORE that tail that are whole rows of 65 ... 96 ratings,
                                   for the dual form at k > 96 (an 80- or 96-size system); 0 = never */
    int32_t F_scale_ready;      /* != 0: F_scale[0 .. 1] already hold als_factor_scale(F) (several calls on one F) */
    const int64_t* indptr;
    const int32_t* indices;
    const float*   vals;
    const float*   F;           /* gathered side factors [ncols][ld] */
    const float*   bias_self;   /* [nrows], values before this step */
    const float*   bias_other;  /* [ncols] */
    const double*  mu;          /* device scalar */
    float          lambda_scalar;
    const float*   lambda_row;          /* nullable */
    float          lambda_bias_scalar;
    const float*   lambda_bias_row;     /* nullable */
    const float*   rhs_extra;           /* nullable [nrows][ld] */
    const float*   diag_extra;          /* nullable [nrows] */
    float*         X_out;               /* [nrows][ld] */
    float*         bias_out;            /* [nrows] (may alias bias_self) */
    float*         gram_out;            /* nullable */
    float*         factor_out;          /* nullable -> factor-only mode */
    float*         rhs_out;
    float*         colsum_out;
    float*         sumr_out;
    float*         sumr2_out;           /* nullable [nrows]: sum_t (vals[t]-mu-bias_other)^2 (factor-only mode) */
    float*         stat_out;            /* nullable [nrows][2]: (sum d, sum d^2) of the row with the NEW x and
                                           bias, d = vals - F_t.x - mu - bias_other - bias_new; lets the caller
                                           skip als_residual_stats when Z == F-side of the other step */
    int32_t*       status;
    const als_task*     tasks;      int64_t ntasks;
    const als_long_row* long_rows;  int64_t nlong;
    void*          workspace;
    /* Conditioning-driven precision (solve_dtype "auto"): with cond_limit > 0 every row's condition estimate is taken
     * from what the fp32 factorisation leaves in registers - kappa = max( (max_i L_ii / min_i L_ii)^2,
     * (trace(G) / rank + lambda) / min_i L_ii^2, and for rows of fewer than 4 k ratings (mean eigenvalue) / lambda ),
     * three lower bounds of cond_2(A).  A row with kappa > cond_limit (or whose fp32 factorisation breaks down, or
     * whose closed-form statistics would cancel) writes none of its results - its row id is appended to redo_rows
     * (order irrelevant) and the call then redoes exactly those rows in fp64 (Gram, factorisation, substitutions:
     * the kernel of ALS_GRAM_F64 on the whole row).  The relative error of an fp32 row is then bounded by about
     * cond_limit * 3e-7 (the fp32 rounding of its Gram).  f32 / f16x2 modes only. */
    float          cond_limit;          /* 0 = off */
    int32_t        byproducts_f64;      /* ALS_GRAM_F64 only, != 0: gram_out, rhs_out, colsum_out, sumr_out, sumr2_out are
                                           arrays of DOUBLE (same shapes) - for the W-step / item statistics in fp64
                                           (als_w_params::f64, als_item_stats_f64) - and factor_out an array of double
                                           (als_gs_sweep_params::factor_f64) */
    int32_t*       redo_count;          /* device int32[1]; the call resets it; afterwards: number of rows redone */
    int32_t*       redo_rows;           /* device int32[rows of this orientation] (only the first redo_count are used) */
    float*         cond_out;            /* nullable [nrows]: kappa of every row solved in fp32 (diagnostics) */
    float*         F_scale;             /* ALS_GRAM_F16X2: ALS_FSCALE_FLOATS floats of device memory, zero before the FIRST
                                           use (later calls leave words 2, 3 zero); the call writes {S, 1 / S^2} of F to
                                           words 0, 1 unless F_scale_ready; must not be shared by concurrent calls */
    void*          F_planes;            /* nullable; ALS_GRAM_F16X2 at k = 49 ... 64 or 113 ... 128: scratch of (F_zero_row + 1) * ld 32-bit
                                           words (F_zero_row must be the LAST row of F).  When given, the call first
                                           writes the two fp16 terms of every element of F into it (same layout as F)
                                           and the Gram is built from those pre-split operands, right-hand side and
                                           column sums on the matrix cores too - same Gram bit for bit, far fewer
                                           vector instructions per rating.  Pays where F is small and the launch is
                                           bound by vector issue (the U-step); costs one pass over F.  Must not be
                                           shared by concurrent calls */
} als_row_solve_params;

int als_row_solve(const als_row_solve_params* p, void* stream);

/* Operand scale of the f16x2 Gram for a factor matrix F of `nfloats` floats (a multiple of 4, F 16-byte aligned):
 * scale[0] = S = 2^j with S max|F| in [2^14, 2^15) (clamped to 2^-60 ... 2^60), scale[1] = 1 / S^2; scale[2 .. 3]
 * are working words, zero on entry and on exit.  als_row_solve calls this itself unless F_scale_ready. */
#define ALS_FSCALE_FLOATS 4
int als_factor_scale(const float* F, int64_t nfloats, float* scale, void* stream);

/* ---------------------------------------------------------------------------
 * als_gs_sweep - one level of the Gauss-Seidel Laplacian sweep.
 * Replaces the graph part of the item loop (scripts/als.py:453-461,464-466):
 * for each listed item i (all items of one dependency level):
 *     b = rhs[i] + alpha * sum_j S_ij V[j]        (V read live)
 *     V[i] = (L L^T)^{-1} b ;  b_i[i] = (sumr[i] - colsum[i].V[i]) / denom
 * factor/rhs/colsum/sumr come from als_row_solve in factor-only mode.
 * ------------------------------------------------------------------------- */
typedef struct als_gs_sweep_params {
    int32_t k, ld;
    const int32_t* items;  int64_t nitems;   /* items of this level */
    const int64_t* S_ptr;  const int32_t* S_idx;  const float* S_val;
    float alpha;
    const float* factor;  const float* rhs;  const float* colsum;
    const float* sumr;
    const int64_t* indptr;               /* CSC pointers (nnz per item) */
    float lambda_bias_scalar;  const float* lambda_bias_row;
    float* V;                            /* [n][ld], updated in place */
    float* bias;                         /* [n] */
    /* optional fused statistics (all three or none): */
    const float* sumr2;                  /* [n] from als_row_solve (sumr2_out) */
    const float* lambda_eff;             /* [n] total diagonal shift used in the factor (lambda+1e-10+diag_extra) */
    float* stat_out;                     /* [n][2]: (sum d, sum d^2) with the new V[i], bias[i] */
    int32_t f64;                         /* != 0: factor, rhs, colsum, sumr, sumr2 are arrays of DOUBLE (als_row_solve with
                                            ALS_GRAM_F64 and byproducts_f64); neighbour sums, substitutions, bias and
                                            statistics then run in fp64 (als_gs_sweep / als_gs_sweep_levels only) */
    int32_t reserved;
} als_gs_sweep_params;

int als_gs_sweep(const als_gs_sweep_params* p, void* stream);

/* Whole sweep as ONE persistent launch without level barriers.  p->items lists ALL swept
 * items in (level, id) order (p->nitems of them).  S_idx_wait is p->S_idx with the sign bit set on
 * every edge (i -> j) that item i must wait for (j < i and j swept in this call).  publish: scratch
 * [nrows][ld] floats (nrows = rows of p->V); the call resets it and the kernel hands solved rows from
 * producer to consumer through it (each word doubles as its own "ready" flag).  nondep: optional scratch
 * [nrows][ld]: when given, the neighbour sums that do not depend on the sweep (edges without the wait flag) are
 * formed for all items by one parallel launch before the persistent one - same sums, same order, off the
 * dependency chain.  err: int32[1], the caller
 * zeroes it once; set to 1 when a dependency wait exceeded its bound (2^27 shader-clock ticks, ~60 ms: the launch was not resident as a
 * whole because something else held compute units).  Results are then invalid and the caller should redo the
 * sweep - from the state before it - with als_gs_sweep_levels, which has no residency requirement. */
int als_gs_sweep_dataflow(const als_gs_sweep_params* p, const int32_t* S_idx_wait, float* publish,
                          int64_t nrows, float* nondep, int32_t* err, void* stream);

/* Whole sweep in one call: level l covers p->items[level_offsets[l] .. level_offsets[l+1]) (host
 * array of nlevels+1 offsets into the device array p->items; p->nitems is ignored).  One launch per
 * level on `stream`; levels are dependent, so stream order is the synchronisation. */
int als_gs_sweep_levels(const als_gs_sweep_params* p, const int64_t* level_offsets /* host */,
                        int64_t nlevels, void* stream);

/* ---------------------------------------------------------------------------
 * als_residual_stats - replaces scripts/als.py:503-512: one pass over the
 * ratings (CSR) computing sum(d) and sum(d^2), d = r - (U_u.Z_i + b_u + b_i
 * + mu_old), in fp64.  out[0]=sum d, out[1]=sum d^2 (device doubles).
 * partials: scratch of 2*ceil(ntasks/4) doubles.
 * ------------------------------------------------------------------------- */
int als_residual_stats(int k, int ld, const int64_t* indptr, const int32_t* indices,
                       const float* vals, const float* U, const float* Z,
                       const float* b_u, const float* b_i, const double* mu,
                       const als_task* tasks, int64_t ntasks,
                       double* partials, double* out, void* stream);

/* ---------------------------------------------------------------------------
 * als_w_normal_equations - normal equations of the feature-projection step,
 * replaces scripts/als.py:469-499 (the N_obs x (d k) design matrix is never formed).
 *   phase 0: per-item vectors h_{f,i} = g_i + G_i xw_{f,i} for all features (H, perm space),
 *            from the V-step by-products gram/rhs/colsum (als_row_solve), the new and old item
 *            bias, V and the OLD projections W (Jacobi across features, as the reference).
 *   phase 1: for ONE feature (columns feat_col0 .. +feat_d of X): A_out [(d*k)^2] =
 *            sum_i (x_i x_i^T) (x) G_i and B_out [d*k] = sum_i x_i (x) h_{f,i}, fp64, index
 *            a*k + c with c the factor column in storage order (the reference's vec layout,
 *            scripts/als.py:494); A_out is bitwise symmetric; the caller adds lambda.
 * Items [item_begin, item_end) only (a rank's shard); A/B are then all-reduced by the caller.
 * ------------------------------------------------------------------------- */
typedef struct als_w_params {
    int32_t k, ld;
    int32_t phase;                 /* 0: item vectors, 1: accumulate one feature */
    int32_t nfeat;                 /* phase 0: number of features (<= 8) */
    int64_t item_begin, item_end;
    const float* gram;             /* [n][ld][ld] lower blocks (als_row_solve gram_out) */
    const float* rhs;              /* [n][ld] perm  (rhs_out)    */
    const float* colsum;           /* [n][ld] perm  (colsum_out) */
    const float* V;                /* [n][ld] */
    const float* b_new;            /* [n] item bias after the V-step  */
    const float* b_old;            /* [n] item bias before the V-step */
    int32_t D;                     /* total feature columns of X */
    int32_t f64;                   /* != 0: gram, rhs, colsum, H are arrays of double (als_row_solve byproducts_f64) and W
                                      is the fp64 projection matrix [D][k] (row stride k, storage column order) */
    const float* X;                /* [n][D] all features side by side */
    const int32_t* feat_off;       /* device int32 [nfeat+1] column offsets (phase 0) */
    const float* W;                /* [D][ld] old projections, storage column order (phase 0) */
    float* H;                      /* [nfeat][nrows_h][ld] (phase 0 writes, phase 1 reads) */
    int64_t nrows_h;
    int32_t feat_index, feat_col0, feat_d, nchunks;   /* phase 1 */
    double* partA;                 /* scratch: d(d+1)/2 * nchunks * (ld/16)(ld/16+1)/2 * 256 doubles */
    double* partB;                 /* scratch: d * nchunks * ld doubles */
    double* A_out;                 /* [(d*k)][(d*k)] */
    double* B_out;                 /* [d*k] */
} als_w_params;

int als_w_normal_equations(const als_w_params* p, void* stream);

/* -------------------------------------------------------------------------
 * als_spd_solve_f64: x = (A + diag_add I)^-1 b by a dense fp64 Cholesky factorisation -
 * the reference's `cholesky_solve(A, b)` on the W-step's (d k) x (d k) normal equations
 * (scripts/als.py:497-500; scripts/helpers.py cholesky_solve).
 * A: [N][lda] row-major fp64, symmetric (both triangles valid), not modified.  b, x: [N] (may alias).
 * workspace: als_spd_solve_workspace_bytes(N) bytes of device memory (0 = N out of range).
 * status (device int32): 0 ok; p > 0: pivot p-1 was not positive (not SPD, x is then meaningless).
 * Enqueues 2 ceil(N/64) + 2 small kernels on `stream`.
 * ------------------------------------------------------------------------- */
#define ALS_SPD_MAX_N 8128
size_t als_spd_solve_workspace_bytes(int64_t N);
int als_spd_solve_f64(int64_t N, const double* A, int64_t lda, const double* b, double diag_add,
                      double* x, void* workspace, int32_t* status, void* stream);

/* Closed-form residual sums per item when Z != V (fits with features), replacing the pass over the
 * ratings of scripts/als.py:505-511: stat_out[i] = (sum d, sum d^2), d = r - mu - b_u - b_new[i] - u.Z[i],
 * from the als_row_solve by-products of this iteration's V-step (gram_out, rhs_out, colsum_out, sumr_out,
 * sumr2_out - rhs formed with b_old) over items [item_begin, item_end).  Z: [n][ld] storage order. */
int als_item_stats(int k, int ld, int64_t item_begin, int64_t item_end, const float* gram,
                   const float* rhs, const float* colsum, const float* sumr, const float* sumr2,
                   const int64_t* indptr, const float* Z, const float* b_new, const float* b_old,
                   float* stat_out, void* stream);

/* The same from fp64 by-products (als_row_solve with gram_mode ALS_GRAM_F64 and byproducts_f64). */
int als_item_stats_f64(int k, int ld, int64_t item_begin, int64_t item_end, const double* gram,
                       const double* rhs, const double* colsum, const double* sumr, const double* sumr2,
                       const int64_t* indptr, const float* Z, const float* b_new, const float* b_old,
                       float* stat_out, void* stream);

/* out[0] = sum_i x[2i], out[1] = sum_i x[2i+1] in fp64 (reduction of stat_out).
 * partials: scratch of 2*als_sumsq_partials() doubles. */
int als_sum_pairs(const float* x, int64_t npairs, double* partials, double* out, void* stream);

/* sum of squares of a float array in fp64 (scripts/als.py:514-517 norms).
 * partials: scratch of als_sumsq_partials() doubles. out: device double. */
int als_sumsq_partials(void);
int als_sumsq(const float* x, int64_t n, double* partials, double* out, void* stream);

/* The history row of an iteration in two launches (scripts/als.py:505-517): stats = (sum d, sum d^2) of the residuals
 * before the mu update (als_sum_pairs / als_residual_stats), nnz ratings.  mu (device double) += sum d / nnz;
 * row[6] = {train RMSE with the new mu, |U|_F, |V|_F, |b_u|, |b_i|, mu}.  The arrays must be 16-byte aligned.
 * partials: scratch of 4 * als_sumsq_partials() doubles. */
int als_history_row(const float* U, int64_t nU, const float* V, int64_t nV, const float* b_u, int64_t nbu,
                    const float* b_i, int64_t nbi, const double* stats, int64_t nnz, double* mu,
                    double* partials, double* row, void* stream);

/* Z = V + X W  (scripts/als.py:262-281); X [n][D] fp32 (all features
 * concatenated column-wise), W [D][ld] fp32.  D == 0 copies V. */
int als_compose_z(int64_t n, int ld, int D, const float* V, const float* X,
                  const float* W, float* Z, void* stream);

/* predictions at (u,i) pairs: out[t] = U_u.Z_i + mu + b_u + b_i
 * (the only way callers read predict(): scripts/tune_params.py:165-166) */
int als_predict_at(int k, int ld, int64_t npairs, const int32_t* us, const int32_t* is,
                   const float* U, const float* Z, const float* b_u, const float* b_i,
                   const double* mu, float* out, void* stream);

/* dense completion R_hat[m][n] = U Z^T + mu + b_u + b_i (scripts/als.py:574) */
int als_predict_dense(int k, int ld, int64_t m, int64_t n, const float* U,
                      const float* Z, const float* b_u, const float* b_i,
                      const double* mu, float* out, void* stream);

/* ---------------------------------------------------------------------------
 * Item-item similarity graph on the device (scripts/als.py:224-240 without the n x n matrix).
 * als_topk_similarity: XT is the row-normalised feature matrix Xn [n][d] (fp32) re-laid as [nsteps][n_pad][4]
 *   (feature dims in groups of 4, zero padded: nsteps in {1, 2, 4, 5, 8, 16}, i.e. d <= 64; n_pad a multiple of
 *   16, padding rows zero).  For every row i the topk (<= ALS_TOPK_MAX) largest entries of row i of Xn Xn^T with
 *   its diagonal set to 0 (the diagonal entry competes like any other, as in the reference; zeros are never
 *   edges), ordered by (similarity descending, j ascending) - among equal similarities the LOWEST column indices win (the
 *   reference's argpartition keeps an implementation-defined subset of such ties):
 *   top_val / top_idx [n][topk] (unused slots 0 / -1), top_cnt [n] = min(topk, n).
 * als_graph_classify: S = max(S, S^T) on those lists: own[i][t] = 1 when entry t of row i is an edge of the
 *   symmetric graph, mirror[i][t] = 1 when its transpose (j, i, s) must be added because j's list does not
 *   contain i (one-sided positive entries; one-sided negative ones and zeros are not edges).
 * ------------------------------------------------------------------------- */
#define ALS_TOPK_MAX 128
int als_topk_similarity(int64_t n, int64_t n_pad, int nsteps, const float* XT, int topk, float* top_val,
                        int32_t* top_idx, int32_t* top_cnt, void* stream);
int als_graph_classify(int64_t n, int topk, const float* top_val, const int32_t* top_idx,
                       const int32_t* top_cnt, uint8_t* own, uint8_t* mirror, void* stream);

/* ---------------------------------------------------------------------------
 * Item-feature normalisation (scripts/prepare_features.py:95-124, 131-201) of a float64 [n][d] matrix X (device):
 * method 0 none (cast), 1 row_l1, 2 row_l2, 3 col_zscore, 4 col_minmax; out: float32 [n][d].  Sums run in numpy's
 * order, so out is bitwise the reference's result.  colwork: 2*d doubles (methods 3, 4).  status (device int32,
 * zeroed by the caller): bit 0 set when X holds a NaN / inf (the reference raises ValueError; impute first).
 * ------------------------------------------------------------------------- */
int als_normalize_features(int64_t n, int d, const double* X, int method, double eps, float* out,
                           double* colwork, int32_t* status, void* stream);
/* Median imputation in place (scripts/prepare_features.py:82-92): every NaN / +-inf of X [n][d] (device, float64)
 * becomes the median of its column's finite entries (mean of the two middle ones for an even count, 0 for a column
 * without finite entries).  Exact order statistics by radix select; work: 3*d doubles. */
int als_impute_col_median(int64_t n, int d, double* X, double* work, void* stream);

/* ---------------------------------------------------------------------------
 * Host-side set-up passes (HOST pointers, synchronous, no GPU involved).  They replace the reference's
 * per-fit index-list construction (scripts/als.py:332-340) and prepare the inputs of the entry points above;
 * the named caller times fit + predict together (scripts/evaluate_models.py:245-255), so this is inside its
 * measured region.
 * ------------------------------------------------------------------------- */
/* COO ratings -> CSR by user (uptr [m+1], uidx / uval [N]) and CSC by item (iptr [n+1], iidx / ival [N]),
 * indices ascending inside every row / column.  Returns -10 for an index outside the m x n shape, -11 for a
 * duplicate (user, item) pair. */
int als_host_coo_to_sides(int64_t m, int64_t n, int64_t N, const int64_t* rows, const int64_t* cols,
                          const float* vals, int64_t* uptr, int32_t* uidx, float* uval,
                          int64_t* iptr, int32_t* iidx, float* ival);

/* Task list of als_row_solve for rows [row_begin, row_end) with at least one rating: rows longer than `chunk`
 * (= ALS_SPLIT_CHUNK) are split, tasks are ordered longest first, whole rows of at most `mid_len` / `dual_len`
 * ratings go last (ndual_mid / ndual_tail of als_row_solve_params).  counts[6] = {ntasks, nlong, nslots,
 * ratings covered, ndual, nmid}; call once with tasks == NULL to size the arrays. */
int als_host_row_tasks(const int64_t* indptr, int64_t row_begin, int64_t row_end, int32_t chunk,
                       int32_t dual_len, int32_t mid_len, als_task* tasks, als_long_row* long_rows,
                       int64_t* counts);

/* Dependency levels of the index-ordered Gauss-Seidel sweep over the active items of [begin, end):
 * level[n] (-1 = not swept), items in (level, id) order, offsets[nlevels + 1] (n + 1 entries provided),
 * wait (nullable, [nnz(S)]) = the S_idx_wait input of als_gs_sweep_dataflow.  out[2] = {nitems, nlevels}. */
int als_host_level_schedule(int64_t n, const int64_t* S_ptr, const int32_t* S_idx, const uint8_t* active,
                            int64_t begin, int64_t end, int64_t* level, int32_t* items,
                            int64_t* offsets, int32_t* wait, int64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* ALS_HIP_H */
