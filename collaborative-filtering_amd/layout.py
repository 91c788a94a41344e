"""Host-side preparation of the HBM-resident data structures.

Everything here runs once per `fit`, before the timed loop: COO -> CSR/CSC,
the row-task lists consumed by `als_row_solve` (include/als_hip.h), the
perm-space index map, and the Gauss-Seidel level schedule of the Laplacian
sweep.  It restates the reference's adjacency construction
(scripts/als.py:332-340) on sparse input; nothing is dense m x n.

The numpy functions are the executable specification; the `*_native`
variants run the same passes in the C-ABI library (csrc/host_setup.cpp,
single O(nnz) loops off the interpreter) and are what the engine uses with
the HIP backend - the reference's caller times fit + predict together, so
set-up is inside its measured region.  tests/test_layout_properties.py holds
the two to identical outputs.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

DUAL_MAX_LEN = 64        # rows this short may be solved in the dual form (csrc/row_solve.hip, k_row_dual)
DUAL_MID_LEN = 96        # ... and, above k = 96, rows up to this length (k_row_dual_mid)
SPLIT_CHUNK = 4096          # == ALS_SPLIT_CHUNK in include/als_hip.h
NXCD = 8                    # == ALS_NXCD: XCDs (private L2s) of the MI355X


def padded_k(k: int) -> int:
    if k < 1 or k > 160:
        raise ValueError(f"n_factors={k} outside the supported range 1..160")
    return 16 * ((k + 15) // 16)


def perm_of_col(k: int) -> np.ndarray:
    """perm-space position of every storage column (length ld)."""
    ld = padded_k(k)
    kb = ld // 16
    c = np.arange(ld)
    return (16 * (c % kb) + c // kb).astype(np.int64)


@dataclass
class SparseSide:
    """One orientation of the rating matrix (CSR by user or CSC by item)."""
    nrows: int
    ncols: int
    indptr: np.ndarray      # int64 [nrows+1]
    indices: np.ndarray     # int32 [nnz]
    vals: np.ndarray        # float32 [nnz]


def coo_to_sides(rows, cols, vals, shape) -> Tuple[SparseSide, SparseSide]:
    """Row-major-sorted CSR (users) and column-major-sorted CSC (items).

    Index order inside a row / column is ascending, as `np.flatnonzero` yields
    in the reference (scripts/als.py:338-339).
    """
    m, n = int(shape[0]), int(shape[1])
    rows = np.asarray(rows)
    cols = np.asarray(cols)
    vals = np.asarray(vals, dtype=np.float32)
    if rows.size:
        if rows.min() < 0 or rows.max() >= m or cols.min() < 0 or cols.max() >= n:
            raise ValueError("rating index outside the matrix shape")
    key = rows.astype(np.int64) * n + cols.astype(np.int64)
    order = np.argsort(key, kind="stable")
    if order.size > 1 and np.any(np.diff(key[order]) == 0):
        raise ValueError("duplicate (user, item) entries")
    ru, ri, rv = rows[order], cols[order], vals[order]
    uptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(np.bincount(ru, minlength=m), out=uptr[1:])
    corder = np.argsort(ri.astype(np.int64) * m + ru.astype(np.int64), kind="stable")
    iptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(ri, minlength=n), out=iptr[1:])
    csr = SparseSide(m, n, uptr, ri.astype(np.int32), rv)
    csc = SparseSide(n, m, iptr, ru[corder].astype(np.int32), rv[corder])
    return csr, csc


def dense_to_coo(R: np.ndarray):
    """Dense NaN-coded matrix -> COO (scripts/als.py:335,340)."""
    ru, ri = np.nonzero(~np.isnan(R))
    return ru, ri, R[ru, ri]


@dataclass
class RowTasks:
    """Task list of one orientation / one shard (struct als_task, als_long_row)."""
    tasks: np.ndarray        # int32 [ntasks, 4]  (row, seg, slot, 0)
    long_rows: np.ndarray    # int32 [nlong, 4]   (row, slot0, nslots, 0)
    nslots: int
    nnz: int                 # ratings covered
    ndual: int = 0           # trailing tasks that are whole rows of at most `dual_len` ratings
    nmid: int = 0            # tasks just before them: whole rows of dual_len < n <= `mid_len` ratings


def dual_max_len(k: int) -> int:
    """Longest row the engine solves in the dual form at rank k: at most 64 ratings for k > 64 (n x n instead
    of k x k system, 5x fewer cycles per row at k = 128).  For k <= 64 the kernel supports it too (rows with
    fewer 16-rating blocks than k/16), but at k = 64 it measured the same as the primal kernel (U-step 7.68 ms
    either way; after the panel trimming of round 1: 6.68 primal vs 6.84 / 6.55 / 6.64 with rows of up to 16 /
    32 / 48 ratings dual - run-to-run noise), so the engine leaves those rows primal.  ALS_DUAL_LEN overrides."""
    import os
    env = os.environ.get("ALS_DUAL_LEN")
    if env is not None:                      # tuning knob (multiples of 16, < padded k)
        return max(0, min(int(env), DUAL_MAX_LEN, padded_k(k) - 16))
    return DUAL_MAX_LEN if padded_k(k) // 16 >= 5 else 0


def dual_mid_len(k: int) -> int:
    """Above k = 96 rows of 65 ... 96 ratings take the dual form too (an 80- or 96-size system at two waves
    per SIMD instead of the k x k one at one wave per SIMD); 0 = no such class."""
    return DUAL_MID_LEN if padded_k(k) // 16 >= 7 else 0


def build_row_tasks(indptr: np.ndarray, row_begin: int = 0, row_end: Optional[int] = None,
                    chunk: int = SPLIT_CHUNK, dual_len: int = DUAL_MAX_LEN, mid_len: int = 0) -> RowTasks:
    """Tasks for rows [row_begin, row_end) with at least one rating.

    Rows longer than `chunk` are split into segments whose partial normal
    equations are summed in segment order by the finishing kernel.  Tasks are
    ordered longest-first so that the tail of the launch is made of short rows.
    """
    row_end = len(indptr) - 1 if row_end is None else row_end
    counts = np.diff(indptr[row_begin:row_end + 1])
    rows = np.nonzero(counts > 0)[0]
    cnt = counts[rows]
    rows = rows + row_begin
    nseg = (cnt + chunk - 1) // chunk
    is_long = nseg > 1
    # short rows: one task each
    s_rows, s_len = rows[~is_long], cnt[~is_long]
    # long rows: nseg tasks each, consecutive slots
    l_rows, l_nseg, l_cnt = rows[is_long], nseg[is_long], cnt[is_long]
    slot0 = np.zeros(l_rows.size, dtype=np.int64)
    if l_rows.size:
        slot0[1:] = np.cumsum(l_nseg)[:-1]
    nslots = int(l_nseg.sum())
    t_row = np.repeat(l_rows, l_nseg)
    t_seg = (np.arange(nslots) - np.repeat(slot0, l_nseg)).astype(np.int64)
    t_slot = np.arange(nslots, dtype=np.int64)
    t_len = np.minimum(chunk, np.repeat(l_cnt, l_nseg) - t_seg * chunk)
    all_row = np.concatenate([t_row, s_rows])
    all_seg = np.concatenate([t_seg, np.zeros(s_rows.size, dtype=np.int64)])
    all_slot = np.concatenate([t_slot, -np.ones(s_rows.size, dtype=np.int64)])
    all_len = np.concatenate([t_len, s_len])
    # Longest first.  Every segment of a split row but its last has exactly `chunk` ratings; those lead, ordered by
    # the QUANTILE of the row they cover, (seg + 1/2) / nseg, not row by row: a row's ratings are stored in
    # ascending index of the other side, so workgroups running at the same time then gather from overlapping
    # slices of the other side's factor table (cache-resident: 244 MiB of U at cfg 4 against the 256 MiB Infinity
    # Cache and 4 MiB of L2 per XCD) instead of from the whole table.  Slots - the order a row's partial sums are
    # added in - do not change.  Equal lengths otherwise: last segments in row order, then whole rows in row order.
    rep_nseg = np.repeat(l_nseg, l_nseg)
    inner = t_seg < rep_nseg - 1
    grp = np.concatenate([np.where(inner, 0, 1), np.full(s_rows.size, 2)])
    key = np.concatenate([np.where(inner, (2 * t_seg + 1) / (2.0 * np.maximum(rep_nseg, 1)), 0.0), np.zeros(s_rows.size)])
    order = np.lexsort((np.arange(all_len.size), key, grp, -all_len))
    # ... and dealt over the XCDs: als_row_solve runs task b in workgroup b (one wave per workgroup), which the
    # dispatcher places on XCD b % NXCD, so position b receives element (b % 8) * (n / 8) + b / 8 of the quantile order
    # (bijective form for n % 8 != 0): every XCD works through ONE contiguous eighth of the order - its own slice of the
    # factor table in its own L2 - instead of all eight sharing every slice.
    nf = int(inner.sum())
    if nf >= 2 * NXCD:
        b = np.arange(nf)
        q, r = divmod(nf, NXCD)
        x = b % NXCD
        order[:nf] = order[:nf][np.where(x < r, x * (q + 1), r * (q + 1) + (x - r) * q) + b // NXCD]
    # whole rows of at most `dual_len` ratings go last (longest-first inside both parts): als_row_solve may
    # hand that tail to the dual-form kernel (`ndual_tail`, k > 64)
    # ... preceded by the whole rows of dual_len < n <= mid_len ratings (`ndual_mid`, k > 96)
    whole = all_slot[order] < 0
    short_whole = (all_len[order] <= dual_len) & whole
    mid_whole = (all_len[order] <= mid_len) & whole & ~short_whole
    order = np.concatenate([order[~short_whole & ~mid_whole], order[mid_whole], order[short_whole]])
    tasks = np.zeros((all_row.size, 4), dtype=np.int32)
    tasks[:, 0] = all_row[order]
    tasks[:, 1] = all_seg[order]
    tasks[:, 2] = all_slot[order]
    long_rows = np.zeros((l_rows.size, 4), dtype=np.int32)
    long_rows[:, 0] = l_rows
    long_rows[:, 1] = slot0
    long_rows[:, 2] = l_nseg
    return RowTasks(tasks, long_rows, nslots, int(cnt.sum()), int(short_whole.sum()), int(mid_whole.sum()))


def shard_bounds(nrows: int, world: int, multiple: int = 1) -> Tuple[int, List[Tuple[int, int]]]:
    """Equal contiguous row shards: (rows per shard, [(begin, end) per rank]); rows per shard is
    rounded up to a multiple of `multiple` (sub-chunks of a shard then have equal sizes too)."""
    per = (nrows + world - 1) // world
    per = ((per + multiple - 1) // multiple) * multiple
    return per, [(min(r * per, nrows), min((r + 1) * per, nrows)) for r in range(world)]


def row_cost_weight(k: int) -> float:
    """Fixed cost of one row of als_row_solve in units of one rating's cost: the k x k factorisation, the two
    substitutions and the row's set-up against the per-rating Gram work.  Fitted from the one-GPU phase times at
    k = 64 (Cholesky + substitutions + fixed 3.4 ms per 10^6 rows against 3.4 ms per 10^8
    ratings -> ~100) and scaled with k (per row ~k^2 vector + k^3 matrix work, per rating ~k^2)."""
    return 1.6 * float(k)


def shard_bounds_nnz(indptr: np.ndarray, world: int, chunks: int = 1, row_cost: float = 0.0):
    """Contiguous row shards balanced by COST = ratings + row_cost * rows (row_cost = 0:
    by ratings alone): shard r ends at the first row where the running cost reaches (r + 1) / world of the total.
    Returns ([(begin, end)] per rank, [[(begin, end)] * chunks per rank]): every shard is cut the same way into
    `chunks` sub-ranges (the U-step solves and all-gathers them one after the other).  Deterministic from
    `indptr` alone, so every rank computes the same tables.  Row counts differ between shards; empty shards are
    possible (world > rows).  Only rows with ratings cost anything (empty rows are not solved)."""
    indptr = np.asarray(indptr, dtype=np.int64)
    nrows = len(indptr) - 1
    total = int(indptr[-1])
    if row_cost > 0.0 and nrows:
        busy = np.zeros(nrows + 1, dtype=np.float64)
        np.cumsum(np.diff(indptr) > 0, out=busy[1:])
        cost = indptr.astype(np.float64) + float(row_cost) * busy
    else:
        cost = indptr.astype(np.float64)

    def cut(lo: int, hi: int, parts: int):
        c_lo, c_hi = float(cost[lo]), float(cost[hi])
        edges = [lo]
        for j in range(1, parts):
            target = c_lo + (c_hi - c_lo) * j / parts
            e = int(np.searchsorted(cost, target, side="left"))
            edges.append(min(max(e, edges[-1]), hi))
        edges.append(hi)
        return [(edges[j], edges[j + 1]) for j in range(parts)]

    if total == 0:                                   # nothing to balance: equal row counts
        per = -(-nrows // world)
        bounds = [(min(r * per, nrows), min((r + 1) * per, nrows)) for r in range(world)]
    else:
        bounds = cut(0, nrows, world)
    return bounds, [cut(b, e, chunks) for b, e in bounds]


def shard_costs(indptr: np.ndarray, bounds, row_cost: float) -> np.ndarray:
    """Predicted cost (ratings + row_cost * non-empty rows) of every shard of `bounds`."""
    indptr = np.asarray(indptr, dtype=np.int64)
    lens = np.diff(indptr)
    return np.array([float(indptr[e] - indptr[b]) + row_cost * float(np.count_nonzero(lens[b:e])) for b, e in bounds])


def build_similarity_dense(X: np.ndarray, topk: Optional[int], eps: float) -> np.ndarray:
    """Item-item cosine top-k graph, symmetrised by max.

    Same numpy call sequence as the reference (scripts/als.py:224-240) so that
    `argpartition`'s tie order at the top-k boundary (which decides the graph
    when features are binary) is the reference's.  O(n^2 d): host-side, small
    n only; larger problems pass a precomputed CSR graph to `fit`.
    """
    norms = np.sqrt((X * X).sum(axis=1, keepdims=True)) + eps
    Xn = X / norms
    S = Xn @ Xn.T
    np.fill_diagonal(S, 0.0)
    if topk is not None and topk < S.shape[0]:
        for i in range(S.shape[0]):
            drop = np.argpartition(S[i], -topk)[:-topk]
            S[i, drop] = 0.0
    return np.maximum(S, S.T)


def _graph_csr_from_coo(r, c, v, n, device):
    """Sorted COO (unique entries) -> (ptr int64, idx int32, val float32, D float32); D = S.sum(axis=1) from an
    fp64 prefix sum (deterministic)."""
    import torch
    order = torch.argsort(r * n + c)
    r, c, v = r[order], c[order], v[order]
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    ptr[1:] = torch.cumsum(torch.bincount(r, minlength=n), 0)
    csum = torch.zeros(v.numel() + 1, dtype=torch.float64, device=device)
    csum[1:] = torch.cumsum(v.to(torch.float64), 0)
    D = (csum[ptr[1:]] - csum[ptr[:-1]]).to(torch.float32)
    return ptr, c.to(torch.int32), v.to(torch.float32), D


TOPK_KERNEL_MAX_D = 64
TOPK_KERNEL_MAX_K = 128      # ALS_TOPK_MAX


def build_similarity_kernel(lib, X, topk: int, eps: float, device, stream=None):
    """The graph of `build_similarity_dense` by the hand-written kernels of csrc/graph_build.hip (SURVEY section
    8(f) n2): als_topk_similarity (cosine products on the fp32 matrix cores + per-row top-k under the total order
    (similarity descending, index ascending)) and als_graph_classify (max-symmetrisation on the lists); the
    surviving entries are sorted into CSR here.  fp32 throughout; d <= 64, topk <= 128, topk < n."""
    import ctypes as C
    import torch
    if torch.is_tensor(X):
        Xd = X.to(device=device, dtype=torch.float32)
    else:
        Xd = torch.as_tensor(np.asarray(X, dtype=np.float32), device=device)
    n, d = Xd.shape
    Xn = Xd / (torch.sqrt((Xd * Xd).sum(1, keepdim=True)) + np.float32(eps))
    ns = next(s for s in (1, 2, 4, 5, 8, 16) if 4 * s >= d)
    n_pad = 16 * ((n + 15) // 16)
    XT = torch.zeros(n_pad, 4 * ns, dtype=torch.float32, device=device)
    XT[:n, :d] = Xn
    XT = XT.view(n_pad, ns, 4).permute(1, 0, 2).contiguous()
    tv = torch.empty(n, topk, dtype=torch.float32, device=device)
    ti = torch.empty(n, topk, dtype=torch.int32, device=device)
    tc = torch.empty(n, dtype=torch.int32, device=device)
    own = torch.empty(n, topk, dtype=torch.uint8, device=device)
    mir = torch.empty(n, topk, dtype=torch.uint8, device=device)
    st = C.c_void_p(torch.cuda.current_stream(device).cuda_stream if stream is None else stream)
    p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
    rc = lib.als_topk_similarity(n, n_pad, ns, p(XT), int(topk), p(tv), p(ti), p(tc), st)
    if rc == 0:
        rc = lib.als_graph_classify(n, int(topk), p(tv), p(ti), p(tc), p(own), p(mir), st)
    if rc != 0:
        raise RuntimeError(f"graph build kernels failed with status {rc}")
    rows = torch.arange(n, device=device)[:, None].expand(-1, topk)
    o, mr = own.bool(), mir.bool()
    r = torch.cat([rows[o], ti[mr].to(torch.int64)])
    c = torch.cat([ti[o].to(torch.int64), rows[mr]])
    v = torch.cat([tv[o], tv[mr]])
    return _graph_csr_from_coo(r, c, v, n, device)


def build_similarity_device(X, topk: Optional[int], eps: float, device, block: int = 4096, lib=None):
    """Device-side build of the same graph as `build_similarity_dense`, without the n x n matrix
    (SURVEY section 8(f) n2): cosine products, per-row top-k, symmetrised by max, returned as CSR
    (ptr int64, idx int32, val float32, D float32) device tensors.  With the C-ABI library (`lib`) and
    d <= 64, topk <= 128 < n this is `build_similarity_kernel`; otherwise the blocked torch formulation below.

    NOT tie-identical to the reference: among equal similarities at the top-k boundary numpy's
    `argpartition` (scripts/als.py:235) keeps an implementation-defined subset, here the lowest
    column indices win.  With all-distinct similarities the graphs are identical.  Zero similarities
    are not edges, as in the dense form.  Opt-in (`ALS(..., graph_build="device")`); the host build
    stays the default because binary features (genres) tie massively.
    """
    import torch
    Xshape = np.shape(X)
    if (lib is not None and topk is not None and topk < Xshape[0] and topk <= TOPK_KERNEL_MAX_K
            and Xshape[1] <= TOPK_KERNEL_MAX_D and torch.device(device).type == "cuda"):
        return build_similarity_kernel(lib, X, topk, eps, device)
    if lib is not None and torch.device(device).type == "cuda":
        # not silently: the caller asked for the kernels and gets the O(n^2) blocked torch formulation instead
        import logging
        logging.getLogger(__name__).warning(
            "graph_build='device': %d feature columns / top-k %s are outside what the top-k kernels take "
            "(d <= %d, top-k <= %d < n); building the graph with the blocked torch formulation instead",
            Xshape[1], topk, TOPK_KERNEL_MAX_D, TOPK_KERNEL_MAX_K)
    Xd = torch.as_tensor(np.asarray(X), device=device)
    n = Xd.shape[0]
    Xn = Xd / (torch.sqrt((Xd * Xd).sum(1, keepdim=True)) + eps)
    keep_all = topk is None or topk >= n
    rows, cols, vals = [], [], []
    for b in range(0, n, block):
        e = min(b + block, n)
        Sb = Xn[b:e] @ Xn.T                                   # [e-b, n] in X's dtype
        ar = torch.arange(b, e, device=device)
        Sb[ar - b, ar] = 0.0
        if keep_all:
            nz = torch.nonzero(Sb)
            rows.append(nz[:, 0] + b); cols.append(nz[:, 1]); vals.append(Sb[nz[:, 0], nz[:, 1]])
        else:
            # stable descending sort: equal values keep ascending column order -> lowest indices win
            tv, ti = torch.sort(Sb, dim=1, descending=True, stable=True)
            tv, ti = tv[:, :topk], ti[:, :topk]
            ok = tv != 0
            rows.append(ar[:, None].expand(-1, topk)[ok]); cols.append(ti[ok]); vals.append(tv[ok])
    r = torch.cat(rows).to(torch.int64); c = torch.cat(cols).to(torch.int64); v = torch.cat(vals)
    key = torch.cat([r * n + c, c * n + r])                 # max(S, S^T): both orientations, reduce by max
    val = torch.cat([v, v])
    uk, inv = torch.unique(key, return_inverse=True)
    sv = torch.full((uk.numel(),), -float("inf"), dtype=val.dtype, device=device).scatter_reduce_(0, inv, val, reduce="amax")
    sr = torch.div(uk, n, rounding_mode="floor")
    sc = uk - sr * n
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
    ptr[1:] = torch.cumsum(torch.bincount(sr, minlength=n), 0)
    sv32 = sv.to(torch.float32)
    csum = torch.zeros(sv.numel() + 1, dtype=torch.float64, device=device)
    csum[1:] = torch.cumsum(sv.to(torch.float64), 0)
    D = (csum[ptr[1:]] - csum[ptr[:-1]]).to(torch.float32)
    return ptr, sc.to(torch.int32), sv32, D


def dense_graph_to_csr(S: np.ndarray):
    ri, ci = np.nonzero(S)
    ptr = np.zeros(S.shape[0] + 1, dtype=np.int64)
    np.cumsum(np.bincount(ri, minlength=S.shape[0]), out=ptr[1:])
    return ptr, ci.astype(np.int32), S[ri, ci]


@dataclass
class LevelSchedule:
    items: np.ndarray        # int32: swept items, grouped by level, ascending id inside
    offsets: np.ndarray      # int64 [nlevels+1] into items
    level: np.ndarray        # int64 [n]: level of every item, -1 if it is not swept


def wait_edges(S_ptr: np.ndarray, S_idx: np.ndarray, level: np.ndarray) -> np.ndarray:
    """S_idx with the sign bit set on the edges (i -> j) the sweep of item i must wait for:
    j < i and j swept (level[j] >= 0).  Input of als_gs_sweep_dataflow."""
    n = len(S_ptr) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(S_ptr))
    j = S_idx.astype(np.int64)
    wait = (j < rows) & (level[j] >= 0) & (level[rows] >= 0)
    out = S_idx.astype(np.int32).copy()
    out[wait] |= np.int32(-2147483648)
    return out


def build_level_schedule(S_ptr: np.ndarray, S_idx: np.ndarray, active: np.ndarray,
                         begin: int = 0, end: Optional[int] = None) -> LevelSchedule:
    """Dependency levels of the index-ordered Gauss-Seidel sweep.

    Item i (active = has ratings, begin <= i < end) must run after every
    neighbour j < i that is itself swept in [begin, end); neighbours j > i
    must still hold their previous value, which symmetry of S guarantees
    (j > i neighbour => level(j) > level(i)).  level(i) = 1 + max level of
    earlier swept neighbours.
    """
    n = len(S_ptr) - 1
    end = n if end is None else end
    level = np.full(n, -1, dtype=np.int64)
    for i in range(begin, end):
        if not active[i]:
            continue
        nb = S_idx[S_ptr[i]:S_ptr[i + 1]]
        nb = nb[(nb < i) & (nb >= begin)]
        lv = level[nb].max() if nb.size else -1
        level[i] = lv + 1                      # inactive neighbours carry -1
    swept = np.nonzero(level >= 0)[0]
    order = np.lexsort((swept, level[swept]))
    items = swept[order].astype(np.int32)
    nlev = int(level.max()) + 1 if swept.size else 0
    offsets = np.zeros(nlev + 1, dtype=np.int64)
    if swept.size:
        np.cumsum(np.bincount(level[swept], minlength=nlev), out=offsets[1:])
    return LevelSchedule(items, offsets, level)


# --------------------------------------------------------------------------------------------------------
# native (C-ABI, host) forms of the set-up passes: same outputs as the numpy definitions above
# --------------------------------------------------------------------------------------------------------
def _vp(a: np.ndarray):
    import ctypes as C
    return C.c_void_p(a.ctypes.data)


def coo_to_sides_native(lib, rows, cols, vals, shape) -> Tuple[SparseSide, SparseSide]:
    m, n = int(shape[0]), int(shape[1])
    rows = np.ascontiguousarray(rows, dtype=np.int64)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=np.float32)
    N = rows.size
    uptr, iptr = np.empty(m + 1, np.int64), np.empty(n + 1, np.int64)
    uidx, iidx = np.empty(N, np.int32), np.empty(N, np.int32)
    uval, ival = np.empty(N, np.float32), np.empty(N, np.float32)
    rc = lib.als_host_coo_to_sides(m, n, N, _vp(rows), _vp(cols), _vp(vals), _vp(uptr), _vp(uidx), _vp(uval),
                                   _vp(iptr), _vp(iidx), _vp(ival))
    if rc == -10:
        raise ValueError("rating index outside the matrix shape")
    if rc == -11:
        raise ValueError("duplicate (user, item) entries")
    if rc != 0:
        raise RuntimeError(f"als_host_coo_to_sides failed with status {rc}")
    return SparseSide(m, n, uptr, uidx, uval), SparseSide(n, m, iptr, iidx, ival)


def build_row_tasks_native(lib, indptr: np.ndarray, row_begin: int = 0, row_end: Optional[int] = None,
                           chunk: int = SPLIT_CHUNK, dual_len: int = DUAL_MAX_LEN, mid_len: int = 0) -> RowTasks:
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    row_end = len(indptr) - 1 if row_end is None else row_end
    counts = np.zeros(6, np.int64)
    args = (_vp(indptr), int(row_begin), int(row_end), int(chunk), int(dual_len), int(mid_len))
    rc = lib.als_host_row_tasks(*args, None, None, _vp(counts))
    if rc != 0:
        raise RuntimeError(f"als_host_row_tasks failed with status {rc}")
    tasks = np.zeros((int(counts[0]), 4), np.int32)
    long_rows = np.zeros((int(counts[1]), 4), np.int32)
    rc = lib.als_host_row_tasks(*args, _vp(tasks), _vp(long_rows), _vp(counts))
    if rc != 0:
        raise RuntimeError(f"als_host_row_tasks failed with status {rc}")
    return RowTasks(tasks, long_rows, int(counts[2]), int(counts[3]), int(counts[4]), int(counts[5]))


def build_level_schedule_native(lib, S_ptr: np.ndarray, S_idx: np.ndarray, active: np.ndarray, begin: int = 0,
                                end: Optional[int] = None, want_wait: bool = True):
    """(LevelSchedule, S_idx_wait or None) - `build_level_schedule` and `wait_edges` in one pass."""
    S_ptr = np.ascontiguousarray(S_ptr, dtype=np.int64)
    S_idx = np.ascontiguousarray(S_idx, dtype=np.int32)
    act = np.ascontiguousarray(active, dtype=np.uint8)
    n = len(S_ptr) - 1
    end = n if end is None else end
    level = np.empty(n, np.int64)
    items = np.empty(n, np.int32)
    offsets = np.empty(n + 1, np.int64)
    wait = np.empty(S_idx.size, np.int32) if want_wait else None
    out = np.zeros(2, np.int64)
    rc = lib.als_host_level_schedule(n, _vp(S_ptr), _vp(S_idx), _vp(act), int(begin), int(end), _vp(level),
                                     _vp(items), _vp(offsets), _vp(wait) if want_wait else None, _vp(out))
    if rc != 0:
        raise RuntimeError(f"als_host_level_schedule failed with status {rc}")
    return LevelSchedule(items[: int(out[0])].copy(), offsets[: int(out[1]) + 1].copy(), level), wait
