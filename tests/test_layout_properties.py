"""Property tests (hypothesis) of the host-side layout logic that the kernels rely on."""
import numpy as np
from hypothesis import given, settings, strategies as st

from collaborative_filtering_amd import layout


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 80), min_size=1, max_size=60), st.integers(3, 17), st.data())
def test_row_tasks_cover_every_rating_once(lens, chunk, data):
    indptr = np.zeros(len(lens) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(lens)
    b = data.draw(st.integers(0, len(lens)))
    e = data.draw(st.integers(b, len(lens)))
    t = layout.build_row_tasks(indptr, b, e, chunk=chunk)
    covered = np.zeros(int(indptr[-1]), dtype=np.int64)
    seen_slots = []
    for row, seg, slot, _ in t.tasks:
        lo = indptr[row] + seg * chunk
        hi = min(lo + chunk, indptr[row + 1])
        assert b <= row < e and lo < hi
        covered[lo:hi] += 1
        n_seg = -(-(indptr[row + 1] - indptr[row]) // chunk)
        assert (slot < 0) == (n_seg == 1)
        if slot >= 0:
            seen_slots.append(int(slot))
    inside = np.zeros_like(covered)
    inside[indptr[b]:indptr[e]] = 1
    np.testing.assert_array_equal(covered, inside)                      # exactly the shard's ratings, once
    assert sorted(seen_slots) == list(range(t.nslots))                   # slots are a permutation of 0..nslots-1
    for row, slot0, nslots, _ in t.long_rows:                            # consecutive slots per long row
        segs = sorted((int(s), int(sl)) for r, s, sl, _ in t.tasks if r == row)
        assert [sl for _, sl in segs] == list(range(slot0, slot0 + nslots))
    # the inner (full-length) segments of split rows lead the list, in ascending quantile (seg + 1/2) / nseg of
    # the row they cover (cache locality of the gathers; slots - the summation order - are untouched)
    nseg_of = {int(r): int(ns) for r, _, ns, _ in t.long_rows}
    inner = [(int(r), int(s)) for r, s, sl, _ in t.tasks if sl >= 0 and s < nseg_of[int(r)] - 1]
    assert [(int(r), int(s)) for r, s, _, _ in t.tasks[:len(inner)]] == inner
    q = [(2 * s + 1) / (2.0 * nseg_of[r]) for r, s in inner]
    if len(q) < 2 * layout.NXCD:
        assert q == sorted(q)
    else:       # dealt over the XCDs: the positions b = x (mod 8) hold ONE contiguous, ascending part of the order each
        parts = [q[x::layout.NXCD] for x in range(layout.NXCD)]
        assert all(p == sorted(p) for p in parts)
        assert all(parts[x][-1] <= parts[x + 1][0] for x in range(layout.NXCD - 1))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    lens_sorted = [min(chunk, indptr[r + 1] - indptr[r] - s * chunk) for r, s, _, _ in t.tasks]
    assert t.nmid == 0                                                    # mid_len defaults to 0
    head, tail = lens_sorted[:len(lens_sorted) - t.ndual], lens_sorted[len(lens_sorted) - t.ndual:]
    assert head == sorted(head, reverse=True) and tail == sorted(tail, reverse=True)   # longest first, twice
    # the tail is exactly the whole rows of at most DUAL_MAX_LEN ratings
    nt = len(t.tasks)
    for j, (r, s, slot, _) in enumerate(t.tasks):
        whole_short = slot < 0 and (indptr[r + 1] - indptr[r]) <= layout.DUAL_MAX_LEN
        assert whole_short == (j >= nt - t.ndual)
    assert t.nnz == int(indptr[e] - indptr[b])


def _random_symmetric_graph(n, rng, p):
    A = np.triu(rng.random((n, n)) < p, 1)
    A = A | A.T
    ptr = np.zeros(n + 1, dtype=np.int64)
    ptr[1:] = np.cumsum(A.sum(1))
    idx = np.concatenate([np.flatnonzero(A[i]) for i in range(n)]).astype(np.int32) if A.any() else np.zeros(0, np.int32)
    return A, ptr, idx


@settings(max_examples=40, deadline=None)
@given(st.integers(2, 60), st.floats(0.02, 0.5), st.integers(0, 10_000), st.data())
def test_level_schedule_respects_gauss_seidel_order(n, p, seed, data):
    rng = np.random.default_rng(seed)
    A, ptr, idx = _random_symmetric_graph(n, rng, p)
    active = rng.random(n) < 0.8
    b = data.draw(st.integers(0, n - 1))
    e = data.draw(st.integers(b + 1, n))
    s = layout.build_level_schedule(ptr, idx, active, b, e)
    swept = set(int(i) for i in s.items)
    assert swept == {i for i in range(b, e) if active[i]}
    lev = s.level
    for i in swept:
        for j in np.flatnonzero(A[i]):
            if j in swept:
                assert (lev[j] < lev[i]) == (j < i)          # earlier neighbours strictly before, later strictly after
    # items grouped by level, ascending id inside a level, offsets consistent
    for l in range(len(s.offsets) - 1):
        grp = s.items[s.offsets[l]:s.offsets[l + 1]]
        assert np.all(lev[grp] == l) and np.all(np.diff(grp) > 0)
    assert s.offsets[-1] == len(s.items)
    # wait edges: exactly the earlier swept neighbours of swept items
    w = layout.wait_edges(ptr, idx, lev)
    rows = np.repeat(np.arange(n), np.diff(ptr))
    expect = (idx < rows) & np.isin(idx, list(swept)) & np.isin(rows, list(swept))
    np.testing.assert_array_equal(w < 0, expect)
    np.testing.assert_array_equal(w & 0x7FFFFFFF, idx)


@given(st.integers(1, 10_000), st.integers(1, 9), st.integers(1, 5))
def test_shard_bounds_tile_the_rows(nrows, world, multiple):
    per, bounds = layout.shard_bounds(nrows, world, multiple)
    assert per % multiple == 0 and per * world >= nrows
    assert bounds[0][0] == 0 and bounds[-1][1] == nrows
    for (b0, e0), (b1, e1) in zip(bounds, bounds[1:]):
        assert e0 == b1 and b0 <= e0
    assert all(e - b <= per for b, e in bounds)


@given(st.integers(1, 160))
def test_perm_is_a_bijection_with_contiguous_lane_loads(k):
    pos = layout.perm_of_col(k)
    ld = layout.padded_k(k)
    kb = ld // 16
    assert sorted(pos) == list(range(ld))
    # lane c's KB contiguous storage columns are position c of blocks 0..KB-1
    for c in range(16):
        assert [int(pos[kb * c + b]) for b in range(kb)] == [16 * b + c for b in range(kb)]


@settings(max_examples=30, deadline=None)
@given(st.integers(1, 12), st.integers(1, 9), st.integers(0, 1000))
def test_coo_to_sides_round_trip(m, n, seed):
    rng = np.random.default_rng(seed)
    mask = rng.random((m, n)) < 0.4
    r, c = np.nonzero(mask)
    perm = rng.permutation(r.size)
    v = rng.integers(1, 10, size=r.size).astype(np.float32)
    csr, csc = layout.coo_to_sides(r[perm], c[perm], v[perm], (m, n))
    dense = np.zeros((m, n), dtype=np.float32)
    dense[r, c] = v
    for u in range(m):
        cols = csr.indices[csr.indptr[u]:csr.indptr[u + 1]]
        assert np.all(np.diff(cols) > 0)
        np.testing.assert_array_equal(dense[u, cols], csr.vals[csr.indptr[u]:csr.indptr[u + 1]])
        assert cols.size == mask[u].sum()
    for i in range(n):
        rows = csc.indices[csc.indptr[i]:csc.indptr[i + 1]]
        assert np.all(np.diff(rows) > 0)
        np.testing.assert_array_equal(dense[rows, i], csc.vals[csc.indptr[i]:csc.indptr[i + 1]])


@settings(max_examples=40, deadline=None)
@given(st.lists(st.integers(0, 200), min_size=1, max_size=80))
def test_row_tasks_three_classes(lens):
    """dual_len / mid_len split the whole rows into [others | 65..96 | <= 64], longest first inside each."""
    indptr = np.zeros(len(lens) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(lens)
    t = layout.build_row_tasks(indptr, dual_len=64, mid_len=96)
    n = len(t.tasks)
    ln = [int(indptr[r + 1] - indptr[r]) for r, _, _, _ in t.tasks]
    assert t.ndual == sum(1 for l in lens if 0 < l <= 64) and t.nmid == sum(1 for l in lens if 64 < l <= 96)
    a, b = n - t.ndual - t.nmid, n - t.ndual
    assert all(l > 96 for l in ln[:a]) and all(64 < l <= 96 for l in ln[a:b]) and all(0 < l <= 64 for l in ln[b:])
    for part in (ln[:a], ln[a:b], ln[b:]):
        assert part == sorted(part, reverse=True)



# ---------------------------------------------------------------------------------------------------------
# native set-up passes (csrc/host_setup.cpp) == the numpy definitions above, output for output
# ---------------------------------------------------------------------------------------------------------
def _lib():
    import os
    import __graft_entry__ as ge
    from collaborative_filtering_amd import _hip
    if not os.path.exists(_hip.LIB_PATH):
        ge.build()
    return _hip.load()


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 60), min_size=1, max_size=60), st.integers(3, 17), st.integers(0, 12), st.integers(0, 20),
       st.data())
def test_native_row_tasks_equal_numpy(lens, chunk, dual_len, mid_len, data):
    lib = _lib()
    indptr = np.zeros(len(lens) + 1, dtype=np.int64)
    indptr[1:] = np.cumsum(lens)
    b = data.draw(st.integers(0, len(lens)))
    e = data.draw(st.integers(b, len(lens)))
    a = layout.build_row_tasks(indptr, b, e, chunk=chunk, dual_len=dual_len, mid_len=mid_len)
    n = layout.build_row_tasks_native(lib, indptr, b, e, chunk=chunk, dual_len=dual_len, mid_len=mid_len)
    np.testing.assert_array_equal(a.tasks, n.tasks)
    np.testing.assert_array_equal(a.long_rows, n.long_rows)
    assert (a.nslots, a.nnz, a.ndual, a.nmid) == (n.nslots, n.nnz, n.ndual, n.nmid)


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 40), st.floats(0.0, 0.5), st.integers(0, 1000), st.data())
def test_native_level_schedule_equals_numpy(n, p, seed, data):
    lib = _lib()
    rng = np.random.default_rng(seed)
    A = np.triu(rng.random((n, n)) < p, 1)
    A = A | A.T
    ptr, idx, _ = layout.dense_graph_to_csr(A.astype(np.float32))
    active = rng.random(n) < 0.8
    b = data.draw(st.integers(0, n))
    e = data.draw(st.integers(b, n))
    s = layout.build_level_schedule(ptr, idx, active, b, e)
    w = layout.wait_edges(ptr, idx, s.level)
    sn, wn = layout.build_level_schedule_native(lib, ptr, idx, active, b, e)
    np.testing.assert_array_equal(s.level, sn.level)
    np.testing.assert_array_equal(s.items, sn.items)
    np.testing.assert_array_equal(s.offsets, sn.offsets)
    np.testing.assert_array_equal(w, wn)
    assert layout.build_level_schedule_native(lib, ptr, idx, active, b, e, want_wait=False)[1] is None


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 12), st.integers(1, 9), st.integers(0, 1000), st.booleans())
def test_native_coo_to_sides_equals_numpy(m, n, seed, shuffled):
    lib = _lib()
    rng = np.random.default_rng(seed)
    r, c = np.nonzero(rng.random((m, n)) < 0.4)
    v = rng.integers(1, 10, size=r.size).astype(np.float32)
    if shuffled:
        perm = rng.permutation(r.size)
        r, c, v = r[perm], c[perm], v[perm]
    for a, b in zip(layout.coo_to_sides(r, c, v, (m, n)), layout.coo_to_sides_native(lib, r, c, v, (m, n))):
        np.testing.assert_array_equal(a.indptr, b.indptr)
        np.testing.assert_array_equal(a.indices, b.indices)
        np.testing.assert_array_equal(a.vals, b.vals)


def test_native_coo_to_sides_rejects_bad_input():
    import pytest
    lib = _lib()
    with pytest.raises(ValueError, match="duplicate"):
        layout.coo_to_sides_native(lib, [0, 1, 0], [1, 1, 1], [1.0, 2.0, 3.0], (2, 2))
    with pytest.raises(ValueError, match="outside"):
        layout.coo_to_sides_native(lib, [0, 2], [1, 1], [1.0, 2.0], (2, 2))
