"""Seeded synthetic inputs shared by the golden-fixture generator and the tests.

Nothing here comes from the reference: it is the build's own generator
(SURVEY.md section 8(d) recipe at test sizes).  Determinism relies on numpy's
PCG64 `Generator` streams, which are the same in this container and on the
GPU box (same image, numpy 2.2).
"""
from __future__ import annotations

import numpy as np


def make_ratings(m: int, n: int, nnz: int, seed: int, *, user_exp: float = 0.6,
                 item_exp: float = 0.9, empty_users=(), empty_items=()):
    """Power-law COO ratings with a planted rank-8 model, half-star values.

    Returns (rows, cols, vals) sorted row-major, values float64 in [0.5, 5].
    """
    rng = np.random.default_rng(seed)
    pu = (np.arange(1, m + 1, dtype=np.float64)) ** (-user_exp)
    pi = (np.arange(1, n + 1, dtype=np.float64)) ** (-item_exp)
    pu /= pu.sum()
    pi /= pi.sum()
    uperm = rng.permutation(m)
    iperm = rng.permutation(n)
    want = min(nnz, m * n)
    keys = np.zeros(0, dtype=np.int64)
    while keys.size < want:
        draw = max(int(1.3 * (want - keys.size)), 64)
        u = uperm[rng.choice(m, size=draw, p=pu)]
        i = iperm[rng.choice(n, size=draw, p=pi)]
        new = u.astype(np.int64) * n + i
        # keep first occurrences in draw order so truncation is seed-stable
        allk = np.concatenate([keys, new])
        _, first = np.unique(allk, return_index=True)
        keys = allk[np.sort(first)]
    keys = keys[:want]
    u, i = np.divmod(keys, n)
    keep = ~np.isin(u, np.asarray(empty_users, dtype=np.int64)) \
        & ~np.isin(i, np.asarray(empty_items, dtype=np.int64))
    u, i = u[keep], i[keep]
    us = rng.normal(scale=0.5, size=(m, 8))
    vs = rng.normal(scale=0.5, size=(n, 8))
    raw = 3.5 + np.sum(us[u] * vs[i], axis=1) + rng.normal(scale=0.5, size=u.size)
    vals = np.clip(np.round(raw * 2.0) / 2.0, 0.5, 5.0)
    order = np.lexsort((i, u))
    return u[order], i[order], vals[order]


def to_dense(rows, cols, vals, shape) -> np.ndarray:
    R = np.full(shape, np.nan, dtype=np.float64)
    R[rows, cols] = vals
    return R


# column rates of a 19-genre multi-hot matrix with MovieLens-like skew
_GENRE_RATES = np.array([0.18, 0.12, 0.07, 0.07, 0.38, 0.12, 0.05, 0.45, 0.08,
                         0.02, 0.10, 0.04, 0.06, 0.16, 0.10, 0.19, 0.04, 0.02,
                         0.01])


def make_features(n: int, seed: int):
    """(genres n x 19 row-L2 float32, years n x 1 z-scored float32)."""
    rng = np.random.default_rng(seed)
    G = (rng.random((n, 19)) < _GENRE_RATES[None, :]).astype(np.float64)
    nrm = np.sqrt((G * G).sum(axis=1, keepdims=True))
    G = (G / np.maximum(nrm, 1e-8)).astype(np.float32)
    y = rng.normal(size=(n, 1))
    y = ((y - y.mean()) / y.std()).astype(np.float32)
    return G, y


def make_folds(nnz: int, n_splits: int, seed: int):
    """Entrywise K-fold over COO positions (positions, not flat indices)."""
    rng = np.random.default_rng(seed)
    pos = np.arange(nnz)
    rng.shuffle(pos)
    return [np.sort(c) for c in np.array_split(pos, n_splits)]
