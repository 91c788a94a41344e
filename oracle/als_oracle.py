"""CPU oracle for the ALS fit/predict hot path.  TEST INFRASTRUCTURE ONLY.

This module is a float64 numpy/scipy restatement of the reference solver
(`/root/reference/scripts/als.py`, `helpers.py`) working on sparse COO/CSR/CSC
input instead of the reference's dense NaN matrix.  It exists so that the HIP
path can be checked against something that (a) is pinned against the real
reference through the committed golden fixtures (`tests/golden/*.npz`, made by
`tests/golden/make_golden.py`, which imports the real reference in the build
container) and (b) can travel to the GPU box, where `/root/reference` does not
exist.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this file.  The product package never does.

Parity status: PINNED - every function below is exercised by
`tests/test_oracle_golden.py` against outputs of the unmodified reference
(rtol 1e-9 on U, V, W, b_u, b_i, mu and all five history series).

Every function cites the reference lines it restates (paths relative to
/root/reference).  All reference quirks called out in SURVEY.md section 8(a) are
reproduced deliberately and marked [quirk].
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.linalg import cho_factor, cho_solve

SCALE_FACTOR = 0.1   # scripts/als.py:93
EPS = 1e-10          # scripts/als.py:94


# --------------------------------------------------------------------------
# numerics
# --------------------------------------------------------------------------
def cholesky_solve(A: np.ndarray, b: np.ndarray) -> np.ndarray:
    """SPD solve; raises numpy.linalg.LinAlgError when A is not SPD.

    Follows scripts/helpers.py:5-20 (cho_factor + cho_solve, check_finite off).
    """
    c, low = cho_factor(A, check_finite=False)
    return cho_solve((c, low), b, check_finite=False)


# --------------------------------------------------------------------------
# sparse containers
# --------------------------------------------------------------------------
@dataclass
class Ratings:
    """Observed ratings as CSR (by user), CSC (by item) and row-major COO.

    The COO order equals `np.where(mask)` of the reference (scripts/als.py:340):
    row-major, columns ascending inside a row.  CSR rows therefore list item
    ids ascending (als.py:338) and CSC columns list user ids ascending
    (als.py:339).
    """
    m: int
    n: int
    coo_u: np.ndarray          # (N,) int64, row-major order
    coo_i: np.ndarray          # (N,) int64
    coo_r: np.ndarray          # (N,) float64
    csr_ptr: np.ndarray        # (m+1,) int64
    csc_ptr: np.ndarray        # (n+1,) int64
    csc_u: np.ndarray          # (N,) int64 user ids, grouped by item
    csc_r: np.ndarray          # (N,) float64 values, grouped by item

    @property
    def nnz(self) -> int:
        return int(self.coo_r.shape[0])


def ratings_from_coo(rows, cols, vals, shape) -> Ratings:
    m, n = int(shape[0]), int(shape[1])
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    vals = np.asarray(vals, dtype=np.float64)
    order = np.lexsort((cols, rows))
    ru, ri, rv = rows[order], cols[order], vals[order]
    csr_ptr = np.zeros(m + 1, dtype=np.int64)
    np.add.at(csr_ptr, ru + 1, 1)
    csr_ptr = np.cumsum(csr_ptr)
    corder = np.lexsort((ru, ri))
    csc_ptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(csc_ptr, ri + 1, 1)
    csc_ptr = np.cumsum(csc_ptr)
    return Ratings(m, n, ru, ri, rv, csr_ptr, csc_ptr, ru[corder], rv[corder])


def ratings_from_dense(R: np.ndarray) -> Ratings:
    """Dense NaN matrix -> Ratings (scripts/als.py:332-340)."""
    mask = ~np.isnan(R)
    ru, ri = np.where(mask)
    return ratings_from_coo(ru, ri, R[ru, ri], R.shape)


# --------------------------------------------------------------------------
# configuration (field-for-field mirror of scripts/als_config.py:57-95)
# --------------------------------------------------------------------------
@dataclass
class OracleConfig:
    n_factors: int
    n_iters: int
    lambda_u: float
    lambda_v: float
    pop_reg_mode: Optional[str] = None
    random_state: int = 42
    update_w_every: int = 5
    lambda_bu: Optional[float] = None
    lambda_bi: Optional[float] = None
    alpha: float = 0.0
    sim: Optional[dict] = None      # {"feature_name","topk","eps"} or None
    lambda_w: Dict[str, float] = field(default_factory=dict)


# --------------------------------------------------------------------------
# similarity graph
# --------------------------------------------------------------------------
def build_item_similarity(X: np.ndarray, topk: Optional[int], eps: float) -> np.ndarray:
    """Dense cosine top-k similarity, symmetrised by max.

    Follows scripts/als.py:224-240 call for call (same numpy primitives, so
    the argpartition tie order - SURVEY section 7.7 - is the same on the same numpy).
    dtype follows X (float32 features give a float32 S).
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


def dense_to_csr(S: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    ptr = [0]
    idx: List[np.ndarray] = []
    val: List[np.ndarray] = []
    for i in range(S.shape[0]):
        nz = np.flatnonzero(S[i])
        idx.append(nz)
        val.append(S[i, nz])
        ptr.append(ptr[-1] + nz.size)
    return (np.asarray(ptr, dtype=np.int64),
            np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64),
            np.concatenate(val) if val else np.zeros(0, S.dtype))


# --------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------
class OracleALS:
    """float64 restatement of `scripts.als.ALS` on sparse ratings."""

    def __init__(self, cfg: OracleConfig):
        self.cfg = cfg
        self.k = int(cfg.n_factors)
        # [quirk] `x or y`: None *and* 0.0 fall back (scripts/als.py:166-167)
        self.lambda_bu = cfg.lambda_bu or cfg.lambda_u
        self.lambda_bi = cfg.lambda_bi or cfg.lambda_v
        self.W: Dict[str, np.ndarray] = {}
        self.U = self.V = self.b_u = self.b_i = None
        self.mu = 0.0
        self.S_dense: Optional[np.ndarray] = None
        self.S_csr = None
        self.D: Optional[np.ndarray] = None
        self.history = {"train_rmse": [], "U_norm": [], "V_norm": [],
                        "bu_norm": [], "bi_norm": []}

    # -- helpers ----------------------------------------------------------
    def item_reg(self, counts: np.ndarray) -> np.ndarray:
        """scripts/als.py:243-259."""
        mode = self.cfg.pop_reg_mode
        if not mode:
            return np.full_like(counts, self.cfg.lambda_v, dtype=float)
        if mode == "inverse_sqrt":
            return self.cfg.lambda_v / np.sqrt(counts + 1.0)
        raise ValueError(f"Unknown pop_reg_mode '{mode}'")

    def compose_Z(self, features: Dict[str, np.ndarray]) -> np.ndarray:
        """Z = V + sum_f X_f W_f (scripts/als.py:262-281)."""
        Z = self.V.copy()
        for name, X in features.items():
            W = self.W.get(name)
            if W is not None:
                Z += X @ W
        return Z

    # -- setup ------------------------------------------------------------
    def setup(self, rt: Ratings, features: Dict[str, np.ndarray],
              S_csr=None) -> None:
        """Validation, graph and initialisation (scripts/als.py:329-384).

        `S_csr=(ptr, idx, val)` supplies a precomputed similarity graph (used
        at sizes where the reference's dense n x n build is infeasible).
        """
        cfg = self.cfg
        rng = np.random.default_rng(cfg.random_state)               # :329
        m, n, k = rt.m, rt.n, self.k
        for name, X in features.items():                            # :346-351
            if X.shape[0] != n:
                raise ValueError(f"Feature '{name}' has {X.shape[0]} rows; "
                                 f"expected {n} (number of items).")
            if not np.isfinite(X).all():
                raise ValueError(f"Feature '{name}' contains infinite values.")
        use_graph = (cfg.alpha > 0.0) and (cfg.sim is not None)     # :354
        self.S_dense = None
        self.S_csr = None
        self.D = None
        if use_graph:
            if S_csr is not None:
                ptr, idx, val = S_csr
                self.S_csr = (np.asarray(ptr, np.int64), np.asarray(idx, np.int64),
                              np.asarray(val))
                # D = S.sum(axis=1) in S's dtype (:357); sparse sum order differs
                # from the dense pairwise sum only at rounding level of S.dtype
                self.D = np.array([val[ptr[i]:ptr[i + 1]].sum(dtype=val.dtype)
                                   for i in range(n)], dtype=val.dtype)
            else:
                X = features.get(cfg.sim["feature_name"])           # :215-222
                if X is not None:
                    self.S_dense = build_item_similarity(
                        X, cfg.sim.get("topk"), cfg.sim.get("eps", EPS))
                    self.D = self.S_dense.sum(axis=1)               # :357
        self.use_graph = (self.S_dense is not None) or (self.S_csr is not None)
        # mean of observed entries (nanmean over the dense matrix, :360)
        self.mu = float(np.mean(rt.coo_r))
        self.b_u = np.zeros(m, dtype=float)                         # :361
        self.b_i = np.zeros(n, dtype=float)                         # :362
        self.U = rng.normal(scale=SCALE_FACTOR, size=(m, k))        # :367
        self.V = rng.normal(scale=SCALE_FACTOR, size=(n, k))        # :368
        for name, X in features.items():                            # :371-376
            self.W[name] = rng.normal(scale=SCALE_FACTOR, size=(X.shape[1], k))
        counts = np.diff(rt.csc_ptr).astype(float)                  # :379
        self.lambda_v_i = self.item_reg(counts).astype(float)       # :382
        self.lambda_bi_i = np.full(n, float(self.lambda_bi))        # :384

    # -- half steps -------------------------------------------------------
    def user_step(self, rt: Ratings, Z: np.ndarray, rows=None) -> None:
        """scripts/als.py:414-433 (users without ratings keep their init)."""
        k = self.k
        I = np.eye(k)
        cfg = self.cfg
        rows = range(rt.m) if rows is None else rows
        for u in rows:
            lo, hi = rt.csr_ptr[u], rt.csr_ptr[u + 1]
            if hi == lo:
                continue
            idx = rt.coo_i[lo:hi]
            Ru = rt.coo_r[lo:hi]
            Z_u = Z[idx]
            r_u = Ru - (self.mu + self.b_u[u] + self.b_i[idx])      # :425 (old b_u)
            A = Z_u.T @ Z_u + (cfg.lambda_u + EPS) * I              # :426
            b = Z_u.T @ r_u                                         # :427
            self.U[u] = cholesky_solve(A, b)                        # :428
            denom = idx.size + self.lambda_bu + EPS                 # :431
            pred_wo_bu = (Z_u @ self.U[u]) + self.mu + self.b_i[idx]
            self.b_u[u] = float(np.sum(Ru - pred_wo_bu) / denom)    # :433

    def _graph_row(self, i: int) -> np.ndarray:
        """alpha-free Laplacian rhs S[i] @ V with the *live* V (als.py:458)."""
        if self.S_dense is not None:
            return self.S_dense[i] @ self.V
        ptr, idx, val = self.S_csr
        lo, hi = ptr[i], ptr[i + 1]
        return val[lo:hi].astype(np.float64) @ self.V[idx[lo:hi]]

    def item_step(self, rt: Ratings, cols=None) -> None:
        """scripts/als.py:436-466.

        [quirk] the feature part of Z is ignored here (:447,:465): V is fitted
        and b_i is updated as if Z == V.  With the graph on, the sweep is
        Gauss-Seidel in index order because `self.V` is read live (:458).
        """
        k = self.k
        I = np.eye(k)
        alpha = float(self.cfg.alpha)
        cols = range(rt.n) if cols is None else cols
        for i in cols:
            lo, hi = rt.csc_ptr[i], rt.csc_ptr[i + 1]
            if hi == lo:
                continue
            idx = rt.csc_u[lo:hi]
            Ri = rt.csc_r[lo:hi]
            U_i = self.U[idx]
            r = Ri - (self.mu + self.b_u[idx] + self.b_i[i])        # :447
            reg_i = self.lambda_v_i[i] + EPS                        # :450
            if self.use_graph:
                reg_i += alpha * float(self.D[i])                   # :454
            A = U_i.T @ U_i + reg_i * I                             # :455
            b = U_i.T @ r                                           # :456
            if self.use_graph:
                b += alpha * self._graph_row(i)                     # :458
            self.V[i] = cholesky_solve(A, b)                        # :461
            denom = idx.size + self.lambda_bi_i[i] + EPS            # :464
            pred_wo_bi = (U_i @ self.V[i]) + self.mu + self.b_u[idx]
            self.b_i[i] = float(np.sum(Ri - pred_wo_bi) / denom)    # :466

    def w_step(self, rt: Ratings, features: Dict[str, np.ndarray]) -> None:
        """scripts/als.py:469-501.

        [quirk] `residual` is never refreshed between features, so the update
        is Jacobi across features (:474-489).  [quirk] a feature missing from
        lambda_w is fitted with lambda = 0 (+1e-10) (:497).
        """
        k = self.k
        ru, ri, R_obs = rt.coo_u, rt.coo_i, rt.coo_r
        r_obs = R_obs - (self.mu + self.b_u[ru] + self.b_i[ri])     # :470
        r_obs = r_obs - np.sum(self.U[ru] * self.V[ri], axis=1)     # :471
        residual = r_obs.copy()
        for name, X in features.items():                            # :475-479
            W = self.W.get(name)
            if W is not None:
                WX = X[ri] @ W
                residual -= np.sum(self.U[ru] * WX, axis=1)
        for name, X in features.items():                            # :482-501
            d = X.shape[1]
            if name in self.W:
                target = residual.copy()
                WX = X[ri] @ self.W[name]
                target += np.sum(self.U[ru] * WX, axis=1)
            else:
                target = residual
            U_obs = self.U[ru]
            X_obs = X[ri]
            X_design = (X_obs[:, :, None] * U_obs[:, None, :]).reshape(len(ru), d * k)
            lam = float(self.cfg.lambda_w.get(name, 0.0))
            A = X_design.T @ X_design + (lam + EPS) * np.eye(d * k)
            b = X_design.T @ target
            self.W[name] = cholesky_solve(A, b).reshape(d, k)

    def mu_and_history(self, rt: Ratings, features: Dict[str, np.ndarray]) -> None:
        """scripts/als.py:503-517."""
        ru, ri, R_obs = rt.coo_u, rt.coo_i, rt.coo_r
        Z = self.compose_Z(features)
        pred_wo_mu = (np.sum(self.U[ru] * Z[ri], axis=1)
                      + self.b_u[ru] + self.b_i[ri])
        self.mu = float(np.mean(R_obs - pred_wo_mu))
        err = R_obs - (pred_wo_mu + self.mu)
        h = self.history
        h["train_rmse"].append(float(np.sqrt(np.mean(err ** 2))))
        h["U_norm"].append(float(np.linalg.norm(self.U)))
        h["V_norm"].append(float(np.linalg.norm(self.V)))
        h["bu_norm"].append(float(np.linalg.norm(self.b_u)))
        h["bi_norm"].append(float(np.linalg.norm(self.b_i)))

    def converged(self, tol: float, window: int = 2) -> bool:
        """scripts/als.py:283-297."""
        h = self.history["train_rmse"]
        return len(h) >= window + 1 and (h[-window - 1] - h[-1]) <= tol

    # -- driver -----------------------------------------------------------
    def fit(self, rt: Ratings, features: Optional[Dict[str, np.ndarray]] = None,
            tol: Optional[float] = 1e-3, min_iters: int = 5, S_csr=None) -> "OracleALS":
        """scripts/als.py:300-529 (logging/tqdm omitted)."""
        features = features or {}
        self.setup(rt, features, S_csr=S_csr)
        cfg = self.cfg
        for it in range(cfg.n_iters):                               # :408
            Z = self.compose_Z(features)                            # :411
            self.user_step(rt, Z)
            self.item_step(rt)
            if features and ((it % cfg.update_w_every == 0)
                             or (it == cfg.n_iters - 1)):           # :468
                self.w_step(rt, features)
            self.mu_and_history(rt, features)
            if tol is not None and it + 1 >= min_iters and self.converged(tol):
                break                                               # :520-523
        return self

    def predict(self, features: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
        """Dense completion (scripts/als.py:532-574)."""
        if self.U is None or self.V is None:
            raise RuntimeError("Model must be fitted before prediction.")
        features = features or {}
        n = self.V.shape[0]
        for name, X in features.items():
            if X.shape[0] != n:
                raise ValueError(f"Feature '{name}' has {X.shape[0]} rows. "
                                 f"Expected number of rows: {n}.")
            if not np.isfinite(X).all():
                raise ValueError(f"Feature '{name}' contains infinite values.")
        Z = self.compose_Z(features)
        return self.U @ Z.T + self.mu + self.b_u[:, None] + self.b_i[None, :]

    def predict_at(self, flat_idx: np.ndarray,
                   features: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
        """Predictions at flat indices u*n+i (the only way callers read
        `predict`'s output: scripts/tune_params.py:165-166)."""
        n = self.V.shape[0]
        Z = self.compose_Z(features or {})
        u, i = np.divmod(np.asarray(flat_idx, dtype=np.int64), n)
        return (np.sum(self.U[u] * Z[i], axis=1) + self.mu
                + self.b_u[u] + self.b_i[i])


def rmse_on_indices(R_true_flat_vals: np.ndarray, pred_vals: np.ndarray) -> float:
    """scripts/tune_params.py:147-167 on already-gathered values."""
    if R_true_flat_vals.size == 0:
        return float("nan")
    return float(np.sqrt(np.mean((R_true_flat_vals - pred_vals) ** 2)))
