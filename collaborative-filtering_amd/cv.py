"""Sparse-native cross-validation / ablation harness around `ALS` (SURVEY.md section 8(f), row n1 + n3).

The reference drives `fit` / `predict` from two dense loops
(`scripts/evaluate_models.py:194-276`, `scripts/tune_params.py:341-421`): every fold materialises
dense NaN train / valid matrices (`scripts/create_folds.py:177-208`) and a dense m x n prediction
that is then read at the validation flat indices only (`scripts/tune_params.py:147-167`).  This
module restates that caller on COO triplets and flat indices - `fit_coo` + `predict_at`, nothing
m x n - with the same fold files, the same metrics, the same statistics and the same artifact
schema, so that results can be compared file by file:

  folds        make_entrywise_folds / save_folds_npz / load_folds_npz   create_folds.py:50-149
  split        train_valid_split                                        create_folds.py:152-208
  metrics      rmse_at, popularity_bins, split_by_popularity            tune_params.py:147-167,
                                                                        evaluate_models.py:131-191
  statistics   aggregate_convergence, aggregate_bins_mean,              evaluate_models.py:279-379
               sign_test_paired, fdr_bh
  variants     variant_grid                                             evaluate_models.py:382-455
  drivers      eval_variant_cv, run_ablation                            evaluate_models.py:194-276, 708-862

Plots (matplotlib) and the Optuna search are not part of it.
"""
from __future__ import annotations

import csv
import json
import math
import os
import time
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .als import ALS
from .helpers import DEFAULT_RANDOM_STATE, ES_MIN_ITERS, ES_TOL, make_config, normalize_params

N_POP_BINS = 5                     # evaluate_models.py:108
POP_BIN_STRATEGY = "quantile"      # evaluate_models.py:109


# ------------------------------------------------------------------------------- ratings container
@dataclass
class CooRatings:
    """Observed ratings, row-major sorted; `flat = rows * n + cols` is the reference's index space
    (create_folds.py:76, tune_params.py:165-166)."""
    rows: np.ndarray
    cols: np.ndarray
    vals: np.ndarray
    shape: Tuple[int, int]

    def __post_init__(self):
        m, n = self.shape
        flat = self.rows.astype(np.int64) * n + self.cols.astype(np.int64)
        order = np.argsort(flat, kind="stable")
        self.rows = np.asarray(self.rows)[order].astype(np.int64)
        self.cols = np.asarray(self.cols)[order].astype(np.int64)
        self.vals = np.asarray(self.vals, dtype=np.float64)[order]
        self.flat = flat[order]

    @classmethod
    def from_dense(cls, R: np.ndarray) -> "CooRatings":
        r, c = np.nonzero(~np.isnan(R))
        return cls(r, c, R[r, c], R.shape)

    def positions_of(self, flat_idx: np.ndarray) -> np.ndarray:
        pos = np.searchsorted(self.flat, flat_idx)
        if pos.size and (pos.max() >= self.flat.size or np.any(self.flat[pos] != flat_idx)):
            raise ValueError("flat index that is not an observed entry")
        return pos


# ------------------------------------------------------------------------------------------ folds
def make_entrywise_folds(ratings: CooRatings, n_splits: int = 5, seed: int = 42,
                         shuffle: bool = True) -> List[np.ndarray]:
    """K disjoint validation sets of flat indices over the observed entries.  Same generator, same
    shuffle of the same ascending flat-index array as the reference, hence identical folds."""
    obs = ratings.flat.copy()
    if shuffle:
        np.random.default_rng(seed).shuffle(obs)
    return [np.asarray(part, dtype=np.int64) for part in np.array_split(obs, n_splits)]


def save_folds_npz(path: str, folds: Sequence[np.ndarray], shape: Tuple[int, int], seed: int) -> None:
    """Reference fold-file format: `shape`, `seed`, `fold0..foldK-1` (int64 flat indices)."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    payload = {f"fold{i}": np.asarray(f, dtype=np.int64) for i, f in enumerate(folds)}
    np.savez_compressed(path, shape=np.asarray(shape, dtype=np.int64),
                        seed=np.asarray([seed], dtype=np.int64), **payload)


def load_folds_npz(path: str) -> Tuple[List[np.ndarray], Tuple[int, int], int]:
    with np.load(path, allow_pickle=False) as z:
        keys = sorted((k for k in z.files if k.startswith("fold")), key=lambda s: int(s[4:]))
        folds = [z[k].astype(np.int64) for k in keys]
        shape = tuple(int(v) for v in z["shape"])
        seed = int(z["seed"][0])
    return folds, shape, seed


def train_valid_split(ratings: CooRatings, folds: Sequence[np.ndarray], k: int):
    """Fold k as validation: ((rows, cols, vals) train, (rows, cols, vals) valid, val_idx)."""
    val_idx = np.asarray(folds[k], dtype=np.int64)
    vpos = ratings.positions_of(val_idx)
    keep = np.ones(ratings.flat.size, dtype=bool)
    keep[vpos] = False
    train = (ratings.rows[keep], ratings.cols[keep], ratings.vals[keep])
    valid = (ratings.rows[vpos], ratings.cols[vpos], ratings.vals[vpos])
    return train, valid, val_idx


# ---------------------------------------------------------------------------------------- metrics
def rmse_at(y_true: np.ndarray, y_pred: np.ndarray) -> float:
    """RMSE over already-gathered values; NaN on an empty set (tune_params.py:163-164)."""
    if np.size(y_true) == 0:
        return float("nan")
    return float(np.sqrt(np.mean((np.asarray(y_true) - np.asarray(y_pred)) ** 2)))


def popularity_bins(item_counts: np.ndarray, n_bins: int = N_POP_BINS,
                    strategy: str = POP_BIN_STRATEGY) -> Tuple[np.ndarray, np.ndarray]:
    """Item popularity bins from per-item rating counts: (bin per item, bin edges)."""
    counts = np.asarray(item_counts, dtype=float)
    if strategy == "quantile":
        edges = np.quantile(counts, np.linspace(0, 1, n_bins + 1))
    elif strategy == "uniform":
        edges = np.linspace(float(counts.min()), float(counts.max()), n_bins + 1)
    else:
        raise ValueError(f"Unknown popularity binning strategy '{strategy}'")
    edges = np.array(edges, dtype=float)
    for i in range(1, edges.size):                  # strictly increasing edges
        if edges[i] <= edges[i - 1]:
            edges[i] = edges[i - 1] + 1e-9
    which = np.searchsorted(edges, counts, side="right") - 1
    return np.clip(which, 0, n_bins - 1).astype(int), edges


def split_by_popularity(val_idx: np.ndarray, shape: Tuple[int, int], item_bin: np.ndarray,
                        n_bins: int) -> List[np.ndarray]:
    b = item_bin[np.asarray(val_idx) % shape[1]]
    return [np.asarray(val_idx)[b == j] for j in range(n_bins)]


# ------------------------------------------------------------------------------------- statistics
def aggregate_convergence(curves: Sequence[Sequence[float]]) -> Dict[str, Any]:
    """Mean / std of the train-RMSE curves of the folds, iteration by iteration (NaN padded)."""
    if not curves:
        return {"iters": [], "rmse_mean": [], "rmse_std": [], "n_folds": 0}
    width = max(len(c) for c in curves)
    table = np.full((len(curves), width), np.nan)
    for j, c in enumerate(curves):
        table[j, :len(c)] = c
    return {"iters": list(range(1, width + 1)), "rmse_mean": np.nanmean(table, axis=0).tolist(),
            "rmse_std": np.nanstd(table, axis=0).tolist(), "n_folds": len(curves)}


def aggregate_bins_mean(fold_bin_rmse: Sequence[Dict[str, float]]) -> Dict[str, float]:
    if not fold_bin_rmse:
        return {}
    return {k: float(np.nanmean([d[k] for d in fold_bin_rmse])) for k in sorted(fold_bin_rmse[0])}


def sign_test_paired(x: Sequence[float], y: Sequence[float]) -> float:
    """Exact two-sided sign test on the paired differences (ties dropped)."""
    d = [a - b for a, b in zip(x, y) if not np.isclose(a - b, 0.0)]
    n = len(d)
    if n == 0:
        return 1.0
    pos = sum(1 for v in d if v > 0)
    total = 2 ** n
    lower = sum(math.comb(n, i) for i in range(pos + 1)) / total             # P(X <= pos)
    upper = 1.0 - (sum(math.comb(n, i) for i in range(pos)) / total) if pos > 0 else 1.0   # P(X >= pos)
    return float(min(1.0, 2.0 * min(lower, upper)))


def fdr_bh(pvals: Sequence[float]) -> List[float]:
    """Benjamini-Hochberg adjusted p-values, in the input order."""
    m = len(pvals)
    if m == 0:
        return []
    p = np.asarray(pvals, dtype=float)
    order = np.argsort(p)
    scaled = p[order] * m / np.arange(1, m + 1)
    scaled = np.minimum.accumulate(scaled[::-1])[::-1]
    out = np.empty(m)
    out[order] = np.clip(scaled, 0.0, 1.0)
    return out.tolist()


# --------------------------------------------------------------------------------------- variants
def variant_grid(best_params: Dict[str, Any], feature_names: List[str]) -> List[Tuple[str, Dict[str, Any]]]:
    """Baseline plus controlled removals: no_features / only_<f> / no_graph / graph_feature=<f> /
    no_pop_reg, de-duplicated by parameter signature (last name wins, as in the reference)."""
    base = dict(best_params)
    out: List[Tuple[str, Dict[str, Any]]] = [("full", base)]
    alpha = float(base.get("alpha", 0.0))
    graph_on = alpha > 0.0 and base.get("graph_feature", "__none__") in feature_names
    used = {f: float(base.get(f"lambda_w_{f}", 0.0)) > 0.0 for f in feature_names}
    if any(used.values()):
        p = dict(base)
        p.update({f"lambda_w_{f}": 0.0 for f in feature_names})
        out.append(("no_features", p))
        for f in feature_names:
            if used[f]:
                q = dict(base)
                q.update({f"lambda_w_{g}": 0.0 for g in feature_names})
                q[f"lambda_w_{f}"] = float(base.get(f"lambda_w_{f}", 0.0))
                out.append((f"only_{f}", q))
    if graph_on:
        p = dict(base)
        p["alpha"], p["graph_feature"] = 0.0, "__none__"
        out.append(("no_graph", p))
        for f in feature_names:
            if f != base.get("graph_feature"):
                q = dict(base)
                q["alpha"], q["graph_feature"] = alpha, f
                out.append((f"graph_feature={f}", q))
    if base.get("pop_reg_mode", None) is not None:
        p = dict(base)
        p["pop_reg_mode"] = None
        out.append(("no_pop_reg", p))
    seen: Dict[Tuple, Tuple[str, Dict[str, Any]]] = {}
    for name, p in out:
        seen[tuple(sorted(p.items(), key=lambda kv: kv[0]))] = (name, p)
    return list(seen.values())


# ---------------------------------------------------------------------------------------- drivers
@dataclass
class AblationResultRow:
    variant: str
    rmse_mean: float
    rmse_std: float
    time_mean: float
    time_std: float
    mean_iters: float
    early_stopped_folds: int
    target_n_iters: int
    es_tol: float
    es_min_iters: int
    rmse_bins: Dict[str, float]
    params: Dict[str, Any]
    p_raw: Optional[float] = None
    p_fdr: Optional[float] = None
    delta_mean: Optional[float] = None
    fold_rmse: List[float] = field(default_factory=list)


def eval_variant_cv(variant_name: str, ratings: CooRatings, features: Dict[str, np.ndarray],
                    folds: Sequence[np.ndarray], params: Dict[str, Any], item_bin: np.ndarray,
                    n_pop_bins: int, es_tol: Optional[float], es_min_iters: int,
                    convergence_curves: Dict[str, List[List[float]]], verbose_fit: int = 0,
                    als_kwargs: Optional[Dict[str, Any]] = None):
    """One fixed-parameter model across the folds (evaluate_models.py:194-276 on sparse data):
    returns (fold_rmse, fold_time, fold_bin_rmse, fold_iters).  The timed region is fit + the
    predictions the caller reads, as in the reference (there: fit + dense predict)."""
    params = normalize_params(dict(params), ratings.shape, list(features))
    cfg = make_config(params)
    lambda_w = {name: float(params.get(f"lambda_w_{name}", 0.0)) for name in features}
    fold_rmse, fold_time, fold_bins, fold_iters = [], [], [], []
    for k in range(len(folds)):
        (tr, tc, tv), (_, _, vv), val_idx = train_valid_split(ratings, folds, k)
        t0 = time.perf_counter()
        model = ALS(config=cfg, lambda_w=lambda_w, **(als_kwargs or {}))
        model.fit_coo(tr, tc, tv, ratings.shape, features=features, tol=es_tol, min_iters=es_min_iters,
                      verbose=verbose_fit)
        pred = model.predict_at(val_idx, features=features)
        t1 = time.perf_counter()
        curve = list(model.history.get("train_rmse", []))
        convergence_curves.setdefault(variant_name, []).append(curve)
        fold_rmse.append(rmse_at(vv, pred))
        fold_time.append(t1 - t0)
        fold_iters.append(len(curve))
        bins = item_bin[val_idx % ratings.shape[1]]
        fold_bins.append({f"rmse_pop_{b + 1}": rmse_at(vv[bins == b], pred[bins == b]) for b in range(n_pop_bins)})
    return fold_rmse, fold_time, fold_bins, fold_iters


def row_to_dict(r: AblationResultRow, feature_names: List[str]) -> Dict[str, Any]:
    d: Dict[str, Any] = {"variant": r.variant, "rmse_mean": r.rmse_mean, "rmse_std": r.rmse_std,
                         "time_mean": r.time_mean, "time_std": r.time_std, "mean_iters": r.mean_iters,
                         "early_stopped_folds": r.early_stopped_folds, "target_n_iters": r.target_n_iters,
                         "es_tol": r.es_tol, "es_min_iters": r.es_min_iters, "p_raw": r.p_raw,
                         "p_fdr": r.p_fdr, "delta_mean": r.delta_mean}
    d.update(sorted(r.rmse_bins.items()))
    for key in ("alpha", "graph_feature", "pop_reg_mode", "n_factors", "n_iters", "lambda_u", "lambda_v",
                "lambda_bu", "lambda_bi", "update_w_every"):
        if key in r.params:
            d[f"param_{key}"] = r.params[key]
    for f in feature_names:
        d[f"param_lambda_w_{f}"] = r.params.get(f"lambda_w_{f}")
    return d


def run_ablation(ratings, folds, best_params: Dict[str, Any], features: Dict[str, np.ndarray],
                 out_dir: Optional[str] = None, n_pop_bins: int = N_POP_BINS,
                 es_tol: Optional[float] = None, es_min_iters: Optional[int] = None,
                 verbose_fit: int = 0, folds_seed: int = DEFAULT_RANDOM_STATE,
                 als_kwargs: Optional[Dict[str, Any]] = None) -> Tuple[List[AblationResultRow], Dict[str, Any]]:
    """Ablation study on frozen folds (evaluate_models.py:708-862 without the plots).

    `ratings`: CooRatings, a dense NaN array, or a path to the reference's ratings `.npy`;
    `folds`: list of flat-index arrays or a path to a fold file; `best_params`: dict (either the
    raw dict or {"params": {...}}) or a path to the JSON.  Writes `<out_dir>/ablations/ablations.csv`,
    `ablations.json` and `convergence/<variant>.json` when `out_dir` is given.
    """
    if isinstance(ratings, str):
        ratings = np.load(ratings)
    if isinstance(ratings, np.ndarray):
        ratings = CooRatings.from_dense(ratings)
    if isinstance(folds, str):
        folds, fshape, folds_seed = load_folds_npz(folds)
        if tuple(fshape) != tuple(ratings.shape):
            raise AssertionError("Folds were built for a different matrix shape.")
    if isinstance(best_params, str):
        with open(best_params) as fh:
            best_params = json.load(fh)
    best_params = dict(best_params["params"]) if "params" in best_params else dict(best_params)
    es_tol = float(ES_TOL if es_tol is None else es_tol)
    es_min_iters = int(ES_MIN_ITERS if es_min_iters is None else es_min_iters)
    counts = np.bincount(ratings.cols, minlength=ratings.shape[1])
    item_bin, edges = popularity_bins(counts, n_pop_bins, POP_BIN_STRATEGY)
    feature_names = list(features)
    rows: List[AblationResultRow] = []
    curves: Dict[str, List[List[float]]] = {}
    for name, params in variant_grid(best_params, feature_names):
        f_rmse, f_time, f_bins, f_iters = eval_variant_cv(
            name, ratings, features, folds, params, item_bin, n_pop_bins, es_tol, es_min_iters, curves,
            verbose_fit=verbose_fit, als_kwargs=als_kwargs)
        target = int(params.get("n_iters", 0))
        rows.append(AblationResultRow(
            variant=name, rmse_mean=float(np.mean(f_rmse)),
            rmse_std=float(np.std(f_rmse, ddof=1)) if len(f_rmse) > 1 else 0.0,
            time_mean=float(np.mean(f_time)),
            time_std=float(np.std(f_time, ddof=1)) if len(f_time) > 1 else 0.0,
            mean_iters=float(np.mean(f_iters)), early_stopped_folds=int(sum(i < target for i in f_iters)),
            target_n_iters=target, es_tol=es_tol, es_min_iters=es_min_iters,
            rmse_bins=aggregate_bins_mean(f_bins), params=params, fold_rmse=list(f_rmse)))
    base = next((r for r in rows if r.variant == "full"), None)
    if base is not None:
        others = [r for r in rows if r.variant != "full"]
        for r in others:
            r.p_raw = sign_test_paired(r.fold_rmse, base.fold_rmse)
            r.delta_mean = float(np.mean(np.asarray(r.fold_rmse) - np.asarray(base.fold_rmse)))
        for r, adj in zip(others, fdr_bh([r.p_raw for r in others])):
            r.p_fdr = float(adj)
    payload = {"seed": DEFAULT_RANDOM_STATE, "matrix_shape": [int(ratings.shape[0]), int(ratings.shape[1])],
               "folds_seed": int(folds_seed), "feature_names": feature_names, "n_pop_bins": int(n_pop_bins),
               "pop_bin_edges": [float(e) for e in edges], "es_tol": es_tol, "es_min_iters": es_min_iters,
               "variants_evaluated": [r.variant for r in rows], "best_params_used": best_params,
               "results": [row_to_dict(r, feature_names) for r in rows]}
    if out_dir is not None:
        base_dir = os.path.join(out_dir, "ablations")
        os.makedirs(os.path.join(base_dir, "convergence"), exist_ok=True)
        dict_rows = payload["results"]
        cols: List[str] = []
        for d in dict_rows:
            cols.extend(c for c in d if c not in cols)
        with open(os.path.join(base_dir, "ablations.csv"), "w", newline="") as fh:
            w = csv.DictWriter(fh, fieldnames=cols)
            w.writeheader()
            w.writerows(dict_rows)
        with open(os.path.join(base_dir, "ablations.json"), "w") as fh:
            json.dump(payload, fh, indent=2)
        for variant, cs in curves.items():
            with open(os.path.join(base_dir, "convergence", f"{variant}.json"), "w") as fh:
                json.dump(aggregate_convergence(cs), fh, indent=2)
    return rows, payload
