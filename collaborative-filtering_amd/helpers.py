"""Caller-side glue of the reference harness, restated so that an
`evaluate_models`-style loop runs without optuna.

  rmse_on_indices   scripts/tune_params.py:147-167
  normalize_params  scripts/tune_params.py:237-278
  make_config       scripts/tune_params.py:281-322
  cholesky_solve    scripts/helpers.py:5-20 (host utility; the fit path solves on
                    the GPU - this exists for API parity with `scripts.helpers`)
Constants ES_TOL / ES_MIN_ITERS / DEFAULT_RANDOM_STATE: scripts/tune_params.py:114-121.
"""
from __future__ import annotations

from typing import Any, Dict, List

import numpy as np

from .als_config import ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig

ES_TOL: float = 1e-4
ES_MIN_ITERS: int = 10
DEFAULT_RANDOM_STATE: int = 42

_N_FACTORS_MIN = 1
_S_TOPK_MIN = 1
_UPDATE_W_EVERY_MIN = 1


def cholesky_solve(A: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Solve the SPD system A x = b on the host; LinAlgError if A is not SPD."""
    L = np.linalg.cholesky(np.asarray(A, dtype=np.float64))
    y = np.linalg.solve(L, np.asarray(b, dtype=np.float64))
    return np.linalg.solve(L.T, y)


def rmse_on_indices(R_true: np.ndarray, R_pred: np.ndarray, flat_idx: np.ndarray) -> float:
    """RMSE over flat indices u*n+i; NaN when the index set is empty."""
    if flat_idx.size == 0:
        return float("nan")
    diff = R_true.ravel()[flat_idx] - R_pred.ravel()[flat_idx]
    return float(np.sqrt(np.mean(diff ** 2)))


def normalize_params(params: Dict[str, Any], R_shape, feature_names: List[str]) -> Dict[str, Any]:
    """Clamp trial parameters to the data (in place, returns the dict):
    n_factors <= min(m, n); S_topk <= n-1; update_w_every <= n_iters; graph off
    when its feature is unavailable."""
    m, n = R_shape
    params["n_factors"] = max(_N_FACTORS_MIN, min(int(params["n_factors"]), min(m, n)))
    params["S_topk"] = max(_S_TOPK_MIN, min(int(params["S_topk"]), max(1, n - 1)))
    params["update_w_every"] = max(_UPDATE_W_EVERY_MIN,
                                   min(int(params["update_w_every"]), int(params["n_iters"])))
    if (not feature_names) or (params.get("graph_feature") not in feature_names):
        params["alpha"] = 0.0
        params["graph_feature"] = "__none__"
    return params


def make_config(params: Dict[str, Any]) -> ALSConfig:
    """Parameter dict -> ALSConfig; the graph is on only for alpha > 0 and a named feature."""
    core = CoreConfig(
        n_factors=int(params["n_factors"]), n_iters=int(params["n_iters"]),
        lambda_u=float(params["lambda_u"]), lambda_v=float(params["lambda_v"]),
        pop_reg_mode=params.get("pop_reg_mode", None), random_state=DEFAULT_RANDOM_STATE,
        update_w_every=int(params.get("update_w_every", _UPDATE_W_EVERY_MIN)))
    biases = BiasesConfig(lambda_bu=float(params.get("lambda_bu", core.lambda_u)),
                          lambda_bi=float(params.get("lambda_bi", core.lambda_v)))
    alpha = float(params.get("alpha", 0.0))
    gfeat = params.get("graph_feature", "__none__")
    if alpha <= 0.0 or gfeat == "__none__":
        graph = GraphConfig(alpha=0.0, sim=None)
    else:
        graph = GraphConfig(alpha=alpha, sim=GraphSimConfig(
            source="feature", feature_name=gfeat, metric="cosine",
            topk=int(params.get("S_topk", 50)), eps=float(params.get("S_eps", 1e-8))))
    return ALSConfig(core=core, biases=biases, graph=graph)
