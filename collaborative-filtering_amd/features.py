"""Item-feature normalisation - host-side mirror of the reference's
`scripts/prepare_features.py` (`normalize_feature` :131-201, `normalize_features_dict` :204-233).

Same names, arguments, defaults and error behaviour; the arithmetic is numpy in the input's dtype
followed by ONE cast to `dtype` (float32 by default - the dtype the fit's HBM layout stores features
in).  O(n d), runs once per dataset: plumbing around the hot path, deliberately not a kernel.
Pinned by fixtures generated from the unmodified reference (`tests/golden/feat_norm_*.npz`).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Mapping, Optional

import numpy as np

DEFAULT_DTYPE = "float32"
DEFAULT_EPS = 1e-8


def _unit_rows(X: np.ndarray, eps: float, power: int) -> np.ndarray:
    """Rows scaled to unit L1 (power 1) or L2 (power 2) size; sizes below eps are clamped (:95-106)."""
    size = np.abs(X).sum(axis=1, keepdims=True) if power == 1 else np.sqrt((X * X).sum(axis=1, keepdims=True))
    return X / np.maximum(size, eps)


def _standardise_cols(X: np.ndarray, eps: float) -> np.ndarray:
    """(x - mean) / std per column; a (near-)constant column keeps std 1, non-finite results -> 0 (:109-116)."""
    centre = X.mean(axis=0, keepdims=True)
    spread = X.std(axis=0, keepdims=True)
    out = (X - centre) / np.where(spread < eps, 1.0, spread)
    out[~np.isfinite(out)] = 0.0
    return out


def _unit_range_cols(X: np.ndarray, eps: float) -> np.ndarray:
    """Columns mapped to [0, 1]; ranges below eps are clamped (:119-124)."""
    low = X.min(axis=0, keepdims=True)
    return (X - low) / np.maximum(X.max(axis=0, keepdims=True) - low, eps)


_METHODS: Dict[str, Optional[Callable[[np.ndarray, float], np.ndarray]]] = {
    "none": None,
    "row_l1": lambda X, eps: _unit_rows(X, eps, 1),
    "row_l2": lambda X, eps: _unit_rows(X, eps, 2),
    "col_zscore": _standardise_cols,
    "col_minmax": _unit_range_cols,
}


def _fill_with_col_median(X: np.ndarray) -> None:
    """In place: +-inf and NaN -> the column's median over its finite entries (0 for an empty column) (:82-92)."""
    bad = ~np.isfinite(X)
    if not bad.any():
        return
    X[bad] = np.nan
    med = np.nanmedian(X, axis=0)
    med = np.where(np.isfinite(med), med, 0.0)
    X[bad] = med[np.nonzero(bad)[1]]


def normalize_feature(X: np.ndarray, method: str = "none", *, impute: str = "none", eps: float = DEFAULT_EPS,
                      dtype: str = DEFAULT_DTYPE, copy: bool = True) -> np.ndarray:
    """One (n_items,) or (n_items, d) feature array -> (n_items, d) array of `dtype`.

    method: "none" | "row_l1" | "row_l2" | "col_zscore" | "col_minmax";
    impute: "none" (NaN/Inf raise ValueError) | "col_median"."""
    if method not in _METHODS:
        raise ValueError(f"Unknown method '{method}'.")
    if impute not in ("none", "col_median"):
        raise ValueError(f"Unknown impute '{impute}'.")
    X = X.reshape(-1, 1) if X.ndim == 1 else X
    if copy:
        X = X.copy()
    if impute == "col_median":
        _fill_with_col_median(X)
    elif not np.isfinite(X).all():
        raise ValueError("Input feature contains NaN/Inf and impute='none'.")
    fn = _METHODS[method]
    return (X if fn is None else fn(X, eps)).astype(dtype, copy=False)


def normalize_features_dict(features: Mapping[str, np.ndarray], *, method: str = "none", impute: str = "none",
                            eps: float = DEFAULT_EPS, dtype: str = DEFAULT_DTYPE, copy: bool = True,
                            per_feature_overrides: Optional[Mapping[str, Mapping[str, Any]]] = None
                            ) -> Dict[str, np.ndarray]:
    """`normalize_feature` over a {name: array} dict with shared defaults and per-feature overrides."""
    shared = dict(method=method, impute=impute, eps=eps, dtype=dtype, copy=copy)
    overrides = per_feature_overrides or {}
    return {name: normalize_feature(X, **{**shared, **dict(overrides.get(name, {}))})
            for name, X in features.items()}
