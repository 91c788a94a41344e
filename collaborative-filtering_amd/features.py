"""Item-feature normalisation - mirror of the reference's
`scripts/prepare_features.py` (`normalize_feature` :131-201, `normalize_features_dict` :204-233).

Same names, arguments, defaults and error behaviour.  Two paths:
  * host (`normalize_feature`, `normalize_features_dict`): numpy in the input's dtype followed by ONE cast to
    `dtype` - any dtype, no GPU needed;
  * device (`normalize_feature_device`, or `device=` on the dict form): float64 input, float32 result left in
    HBM as a torch tensor, arithmetic in the kernels of csrc/features.hip (`als_normalize_features`), which sum
    in numpy's order - results are bitwise those of the reference (tests/test_gpu_features.py).
Both are pinned by fixtures generated from the unmodified reference (`tests/golden/feat_norm_*.npz`).
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


_METHOD_CODES = {"none": 0, "row_l1": 1, "row_l2": 2, "col_zscore": 3, "col_minmax": 4}


def normalize_feature_device(X, method: str = "none", *, impute: str = "none", eps: float = DEFAULT_EPS,
                             device="cuda:0"):
    """`normalize_feature` on the GPU: X (n_items,) or (n_items, d) float64 (numpy array or torch tensor) ->
    float32 torch tensor (n_items, d) on `device`.  Same methods, imputation and errors as the host form.
    (`ALS.fit` takes host arrays: hand it `.cpu().numpy()` of the result.)"""
    import ctypes as C

    import torch

    from . import _hip
    if method not in _METHODS:
        raise ValueError(f"Unknown method '{method}'.")
    if impute not in ("none", "col_median"):
        raise ValueError(f"Unknown impute '{impute}'.")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("normalize_feature_device needs a ROCm device; use normalize_feature on the host")
    Xt = torch.from_numpy(np.ascontiguousarray(X)) if isinstance(X, np.ndarray) else X
    if Xt.dtype != torch.float64:
        raise TypeError("the device path computes in float64 like the reference does on float64 input; "
                        f"got {Xt.dtype} (use normalize_feature for other dtypes)")
    Xt = Xt.reshape(-1, 1) if Xt.dim() == 1 else Xt
    Xt = Xt.to(dev).contiguous()
    lib = _hip.load()
    with torch.cuda.device(dev):
        n, d = Xt.shape
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        work = torch.empty(3 * d, dtype=torch.float64, device=dev)
        if impute == "col_median":          # in place on the device copy (the caller's array is never touched)
            if Xt.data_ptr() == (X.data_ptr() if torch.is_tensor(X) else 0):
                Xt = Xt.clone()
            rc = lib.als_impute_col_median(n, d, C.c_void_p(Xt.data_ptr()), C.c_void_p(work.data_ptr()), stream)
            if rc != 0:
                raise RuntimeError(f"als_impute_col_median failed with status {rc}")
        out = torch.empty(n, d, dtype=torch.float32, device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        rc = lib.als_normalize_features(n, d, C.c_void_p(Xt.data_ptr()), _METHOD_CODES[method], float(eps),
                                        C.c_void_p(out.data_ptr()), C.c_void_p(work.data_ptr()),
                                        C.c_void_p(status.data_ptr()), stream)
        if rc != 0:
            raise RuntimeError(f"als_normalize_features failed with status {rc}")
        if int(status.item()) & 1:
            raise ValueError("Input feature contains NaN/Inf and impute='none'.")
    return out


def normalize_features_dict(features: Mapping[str, np.ndarray], *, method: str = "none", impute: str = "none",
                            eps: float = DEFAULT_EPS, dtype: str = DEFAULT_DTYPE, copy: bool = True,
                            per_feature_overrides: Optional[Mapping[str, Mapping[str, Any]]] = None,
                            device=None) -> Dict[str, Any]:
    """`normalize_feature` over a {name: array} dict with shared defaults and per-feature overrides.
    device (extra, keyword-only): a ROCm device -> every feature through `normalize_feature_device`
    (float32 tensors in HBM; `dtype` / `copy` do not apply)."""
    shared = dict(method=method, impute=impute, eps=eps, dtype=dtype, copy=copy)
    overrides = per_feature_overrides or {}
    if device is not None:
        out = {}
        for name, X in features.items():
            kw = {**shared, **dict(overrides.get(name, {}))}
            out[name] = normalize_feature_device(X, kw["method"], impute=kw["impute"], eps=kw["eps"], device=device)
        return out
    return {name: normalize_feature(X, **{**shared, **dict(overrides.get(name, {}))})
            for name, X in features.items()}
