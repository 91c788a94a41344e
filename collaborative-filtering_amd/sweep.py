"""Hyper-parameter sweep driver: many fits against ONE resident copy of the data.

The reference's tuner evaluates every trial by K-fold cross-validation (`scripts/tune_params.py:341-421`,
`_cv_score_single_trial`) and runs 150 trials x 3 folds (`:600-741`, `run_tuning`): per fit it rebuilds dense NaN
train / valid matrices (`scripts/create_folds.py:177-208`), the adjacency lists, the dense n x n similarity graph
and the dense m x n prediction.  At the reference's own scale a fit takes ~20 ms on the MI355X, of which a third
was per-fit host set-up and upload.  This driver removes that part:

  * the observed ratings go to HBM once (CSR by user + CSC by item); the train matrices of the K folds are cut
    out of them ON THE DEVICE with a keep mask over the resident entries (no COO -> CSR rebuild, no upload);
  * validation (user, item) index tensors and their true values stay on the device per fold;
  * features, the similarity graph of every distinct (feature, top-k, eps), its level schedule per fold, the task
    lists per (fold, k class) and the (seed, shape, k)-determined initial factors are built once (`als.FitCache`);
  * a fit is then engine allocation + iterations + one prediction launch.

Per-fold scores are those of `cv.eval_variant_cv` (same train entries in the same order, same kernels): identical
RMSE.  The Optuna study itself (sampler, pruner, plots) is not restated - `cv_score(params, trial=...)` takes any
object with `report(value, step)` / `should_prune()` / `set_user_attr(key, value)`, i.e. it is the body of the
reference's `objective` (`scripts/tune_params.py:657-670`), and `run(param_dicts)` evaluates a given list and writes
`<out>/tuning/<study>_{trials.csv, summary.json, best_params.json}` in the reference's artifact layout.
"""
from __future__ import annotations

import csv
import json
import os
import time
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from . import _hip, layout
from .als import ALS, FitCache, _SideDev
from .cv import CooRatings, rmse_at
from .helpers import DEFAULT_RANDOM_STATE, ES_MIN_ITERS, ES_TOL, make_config, normalize_params


class SweepPruned(Exception):
    """Raised by `cv_score` when the trial object asks for pruning (optuna.TrialPruned in the reference)."""


class _Fold:
    __slots__ = ("csr", "csc", "us", "is_", "truth", "val_idx", "cache", "nnz")


class SweepDriver:
    def __init__(self, ratings, features: Dict[str, np.ndarray], folds: Sequence[np.ndarray], *, device=None,
                 als_kwargs: Optional[Dict[str, Any]] = None):
        if isinstance(ratings, np.ndarray):
            ratings = CooRatings.from_dense(ratings)
        for name, X in features.items():                      # _assert_finite_features, tune_params.py:133-144
            if not np.isfinite(X).all():
                raise ValueError(f"Feature '{name}' contains infinite values.")
        self.ratings = ratings
        self.features = dict(features)
        self.shape = (int(ratings.shape[0]), int(ratings.shape[1]))
        self.dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.als_kwargs = dict(als_kwargs or {})
        self.als_kwargs.pop("device", None)
        m, n = self.shape
        dev = self.dev
        with torch.cuda.device(dev):
            # ---- the observed entries, once: CSR (row-major = the order of CooRatings) and CSC
            csr, csc = layout.coo_to_sides_native(_hip.load(), ratings.rows, ratings.cols, ratings.vals, self.shape)
            uptr = torch.from_numpy(csr.indptr).to(dev)
            uidx = torch.from_numpy(csr.indices).to(dev)
            uval = torch.from_numpy(csr.vals).to(dev)
            iptr = torch.from_numpy(csc.indptr).to(dev)
            iidx = torch.from_numpy(csc.indices).to(dev)
            ival = torch.from_numpy(csc.vals).to(dev)
            N = uidx.numel()
            rows_of = torch.repeat_interleave(torch.arange(m, device=dev), uptr[1:] - uptr[:-1])
            # CSR position of every CSC entry: CSC order is (item, user) ascending
            csc_src = torch.argsort(uidx.to(torch.int64) * m + rows_of)
            self.folds: List[_Fold] = []
            for k, val_idx in enumerate(folds):
                val_idx = np.asarray(val_idx, dtype=np.int64)
                vpos = torch.from_numpy(ratings.positions_of(val_idx)).to(dev)       # positions in CSR order
                keep = torch.ones(N, dtype=torch.bool, device=dev)
                keep[vpos] = False
                f = _Fold()
                f.val_idx = val_idx
                f.csr = self._masked_side(uptr, uidx, uval, keep, m, n)
                f.csc = self._masked_side(iptr, iidx, ival, keep[csc_src], n, m)
                f.us = rows_of[vpos].to(torch.int32)
                f.is_ = uidx[vpos].contiguous()
                f.truth = ratings.vals[ratings.positions_of(val_idx)]                # float64, host
                f.cache = FitCache()
                f.nnz = int(f.csr.vals.numel())
                self.folds.append(f)
        self._graphs: Dict[Any, Any] = {}      # (feature, topk, eps) -> CSR (ptr, idx, val, D), built once
        self.n_fits = 0
        self.fit_seconds = 0.0

    def _graph_for(self, cfg):
        """The similarity graph of this configuration (scripts/als.py:194-240), built once per distinct
        (feature, top-k, eps) with the reference's own numpy call sequence (identical tie handling)."""
        sim = cfg.graph.sim
        if cfg.graph.alpha <= 0.0 or sim is None or sim.feature_name not in self.features:
            return None
        key = (sim.feature_name, sim.topk, float(sim.eps))
        if key not in self._graphs:
            Sd = layout.build_similarity_dense(self.features[sim.feature_name], sim.topk, sim.eps)
            ptr, idx, val = layout.dense_graph_to_csr(Sd)
            self._graphs[key] = (ptr, idx, val.astype(np.float32), Sd.sum(axis=1).astype(np.float32))
        return self._graphs[key]

    @staticmethod
    def _masked_side(ptr, idx, val, keep, nrows, ncols) -> _SideDev:
        csum = torch.zeros(keep.numel() + 1, dtype=torch.int64, device=keep.device)
        csum[1:] = torch.cumsum(keep.to(torch.int64), 0)
        return _SideDev(nrows, ncols, csum[ptr].contiguous(), idx[keep].contiguous(), val[keep].contiguous())

    # ------------------------------------------------------------------------------------------------------
    def cv_score(self, params: Dict[str, Any], trial=None, verbose_fit: int = 0, es_tol: Optional[float] = ES_TOL,
                 es_min_iters: int = ES_MIN_ITERS) -> Dict[str, Any]:
        """Mean validation RMSE of one parameter set over the folds (`_cv_score_single_trial`)."""
        params = normalize_params(dict(params), self.shape, list(self.features))       # tune_params.py:660
        cfg = make_config(params)
        lambda_w = {name: float(params.get(f"lambda_w_{name}", 0.0)) for name in self.features}   # :328-337
        target = int(cfg.core.n_iters)
        fold_scores: List[float] = []
        iters: List[int] = []
        t_start = time.perf_counter()
        for i, f in enumerate(self.folds):
            model = ALS(config=cfg, lambda_w=lambda_w, device=self.dev, fit_cache=f.cache, **self.als_kwargs)
            model._fit_sides(f.csr, f.csc, self.features, es_tol, es_min_iters, verbose_fit, self._graph_for(cfg),
                             S_trusted=True)
            with torch.cuda.device(self.dev):
                pred = model._eng.predict_pairs(f.us, f.is_, self.features, features_of_fit=True).cpu().numpy().astype(np.float64)
            rmse = rmse_at(f.truth, pred)
            n_run = len(model.history.get("train_rmse", []))
            fold_scores.append(rmse)
            iters.append(n_run)
            self.n_fits += 1
            if trial is not None:
                trial.report(rmse, step=i)
                if trial.should_prune():
                    self.fit_seconds += time.perf_counter() - t_start
                    raise SweepPruned()
        self.fit_seconds += time.perf_counter() - t_start
        out = {"value": float(np.mean(fold_scores)), "fold_rmse": fold_scores, "iters_per_fold": iters,
               "mean_iters": float(np.mean(iters)), "early_stopped_folds": int(sum(n < target for n in iters)),
               "target_n_iters": target, "es_tol": float(es_tol) if es_tol is not None else None,
               "es_min_iters": int(es_min_iters), "params": params}
        if trial is not None:
            for key in ("es_tol", "es_min_iters", "target_n_iters", "iters_per_fold", "mean_iters",
                        "early_stopped_folds", "fold_rmse"):
                trial.set_user_attr(key, out[key])
        return out

    # ------------------------------------------------------------------------------------------------------
    def run(self, param_dicts: Iterable[Dict[str, Any]], out_dir: Optional[str] = None, study_name: str = "als_tuning",
            folds_seed: int = DEFAULT_RANDOM_STATE, on_trial: Optional[Callable[[int, Dict[str, Any]], None]] = None,
            **score_kw) -> Dict[str, Any]:
        """Evaluate a list of parameter dicts (what a sampler proposed) and write the tuning artifacts."""
        trials = []
        for number, p in enumerate(param_dicts):
            t0 = time.perf_counter()
            res = self.cv_score(p, **score_kw)
            res.update(number=number, duration_s=time.perf_counter() - t0, state="COMPLETE")
            trials.append(res)
            if on_trial is not None:
                on_trial(number, res)
        best = min(trials, key=lambda r: r["value"]) if trials else None
        summary = {"study_name": study_name, "seed": DEFAULT_RANDOM_STATE, "folds_seed": int(folds_seed),
                   "matrix_shape": list(self.shape), "feature_names": list(self.features), "es_tol": ES_TOL,
                   "es_min_iters": ES_MIN_ITERS, "n_trials": len(trials),
                   "n_complete": len(trials), "n_pruned": 0,
                   "best_value": best["value"] if best else None, "best_params": best["params"] if best else None,
                   "fits": self.n_fits, "fit_seconds": self.fit_seconds,
                   "fits_per_second": self.n_fits / self.fit_seconds if self.fit_seconds else None}
        if out_dir is not None and trials:
            tdir = os.path.join(out_dir, "tuning")
            os.makedirs(tdir, exist_ok=True)
            cols = ["number", "value", "state", "duration_s", "mean_iters", "early_stopped_folds", "target_n_iters",
                    "es_tol", "es_min_iters", "iters_per_fold", "fold_rmse"]
            pkeys = sorted({k for t in trials for k in t["params"]})
            with open(os.path.join(tdir, f"{study_name}_trials.csv"), "w", newline="") as fh:
                w = csv.writer(fh)
                w.writerow(cols + [f"params_{k}" for k in pkeys])
                for t in trials:
                    w.writerow([json.dumps(t[c]) if isinstance(t[c], list) else t[c] for c in cols]
                               + [t["params"].get(k) for k in pkeys])
            with open(os.path.join(tdir, f"{study_name}_summary.json"), "w") as fh:
                json.dump(summary, fh, indent=2)
            with open(os.path.join(tdir, f"{study_name}_best_params.json"), "w") as fh:
                json.dump({"value": best["value"], "params": best["params"]}, fh, indent=2)   # read by cv.run_ablation
        summary["trials"] = trials
        return summary
