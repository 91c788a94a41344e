#!/usr/bin/env python3
"""Fits per second at the reference's own scale (610 x 4980, 100K ratings, 3 folds): the per-fit harness
(cv.eval_variant_cv: fit_coo + predict_at, everything rebuilt per fit) against sweep.SweepDriver (ratings,
features, graphs, schedules, initial factors resident across fits).  Same parameter sets, identical fold scores.
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collaborative_filtering_amd import cv, sweep  # noqa: E402
from tests.synth import make_features, make_ratings  # noqa: E402

m, n, nnz = 610, 4980, 100000
rows, cols, vals = make_ratings(m, n, nnz, 5)
G, Y = make_features(n, 6)
feats = {"genres": G, "years": Y}
ratings = cv.CooRatings(rows, cols, vals, (m, n))
folds = cv.make_entrywise_folds(ratings, n_splits=3, seed=42)
rng = np.random.default_rng(0)
base = {"n_iters": 30, "lambda_bu": 3.0, "lambda_bi": 2.0, "update_w_every": 5, "S_eps": 1e-8}
params = []
for t in range(12):                      # a dozen "trials" the way the tuner's search space draws them
    graph = t % 2 == 0
    params.append(dict(base, n_factors=int(rng.choice([16, 32, 64, 100])), lambda_u=float(10 ** rng.uniform(-1, 2)),
                       lambda_v=float(10 ** rng.uniform(-1, 2)), pop_reg_mode=[None, "inverse_sqrt"][t % 3 == 0],
                       alpha=float(rng.uniform(0.1, 2.0)) if graph else 0.0,
                       graph_feature="genres" if graph else "__none__", S_topk=int(rng.choice([20, 50])),
                       lambda_w_genres=float(10 ** rng.uniform(-1, 1.5)), lambda_w_years=float(10 ** rng.uniform(-1, 1.5))))
item_bin, _ = cv.popularity_bins(np.bincount(ratings.cols, minlength=n), 5)

out = {}
for hip_graph in (False, True):  # eager launches / captured-graph replay (early stopping reads between replays)
    kw = {"hip_graph": hip_graph}
    # warm both paths (library load, graph memoisation of the per-fit path)
    cv.eval_variant_cv("w", ratings, feats, folds, dict(params[0]), item_bin, 5, cv.ES_TOL, cv.ES_MIN_ITERS, {}, als_kwargs=kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ref = [cv.eval_variant_cv("x", ratings, feats, folds, dict(p), item_bin, 5, cv.ES_TOL, cv.ES_MIN_ITERS, {}, als_kwargs=kw)
           for p in params]
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    drv = sweep.SweepDriver(ratings, feats, folds, als_kwargs=kw)
    torch.cuda.synchronize()
    t_build = time.perf_counter() - t0
    drv.cv_score(dict(params[0]))
    drv.cv_score(dict(params[1]))
    drv.n_fits, drv.fit_seconds = 0, 0.0
    t0 = time.perf_counter()
    res = drv.run([dict(p) for p in params])
    t_drv = time.perf_counter() - t0
    same = all(t["fold_rmse"] == r[0] for t, r in zip(res["trials"], ref))
    nf = 3 * len(params)
    iters = sum(sum(t["iters_per_fold"]) for t in res["trials"])
    out["hip_graph" if hip_graph else "eager"] = {
        "fits": nf, "iterations_run": iters, "per_fit_harness_ms_per_fit": 1e3 * t_ref / nf,
        "sweep_driver_ms_per_fit": 1e3 * t_drv / nf, "sweep_driver_fits_per_s": nf / t_drv,
        "per_fit_harness_fits_per_s": nf / t_ref, "driver_build_ms": 1e3 * t_build,
        "fold_scores_identical": bool(same)}
print(json.dumps(out, indent=1))
