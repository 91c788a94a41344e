#!/usr/bin/env python3
"""Kept for the record: replaying the captured iteration graphs while the host reads the engine's
status words / history back between the launches.  Until the sweep's hipMemsetD32Async was replaced by a fill kernel
this went wrong from iteration 21 on (the cause was the memset node in two live graphs, not the reads); now every
mode prints a zero difference.  One reference-scale fit (610 x 4980, k = 16, features + Laplacian), 30 iterations,
eager against replay with different host reads after every iteration >= 9."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collaborative_filtering_amd import ALS, cv  # noqa: E402
from collaborative_filtering_amd.helpers import make_config, normalize_params  # noqa: E402
from tests.synth import make_features, make_ratings  # noqa: E402

m, n, nnz = 610, 4980, 100000
rows, cols, vals = make_ratings(m, n, nnz, 5)
G, Y = make_features(n, 6)
feats = {"genres": G, "years": Y}
ratings = cv.CooRatings(rows, cols, vals, (m, n))
folds = cv.make_entrywise_folds(ratings, n_splits=3, seed=42)
p = normalize_params({"n_iters": 30, "lambda_bu": 3.0, "lambda_bi": 2.0, "update_w_every": 5, "S_eps": 1e-8, "n_factors": 16,
                      "lambda_u": 8.74, "lambda_v": 7.02, "pop_reg_mode": None, "alpha": 0.83, "graph_feature": "genres",
                      "S_topk": 50, "lambda_w_genres": 31.1, "lambda_w_years": 28.3}, (m, n), list(feats))
cfg = make_config(p)
lw = {f: float(p[f"lambda_w_{f}"]) for f in feats}
(tr, tc, tv), _, _ = cv.train_valid_split(ratings, folds, 0)


def fit(hip, mode):
    md = ALS(cfg, lambda_w=lw, hip_graph=hip)
    csr, csc = __import__("collaborative_filtering_amd").layout.coo_to_sides(tr, tc, tv, (m, n))
    md._fit_sides(csr, csc, feats, None, 0, 0, None, run=False)
    eng = md._eng
    eng.be.compose_z(eng.V, eng.Xcat, eng.Wcat, eng.Z)
    for it in range(30):
        eng.iteration(it, 30)
        if it >= 9:
            if "status" in mode:
                eng._check_status()                      # three .item() reads
            if "slice" in mode:
                eng.hist[: it + 1, 0].cpu().numpy()      # strided view: a device temporary + a D2H copy
            if "row" in mode:
                eng.hist_row.cpu().numpy()               # contiguous 6 doubles: a D2H copy only
    torch.cuda.synchronize()
    h = eng.hist[:30, 0].cpu().numpy()
    eng._graphs.clear()          # as _Engine.run does: no captured graph may die during another engine's capture
    return h


e = fit(False, "")
for mode in ("", "status", "slice", "row", "status+slice", "status+row"):
    try:
        h = fit(True, mode)
    except Exception as ex:          # noqa: BLE001 - the corrupted state usually ends in "not positive definite"
        print(f"replay, host reads after every iteration >= 9: [{mode}]: the fit went wrong: {type(ex).__name__}: {ex}", flush=True)
        continue
    d = np.abs(e - h)
    print(f"replay, host reads after every iteration >= 9: [{mode or 'none'}]: max |eager - replay| = {d.max():.3e}, "
          f"first differing iteration {int(np.argmax(d > 0)) if (d > 0).any() else -1}", flush=True)
