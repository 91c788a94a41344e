#!/usr/bin/env python3
"""Captured-graph replay WITH the per-iteration read-back of early stopping (ALS_GRAPH_EARLY_STOP=1): is it
bitwise the eager fit?  The read-back is ONE contiguous 128-byte device-to-host copy per iteration (history row +
status words, als._Engine.ctrl); profiles/debug_sweep_hip_graph.py showed that several small copies per iteration
between replays corrupt later replays on ROCm 7.2, a single contiguous one did not in 30 iterations.  This script
runs many more: reference-scale fits (610 x 4980) of several model variants, 120 iterations each with a
tolerance that never triggers (so every iteration is replayed and read back), plus fits that do stop early.
Prints one line per fit; exit code 1 on any difference."""
import os
import sys
import time

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


def params(k, graph, features, n_iters, uw=5):
    p = {"n_iters": n_iters, "lambda_bu": 3.0, "lambda_bi": 2.0, "update_w_every": uw, "S_eps": 1e-8, "n_factors": k,
         "lambda_u": 8.74, "lambda_v": 7.02, "pop_reg_mode": "inverse_sqrt", "alpha": 0.83 if graph else 0.0,
         "graph_feature": "genres" if graph else None, "S_topk": 50, "lambda_w_genres": 31.1, "lambda_w_years": 28.3}
    return normalize_params(p, (m, n), list(feats) if features else [])


def fit(p, use_feats, hip, tol, min_iters):
    cfg = make_config(p)
    lw = {f: float(p[f"lambda_w_{f}"]) for f in feats} if use_feats else None
    md = ALS(cfg, lambda_w=lw, hip_graph=hip)
    t0 = time.perf_counter()
    md.fit_coo(rows, cols, vals, (m, n), features=feats if use_feats else None, tol=tol, min_iters=min_iters, verbose=0)
    torch.cuda.synchronize()
    return md, time.perf_counter() - t0


bad = 0
os.environ["ALS_GRAPH_EARLY_STOP"] = "1"
for k, graph, use_feats, n_iters, tol, min_iters in [
        (16, True, True, 120, -1.0, 5), (64, True, True, 120, -1.0, 5), (64, False, False, 120, -1.0, 5),
        (32, True, False, 120, -1.0, 5), (64, False, True, 120, -1.0, 5), (128, True, True, 60, -1.0, 5),
        (64, True, True, 100, 1e-4, 10), (16, False, True, 100, 1e-4, 10), (32, True, False, 100, 1e-5, 10)]:
    p = params(k, graph, use_feats, n_iters)
    e, te = fit(p, use_feats, False, tol, min_iters)
    for rep in range(2):
        g, tg = fit(p, use_feats, True, tol, min_iters)
        he, hg = np.asarray(e.history["train_rmse"]), np.asarray(g.history["train_rmse"])
        same = (len(he) == len(hg) and np.array_equal(he, hg) and np.array_equal(e.U, g.U) and np.array_equal(e.V, g.V)
                and np.array_equal(e.b_i, g.b_i))
        bad += not same
        print(f"k={k} graph={graph} features={use_feats} tol={tol}: {len(he)} iterations eager {te * 1e3:.0f} ms, "
              f"{len(hg)} replayed {tg * 1e3:.0f} ms, bitwise equal: {same}", flush=True)
sys.exit(1 if bad else 0)
