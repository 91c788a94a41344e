import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from collaborative_filtering_amd import cv
from collaborative_filtering_amd.helpers import make_config, normalize_params
from collaborative_filtering_amd import ALS
from tests.synth import make_features, make_ratings
m, n, nnz = 610, 4980, 100000
rows, cols, vals = make_ratings(m, n, nnz, 5)
G, Y = make_features(n, 6)
feats = {"genres": G, "years": Y}
ratings = cv.CooRatings(rows, cols, vals, (m, n))
folds = cv.make_entrywise_folds(ratings, n_splits=3, seed=42)
rng = np.random.default_rng(0)
base = {"n_iters": 30, "lambda_bu": 3.0, "lambda_bi": 2.0, "update_w_every": 5, "S_eps": 1e-8}
params = []
for t in range(12):
    graph = t % 2 == 0
    params.append(dict(base, n_factors=int(rng.choice([16, 32, 64, 100])), lambda_u=float(10 ** rng.uniform(-1, 2)),
                       lambda_v=float(10 ** rng.uniform(-1, 2)), pop_reg_mode=[None, "inverse_sqrt"][t % 3 == 0],
                       alpha=float(rng.uniform(0.1, 2.0)) if graph else 0.0,
                       graph_feature="genres" if graph else "__none__", S_topk=int(rng.choice([20, 50])),
                       lambda_w_genres=float(10 ** rng.uniform(-1, 1.5)), lambda_w_years=float(10 ** rng.uniform(-1, 1.5))))
p = normalize_params(dict(params[4]), (m, n), list(feats))
print(p)
cfg = make_config(p)
lw = {f: float(p[f"lambda_w_{f}"]) for f in feats}
(tr, tc, tv), _, _ = cv.train_valid_split(ratings, folds, 0)

def fit(hip, use_feats=True, tol=None):
    md = ALS(cfg, lambda_w=lw if use_feats else None, hip_graph=hip)
    md.fit_coo(tr, tc, tv, (m, n), features=feats if use_feats else {"genres": G}, tol=tol, min_iters=10, verbose=0)
    return np.asarray(md.history["train_rmse"]), md

e, me = fit(False, tol=1e-4)
for mode in ("all", "gs", "w", "st", "dummy3", "sync3", "none"):
    os.environ["ALS_DBG_CHECK"] = mode
    h, md = fit(True, tol=1e-4)
    nn = min(len(e), len(h))
    d = np.abs(e[:nn] - h[:nn])
    print(mode, "iters", len(e), len(h), "max diff %.3e" % d.max(), "first", int(np.argmax(d > 0)) if (d > 0).any() else -1, flush=True)
