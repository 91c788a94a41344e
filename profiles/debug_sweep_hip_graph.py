import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from collaborative_filtering_amd import cv, sweep
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
item_bin, _ = cv.popularity_bins(np.bincount(ratings.cols, minlength=n), 5)
def harness(kw):
    return [cv.eval_variant_cv("x", ratings, feats, folds, dict(p), item_bin, 5, cv.ES_TOL, cv.ES_MIN_ITERS, {}, als_kwargs=kw) for p in params]
E = harness({}); H = harness({"hip_graph": True}); H2 = harness({"hip_graph": True})
DE = sweep.SweepDriver(ratings, feats, folds).run([dict(p) for p in params])["trials"]
DH = sweep.SweepDriver(ratings, feats, folds, als_kwargs={"hip_graph": True}).run([dict(p) for p in params])["trials"]
for t, p in enumerate(params):
    e, h, h2, de, dh = E[t][0], H[t][0], H2[t][0], DE[t]["fold_rmse"], DH[t]["fold_rmse"]
    print(t, p["n_factors"], "graph" if p["alpha"] > 0 else "-", "iters E", E[t][3], "H", H[t][3], "DH", DH[t]["iters_per_fold"],
          "E==DE", e == de, "E==H", e == h, "H==H2", h == h2, "E==DH", e == dh,
          "max|E-H| %.2e" % max(abs(a - b) for a, b in zip(e, h)), "max|E-DH| %.2e" % max(abs(a - b) for a, b in zip(e, dh)))
