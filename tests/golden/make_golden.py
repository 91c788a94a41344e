#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz from the REAL reference.

Runs only in the build container (needs /root/reference).  It imports the
unmodified `scripts.als.ALS`, runs it on seeded synthetic inputs from
`tests/synth.py`, and stores inputs (COO triplets, features, config) and
outputs (U, V, W_f, b_u, b_i, mu, history, S as CSR, fold-0 validation
positions and test RMSE).  Only data is written; no reference source is
copied.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

from scripts.als import ALS                                     # noqa: E402
from scripts.als_config import (ALSConfig, BiasesConfig, CoreConfig,   # noqa: E402
                                GraphConfig, GraphSimConfig)
from tests.synth import make_features, make_folds, make_ratings, to_dense  # noqa: E402


def run_case(name, *, m, n, nnz, seed, k, n_iters, lambda_u, lambda_v,
             pop=None, bu=None, bi=None, update_w_every=5, feats=(), lambda_w=None,
             alpha=0.0, sim=None, tol=None, min_iters=5, empty_users=(),
             empty_items=(), store_factors=True, sample_rows=0, store_inputs=True, dup_cols=0):
    rows, cols, vals = make_ratings(m, n, nnz, seed, empty_users=empty_users,
                                    empty_items=empty_items)
    folds = make_folds(rows.size, 5, seed + 7)
    val_pos = folds[0]
    tr = np.ones(rows.size, dtype=bool)
    tr[val_pos] = False
    R_train = to_dense(rows[tr], cols[tr], vals[tr], (m, n))
    G, Y = make_features(n, seed + 11)
    if dup_cols:        # the first columns once more: an EXACTLY rank-deficient design for the W-step
        G = np.concatenate([G, G[:, :dup_cols]], axis=1)
    allf = {"genres": G, "years": Y}
    features = {f: allf[f] for f in feats}
    cfg = ALSConfig(
        core=CoreConfig(n_factors=k, n_iters=n_iters, lambda_u=lambda_u,
                        lambda_v=lambda_v, pop_reg_mode=pop, random_state=42,
                        update_w_every=update_w_every),
        biases=BiasesConfig(lambda_bu=bu, lambda_bi=bi),
        graph=GraphConfig(alpha=alpha,
                          sim=(GraphSimConfig(**sim) if sim else None)),
    )
    model = ALS(config=cfg, lambda_w=lambda_w)
    model.fit(R_train, features=features or None, tol=tol, min_iters=min_iters,
              verbose=0)
    R_hat = model.predict(features=features or None)
    flat = rows[val_pos] * n + cols[val_pos]
    pred_val = R_hat.ravel()[flat]
    test_rmse = float(np.sqrt(np.mean((vals[val_pos] - pred_val) ** 2)))
    out = {
        "shape": np.array([m, n], dtype=np.int64),
        "val_pos": val_pos.astype(np.int64),
        "mu": np.array([model.mu]), "test_rmse": np.array([test_rmse]),
        "pred_val": pred_val,
        "cfg": np.array(json.dumps(dict(
            m=m, n=n, nnz=nnz, seed=seed, n_factors=k, n_iters=n_iters,
            lambda_u=lambda_u, lambda_v=lambda_v, pop_reg_mode=pop,
            lambda_bu=bu, lambda_bi=bi, update_w_every=update_w_every,
            feats=list(feats), lambda_w=lambda_w, alpha=alpha, sim=sim,
            tol=tol, min_iters=min_iters, empty_users=list(empty_users),
            empty_items=list(empty_items), dup_cols=dup_cols))),
    }
    if store_inputs:      # otherwise the test regenerates them from the seed
        out.update(rows=rows.astype(np.int32), cols=cols.astype(np.int32),
                   vals=vals.astype(np.float32))   # half-stars: exact in f32
    for key, series in model.history.items():
        out["hist_" + key] = np.asarray(series, dtype=np.float64)
    if store_factors:
        out.update(U=model.U, V=model.V, b_u=model.b_u, b_i=model.b_i)
    else:
        sel_u = np.linspace(0, m - 1, sample_rows).astype(np.int64)
        sel_i = np.linspace(0, n - 1, sample_rows).astype(np.int64)
        out.update(sel_u=sel_u, sel_i=sel_i, U=model.U[sel_u], V=model.V[sel_i],
                   b_u=model.b_u, b_i=model.b_i)
    for f in feats:
        out["W_" + f] = model.W[f]
    if model.S is not None:
        S = model.S
        ptr = [0]
        idx, val = [], []
        for i in range(n):
            nz = np.flatnonzero(S[i])
            idx.append(nz)
            val.append(S[i, nz])
            ptr.append(ptr[-1] + nz.size)
        out["S_ptr"] = np.asarray(ptr, dtype=np.int64)
        out["S_idx"] = np.concatenate(idx).astype(np.int32)
        out["S_val"] = np.concatenate(val)
        out["S_D"] = S.sum(axis=1)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: iters={len(model.history['train_rmse'])} "
          f"train_rmse={model.history['train_rmse'][-1]:.6f} test_rmse={test_rmse:.6f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KB")


def folds_case():
    """Fold files and a train/valid split produced by the reference's scripts/create_folds.py."""
    from scripts.create_folds import (load_folds_npz, make_entrywise_folds, make_train_valid_split,
                                      save_folds_npz)
    rows, cols, vals = make_ratings(30, 20, 180, 201)
    R = to_dense(rows, cols, vals, (30, 20))
    folds = make_entrywise_folds(R, n_splits=5, seed=42, shuffle=True)
    path = os.path.join(HERE, "ref_folds_30x20.npz")
    save_folds_npz(path, folds, R.shape, 42)                      # the reference's own file format
    R_train, R_val, val_idx = make_train_valid_split(R, folds, 2)
    noshuf = make_entrywise_folds(R, n_splits=4, seed=7, shuffle=False)
    np.savez_compressed(os.path.join(HERE, "ref_folds_30x20_io.npz"),
                        rows=rows.astype(np.int32), cols=cols.astype(np.int32), vals=vals.astype(np.float32),
                        train_flat=np.flatnonzero(~np.isnan(R_train)).astype(np.int64),
                        train_vals=R_train[~np.isnan(R_train)], val_idx=val_idx.astype(np.int64),
                        val_vals=R_val.ravel()[val_idx],
                        **{f"noshuf{i}": f for i, f in enumerate(noshuf)})
    print("folds fixture:", [len(f) for f in folds])


def feature_norm_case():
    """Outputs of the reference's scripts/prepare_features.py on a seeded matrix with NaN / +-inf entries,
    an all-zero row, a constant column and an all-missing column (every method, both impute modes)."""
    from scripts.prepare_features import normalize_feature, normalize_features_dict
    rng = np.random.default_rng(301)
    X = rng.normal(size=(50, 6)) * 3.0 + 1.0
    X[:, 2] = 7.0                       # constant column
    X[:, 5] = np.nan                    # all-missing column
    X[3, 1], X[5, 0], X[6, 4] = np.nan, np.inf, -np.inf
    X[7, :5] = 0.0                      # zero row (row norms clamp to eps)
    clean = rng.normal(size=(50, 4))
    years = rng.integers(1950, 2020, size=50).astype(np.float64)
    out = {"X": X, "clean": clean, "years": years}
    for method in ("none", "row_l1", "row_l2", "col_zscore", "col_minmax"):
        out[f"imputed_{method}"] = normalize_feature(X, method, impute="col_median")
        out[f"clean_{method}"] = normalize_feature(clean, method)
        out[f"clean64_{method}"] = normalize_feature(clean, method, dtype="float64")
    d = normalize_features_dict({"genres": clean, "years": years}, method="none", impute="col_median",
                                per_feature_overrides={"genres": {"method": "row_l2"},
                                                       "years": {"method": "col_zscore"}})
    out["dict_genres"], out["dict_years"] = d["genres"], d["years"]
    np.savez_compressed(os.path.join(HERE, "feat_norm_50x6.npz"), **out)
    print("feature-normaliser fixture:", sorted(out)[:4], "...")


def feature_norm_wide_case():
    """Reference outputs on shapes that reach every branch of numpy's summation order (the device normaliser
    reproduces it): 21 columns (interleaved partial sums + tail), 150 columns (recursive halves), a single column
    of 5000 items (column statistics summed pairwise), 9 columns with missing entries (imputation)."""
    from scripts.prepare_features import normalize_feature
    rng = np.random.default_rng(302)
    mats = {
        "w21": rng.normal(size=(600, 21)) * 10.0 ** rng.uniform(-2, 2, size=(600, 21)),
        "w150": rng.normal(size=(120, 150)) * 10.0 ** rng.uniform(-2, 2, size=(120, 150)),
        "w1": rng.integers(1900, 2020, size=5000).astype(np.float64) + rng.normal(size=5000) * 1e-3,
        "w2": rng.normal(size=(700, 2)) * 5.0 + 2.0,
    }
    mats["w21"][5] = 0.0                                            # zero row
    mats["w21"][:, 7] = -3.0                                        # constant column
    holes = rng.normal(size=(200, 9)) * 4.0
    holes[rng.random(size=holes.shape) < 0.1] = np.nan
    holes[3, 2], holes[8, 2], holes[:, 6] = np.inf, -np.inf, np.nan
    out = dict(mats, holes=holes)
    for method in ("none", "row_l1", "row_l2", "col_zscore", "col_minmax"):
        for name, X in mats.items():
            out[f"{name}_{method}"] = normalize_feature(X, method)
        out[f"holes_{method}"] = normalize_feature(holes, method, impute="col_median")
    np.savez_compressed(os.path.join(HERE, "feat_norm_wide.npz"), **out)
    print("wide feature-normaliser fixture written")


def main():
    only = sys.argv[1:]            # optional fixture-name prefixes: regenerate only those
    global run_case
    if only:
        _run = run_case

        def run_case(name, **kw):                                  # noqa: F811
            if any(name.startswith(p) for p in only):
                _run(name, **kw)
    else:
        folds_case()
        feature_norm_case()
        feature_norm_wide_case()
    sim10 = dict(source="feature", feature_name="genres", metric="cosine",
                 topk=10, eps=1e-8)
    # g1: plain U/V (+ the always-on mu / bias terms)
    run_case("g1_plain", m=60, n=40, nnz=480, seed=101, k=4, n_iters=5,
             lambda_u=1.0, lambda_v=1.0)
    # g2: popularity-scaled lambda_v, explicit bias lambdas / the `or` fallback quirk
    run_case("g2_bias_pop", m=300, n=200, nnz=6000, seed=102, k=8, n_iters=6,
             lambda_u=2.0, lambda_v=3.0, pop="inverse_sqrt", bu=1.5, bi=2.5)
    run_case("g2_bias_zero", m=300, n=200, nnz=6000, seed=102, k=8, n_iters=6,
             lambda_u=2.0, lambda_v=3.0, pop="inverse_sqrt", bu=0.0, bi=0.0)
    # g3: empty users / items keep their random init and zero bias
    run_case("g3_empty", m=12, n=9, nnz=45, seed=103, k=3, n_iters=4,
             lambda_u=0.5, lambda_v=0.5, empty_users=(1, 3, 4, 6),
             empty_items=(0, 2, 4))
    # g4: feature projections, W schedule, Jacobi quirk, lambda_w-missing quirk
    for uw in (1, 2, 5):
        run_case(f"g4_feat_uw{uw}", m=300, n=200, nnz=6000, seed=104, k=8,
                 n_iters=7, lambda_u=2.0, lambda_v=3.0, pop="inverse_sqrt",
                 bu=1.5, bi=2.5, update_w_every=uw, feats=("genres", "years"),
                 lambda_w={"genres": 5.0})
    # g5: graph Laplacian (Gauss-Seidel order), S pinned as CSR
    for a in (0.5, 5.0):
        run_case(f"g5_graph_a{a}", m=300, n=200, nnz=6000, seed=105, k=8,
                 n_iters=6, lambda_u=2.0, lambda_v=3.0, pop="inverse_sqrt",
                 bu=1.5, bi=2.5, update_w_every=2, feats=("genres", "years"),
                 lambda_w={"genres": 5.0, "years": 10.0}, alpha=a, sim=sim10)
    # g6: early stopping at harness settings on a 610 x 4980-shaped problem
    run_case("g6_early_stop", m=610, n=4980, nnz=100000, seed=42, k=16,
             n_iters=60, lambda_u=5.0, lambda_v=5.0, bu=3.0, bi=2.0,
             tol=1e-4, min_iters=10, store_factors=False, sample_rows=64,
             store_inputs=False)
    # g7: factor-count sweep (padding paths of the kernels)
    for k in (1, 16, 50, 64, 128):
        run_case(f"g7_k{k}", m=60, n=40, nnz=480, seed=107, k=k, n_iters=4,
                 lambda_u=1.0, lambda_v=1.5, bu=0.7, bi=0.9)
    # g9: k=64 mid-size with long rows (exercises split-row tasks on the GPU)
    run_case("g9_k64_mid", m=400, n=50, nnz=12000, seed=109, k=64, n_iters=4,
             lambda_u=4.0, lambda_v=6.0, bu=3.0, bi=2.0,
             store_factors=False, sample_rows=50)
    # g10: the FULL model (popularity-scaled lambda_v + bias lambdas + genres/years projections + graph
    # Laplacian) at the wide ranks of BASELINE configs[3]/[4]: k = 64 and k = 128
    for k in (64, 128):
        run_case(f"g10_full_k{k}", m=300, n=200, nnz=6000, seed=110, k=k,
                 n_iters=5, lambda_u=2.0, lambda_v=3.0, pop="inverse_sqrt",
                 bu=1.5, bi=2.5, update_w_every=2, feats=("genres", "years"),
                 lambda_w={"genres": 5.0, "years": 10.0}, alpha=0.5, sim=sim10,
                 store_factors=False, sample_rows=64)
    # g11: the edges of the tuner's search space (scripts/tune_params.py:100-101 explores lambda in
    # [1e-4, 1e4]); most rows have fewer ratings than k, i.e. rank-deficient Grams with cond ~ 1/lambda
    run_case("g11_lam_tuned_k64", m=300, n=200, nnz=6000, seed=111, k=64, n_iters=6,
             lambda_u=1e3, lambda_v=1e-3, pop="inverse_sqrt", bu=3.0, bi=2.0,
             store_factors=False, sample_rows=64)
    run_case("g11_lam1e-2_k64", m=300, n=200, nnz=6000, seed=112, k=64, n_iters=6,
             lambda_u=1e-2, lambda_v=1e-2, bu=3.0, bi=2.0,
             store_factors=False, sample_rows=64)
    run_case("g11_lam1e-4_k64", m=300, n=200, nnz=6000, seed=113, k=64, n_iters=6,
             lambda_u=1e-4, lambda_v=1e-4, bu=3.0, bi=2.0,
             store_factors=False, sample_rows=64)
    # g12: feature projections WITHOUT lambda_w (missing -> lambda = 0, scripts/als.py:497; the harness's
    # `no_features`-style configurations, evaluate_models.py:413-417) on RANK-DEFICIENT designs - the reference's
    # float64 cholesky_solve goes through on the 1e-10 it adds (scripts/als.py:497-500, helpers.py:19-20); a W-step
    # assembled from fp32 item Grams reports "not positive definite" there:
    #   dup: two genre columns appear twice (exactly singular A_f, 128 null directions at k = 64);
    #   k80: k = 80 on 4000 ratings - the configuration DESIGN.md (round 2, section 5) documented as raising.
    # (A design with far fewer ratings than unknowns - 20 x 64 against 384 - was tried first and is not a usable
    # fixture: its W is ~10^3 and the fit is chaotic, fp32 storage of U / V alone moves W by 200 %.)
    run_case("g12_wlam0_dup_k64", m=300, n=200, nnz=6000, seed=122, k=64, n_iters=4,
             lambda_u=2.0, lambda_v=3.0, bu=0.7, bi=0.9, update_w_every=1,
             feats=("genres", "years"), lambda_w=None, dup_cols=2,
             store_factors=False, sample_rows=64)
    run_case("g12_wlam0_k80", m=300, n=200, nnz=4000, seed=124, k=80, n_iters=3,
             lambda_u=2.0, lambda_v=3.0, bu=0.7, bi=0.9, update_w_every=1,
             feats=("genres", "years"), lambda_w=None,
             store_factors=False, sample_rows=64)

if __name__ == "__main__":
    main()
