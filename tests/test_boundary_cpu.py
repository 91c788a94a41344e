"""T2: the drop-in boundary - dataclass fields and defaults, constructor / fit /
predict signatures, error types, history keys, the `or` fallback of the bias lambdas.  Everything
here fails before any device is touched, so it runs on CPU."""
import dataclasses
import inspect

import numpy as np
import pytest

from collaborative_filtering_amd import (ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig,
                                         cholesky_solve, make_config, normalize_params, rmse_on_indices)
from collaborative_filtering_amd import helpers


def _fields(cls):
    return [(f.name, f.default if f.default is not dataclasses.MISSING else "<required>") for f in dataclasses.fields(cls)
            if f.default_factory is dataclasses.MISSING] + \
           [(f.name, "<factory>") for f in dataclasses.fields(cls) if f.default_factory is not dataclasses.MISSING]


def test_config_dataclasses_match_reference_fields_and_defaults():
    # scripts/als_config.py:57-95
    assert _fields(CoreConfig) == [("n_factors", "<required>"), ("n_iters", "<required>"), ("lambda_u", "<required>"),
                                   ("lambda_v", "<required>"), ("pop_reg_mode", None), ("random_state", 42),
                                   ("update_w_every", 5)]
    assert _fields(BiasesConfig) == [("lambda_bu", None), ("lambda_bi", None)]
    assert _fields(GraphSimConfig) == [("source", "feature"), ("feature_name", "genres"), ("metric", "cosine"),
                                       ("topk", 50), ("eps", 1e-8)]
    assert _fields(GraphConfig) == [("alpha", 0.0), ("sim", None)]
    assert sorted(_fields(ALSConfig)) == sorted([("core", "<required>"), ("biases", "<factory>"), ("graph", "<factory>")])
    cfg = ALSConfig(core=CoreConfig(4, 2, 1.0, 1.0))
    assert cfg.biases == BiasesConfig() and cfg.graph == GraphConfig()
    # no validation at construction, as in the reference
    CoreConfig(n_factors=-3, n_iters=0, lambda_u=-1.0, lambda_v=0.0, pop_reg_mode="bogus")


def test_signatures_match_reference():
    sig = inspect.signature(ALS.__init__)
    names = list(sig.parameters)
    assert names[:3] == ["self", "config", "lambda_w"] and sig.parameters["lambda_w"].default is None
    assert all(sig.parameters[n].kind is inspect.Parameter.KEYWORD_ONLY for n in names[3:])   # build-only extras
    fit = inspect.signature(ALS.fit)
    assert list(fit.parameters)[:6] == ["self", "R", "features", "tol", "min_iters", "verbose"]
    assert (fit.parameters["features"].default, fit.parameters["tol"].default, fit.parameters["min_iters"].default,
            fit.parameters["verbose"].default) == (None, 1e-3, 5, 1)                           # scripts/als.py:300-306
    pred = inspect.signature(ALS.predict)
    assert list(pred.parameters) == ["self", "features"] and pred.parameters["features"].default is None


def test_constructor_state_and_quirks():
    with pytest.raises(ValueError):
        ALS(None)                                                                              # scripts/als.py:146-147
    m = ALS(ALSConfig(core=CoreConfig(8, 3, 2.0, 3.0), biases=BiasesConfig(0.0, None)), lambda_w={"genres": 5.0})
    assert m.lambda_bu == 2.0 and m.lambda_bi == 3.0            # `x or y`: 0.0 and None both fall back (:166-167)
    assert ALS(ALSConfig(core=CoreConfig(8, 3, 2.0, 3.0), biases=BiasesConfig(1.5, 2.5))).lambda_bi == 2.5
    assert m.S_topk is None and m.S_eps == 1e-10                # no sim config (:171-174)
    g = ALS(ALSConfig(core=CoreConfig(8, 3, 2.0, 3.0), graph=GraphConfig(0.5, GraphSimConfig(topk=7, eps=1e-5))))
    assert (g.alpha, g.S_topk, g.S_eps) == (0.5, 7, 1e-5)
    assert list(m.history) == ["train_rmse", "U_norm", "V_norm", "bu_norm", "bi_norm"] and all(v == [] for v in m.history.values())
    assert m.U is None and m.V is None and m.b_u is None and m.b_i is None and m.mu == 0.0 and m.S is None and m.W == {}
    lw = {"genres": 5.0}
    m2 = ALS(ALSConfig(core=CoreConfig(8, 3, 2.0, 3.0)), lambda_w=lw)
    lw["genres"] = 9.0
    assert m2.lambda_w == {"genres": 5.0}                       # copied (:153)


def test_error_behaviour_before_any_device_work():
    cfg = ALSConfig(core=CoreConfig(4, 2, 1.0, 1.0))
    with pytest.raises(RuntimeError):
        ALS(cfg).predict()                                                                     # :554-555
    R = np.full((5, 4), np.nan)
    R[0, 1] = 3.0
    with pytest.raises(ValueError, match="rows"):
        ALS(cfg).fit(R, features={"genres": np.zeros((3, 2))}, verbose=0)                     # :347-349
    bad = np.zeros((4, 2))
    bad[1, 1] = np.nan
    with pytest.raises(ValueError, match="infinite"):
        ALS(cfg).fit(R, features={"genres": bad}, verbose=0)                                   # :350-351
    import torch
    with pytest.raises(ValueError, match="infinite"):                  # tensors (the device normaliser's output type)
        ALS(cfg).fit(R, features={"genres": torch.from_numpy(bad)}, verbose=0)     # are validated like arrays
    with pytest.raises(ValueError, match="pop_reg_mode"):
        ALS(ALSConfig(core=CoreConfig(4, 2, 1.0, 1.0, pop_reg_mode="linear"))).fit(R, verbose=0)   # :259
    with pytest.raises(ValueError):
        ALS(cfg).fit_coo([0, 0], [1, 1], [3.0, 4.0], (5, 4), verbose=0)                        # duplicate entries
    with pytest.raises(ValueError):
        ALS(cfg).fit_coo([7], [1], [3.0], (5, 4), verbose=0)                                   # index outside the shape


def test_harness_glue():
    # scripts/tune_params.py:147-167
    Rt = np.arange(12, dtype=float).reshape(3, 4)
    Rp = Rt + 0.5
    assert rmse_on_indices(Rt, Rp, np.array([0, 5, 11])) == pytest.approx(0.5)
    assert np.isnan(rmse_on_indices(Rt, Rp, np.array([], dtype=np.int64)))
    # scripts/tune_params.py:237-278: four clamps
    p = normalize_params({"n_factors": 500, "S_topk": 999, "update_w_every": 70, "n_iters": 20, "alpha": 0.7,
                          "graph_feature": "tags"}, (30, 12), ["genres"])
    assert (p["n_factors"], p["S_topk"], p["update_w_every"], p["alpha"], p["graph_feature"]) == (12, 11, 20, 0.0, "__none__")
    p = normalize_params({"n_factors": 0, "S_topk": 0, "update_w_every": 0, "n_iters": 5, "alpha": 0.7,
                          "graph_feature": "genres"}, (30, 12), ["genres"])
    assert (p["n_factors"], p["S_topk"], p["update_w_every"], p["alpha"]) == (1, 1, 1, 0.7)
    # scripts/tune_params.py:281-322
    c = make_config({"n_factors": 6, "n_iters": 9, "lambda_u": 1.0, "lambda_v": 2.0, "alpha": 0.3, "graph_feature": "genres",
                     "S_topk": 5, "S_eps": 1e-6, "pop_reg_mode": "inverse_sqrt", "update_w_every": 3})
    assert c.core == CoreConfig(6, 9, 1.0, 2.0, "inverse_sqrt", 42, 3)
    assert c.biases == BiasesConfig(1.0, 2.0) and c.graph == GraphConfig(0.3, GraphSimConfig("feature", "genres", "cosine", 5, 1e-6))
    assert make_config({"n_factors": 6, "n_iters": 9, "lambda_u": 1.0, "lambda_v": 2.0, "alpha": 0.0,
                        "graph_feature": "genres"}).graph == GraphConfig(0.0, None)
    assert make_config({"n_factors": 6, "n_iters": 9, "lambda_u": 1.0, "lambda_v": 2.0, "alpha": 0.4}).graph.sim is None
    assert (helpers.ES_TOL, helpers.ES_MIN_ITERS, helpers.DEFAULT_RANDOM_STATE) == (1e-4, 10, 42)


def test_cholesky_solve_host_utility():
    rng = np.random.default_rng(0)
    B = rng.normal(size=(6, 6))
    A = B @ B.T + 6 * np.eye(6)
    b = rng.normal(size=6)
    np.testing.assert_allclose(cholesky_solve(A, b), np.linalg.solve(A, b), rtol=1e-12)
    with pytest.raises(np.linalg.LinAlgError):
        cholesky_solve(-np.eye(3), np.ones(3))                                                 # scripts/helpers.py:19


def test_caller_supplied_graph_is_validated():
    """fit(..., S=csr): the sweep's level schedule relies on S being symmetric (a neighbour j > i must be on a
    later level); an asymmetric, unsorted or out-of-range graph is refused up front with ValueError."""
    import torch
    from collaborative_filtering_amd.als import _check_csr, _validate_graph
    t = torch.tensor
    ptr, idx, val = t([0, 1, 3, 4]), t([1, 0, 2, 1], dtype=torch.int32), t([.5, .5, .25, .25])
    _validate_graph(ptr, idx, val, 3)                                        # symmetric: accepted
    with pytest.raises(ValueError, match="symmetric"):
        _validate_graph(ptr, idx, t([.5, .5, .25, .3]), 3)                   # values differ across the diagonal
    with pytest.raises(ValueError, match="symmetric"):
        _validate_graph(t([0, 1, 2, 2]), t([1, 2], dtype=torch.int32), t([.5, .5]), 3)   # pattern not symmetric
    with pytest.raises(ValueError, match="ascending"):
        _validate_graph(t([0, 0, 2, 2]), t([2, 0], dtype=torch.int32), t([.5, .5]), 3)
    with pytest.raises(ValueError, match="diagonal"):
        _validate_graph(t([0, 1, 1, 1]), t([0], dtype=torch.int32), t([.5]), 3)
    with pytest.raises(ValueError, match="outside"):
        _validate_graph(t([0, 1, 1, 1]), t([7], dtype=torch.int32), t([.5]), 3)
    with pytest.raises(ValueError, match="indptr"):
        _check_csr(t([0, 2, 1, 3]), t([0, 1, 2], dtype=torch.int32), 3, 3, "x")


def test_device_csr_entry_checks_index_range():
    import torch
    from collaborative_filtering_amd.als import _as_side
    ptr = torch.tensor([0, 2, 3])
    vals = torch.ones(3)
    _as_side((ptr, torch.tensor([0, 1, 1], dtype=torch.int32), vals), 2, 2)
    with pytest.raises(ValueError, match="outside"):
        _as_side((ptr, torch.tensor([0, 2, 1], dtype=torch.int32), vals), 2, 2)
    with pytest.raises(ValueError, match="outside"):
        _as_side((ptr.numpy(), np.array([0, -1, 1], dtype=np.int32), vals.numpy()), 2, 2)
