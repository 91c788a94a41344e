"""T1: the CPU oracle (oracle/als_oracle.py) against the golden fixtures made
from the real reference.  This is what pins the oracle ('Oracle')."""
import numpy as np
import pytest

from oracle.als_oracle import rmse_on_indices
from tests.common import Golden, golden_names

RTOL = 1e-9
ATOL = 1e-11


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference(name):
    g = Golden(name)
    o = g.run_oracle()
    d = g.d
    for key in ("train_rmse", "U_norm", "V_norm", "bu_norm", "bi_norm"):
        ref = d["hist_" + key]
        got = np.asarray(o.history[key])
        assert got.shape == ref.shape, f"{key}: iterations {got.shape} vs {ref.shape}"
        np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL, err_msg=key)
    if "sel_u" in d.files:
        U, V = o.U[d["sel_u"]], o.V[d["sel_i"]]
    else:
        U, V = o.U, o.V
    np.testing.assert_allclose(U, d["U"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(V, d["V"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(o.b_u, d["b_u"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(o.b_i, d["b_i"], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(o.mu, float(d["mu"][0]), rtol=RTOL)
    for f in g.cfg["feats"]:
        np.testing.assert_allclose(o.W[f], d["W_" + f], rtol=1e-6, atol=1e-9)
    pred = o.predict_at(g.val_flat(), g.features)
    np.testing.assert_allclose(pred, d["pred_val"], rtol=1e-7, atol=1e-9)
    assert abs(rmse_on_indices(g.val_truth(), pred) - float(d["test_rmse"][0])) < 1e-9


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("g5")])
def test_oracle_sparse_graph_path(name):
    """The CSR-S path of the oracle (used beyond dense-S sizes) against the
    same fixture; D is summed sparsely in float32, so tolerance is f32-level."""
    g = Golden(name)
    o = g.run_oracle(use_pinned_S=True)
    np.testing.assert_allclose(o.history["train_rmse"], g.d["hist_train_rmse"], rtol=1e-6)
    np.testing.assert_allclose(o.V, g.d["V"], rtol=1e-4, atol=1e-6)


def test_empty_rows_keep_init():
    g = Golden("g3_empty")
    o = g.run_oracle()
    rng = np.random.default_rng(42)
    U0 = rng.normal(scale=0.1, size=(g.m, 3))
    V0 = rng.normal(scale=0.1, size=(g.n, 3))
    for u in g.cfg["empty_users"]:
        np.testing.assert_array_equal(o.U[u], U0[u])
        assert o.b_u[u] == 0.0
    for i in g.cfg["empty_items"]:
        np.testing.assert_array_equal(o.V[i], V0[i])
        assert o.b_i[i] == 0.0
