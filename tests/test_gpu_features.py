"""SURVEY 8(f) n4 on the device: `als_normalize_features` (csrc/features.hip) against outputs of the reference's
scripts/prepare_features.py stored in tests/golden/feat_norm_50x6.npz and feat_norm_wide.npz - bitwise, because
the kernels sum in numpy's order (pairwise along the contiguous axis, row after row along the other)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
METHODS = ("none", "row_l1", "row_l2", "col_zscore", "col_minmax")


def _dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no ROCm device is visible")
    return "cuda:0"


@pytest.mark.parametrize("method", METHODS)
def test_device_normaliser_is_bitwise_the_reference(method):
    from collaborative_filtering_amd import features as F
    dev = _dev()
    small = np.load(os.path.join(HERE, "golden", "feat_norm_50x6.npz"))
    wide = np.load(os.path.join(HERE, "golden", "feat_norm_wide.npz"))
    a = F.normalize_feature_device(small["clean"], method, device=dev)
    assert a.dtype.is_floating_point and a.element_size() == 4
    assert np.array_equal(a.cpu().numpy(), small[f"clean_{method}"])
    b = F.normalize_feature_device(small["X"], method, impute="col_median", device=dev)
    assert np.array_equal(b.cpu().numpy(), small[f"imputed_{method}"], equal_nan=True)
    for name in ("w21", "w150", "w1", "w2"):
        got = F.normalize_feature_device(wide[name], method, device=dev).cpu().numpy()
        want = wide[f"{name}_{method}"]
        assert got.shape == want.shape
        assert np.array_equal(got, want), (name, method, float(np.abs(got - want).max()))
    h = F.normalize_feature_device(wide["holes"], method, impute="col_median", device=dev).cpu().numpy()
    assert np.array_equal(h, wide[f"holes_{method}"])


def test_device_normaliser_dict_and_errors():
    from collaborative_filtering_amd import features as F
    dev = _dev()
    g = np.load(os.path.join(HERE, "golden", "feat_norm_50x6.npz"))
    d = F.normalize_features_dict({"genres": g["clean"], "years": g["years"]}, method="none", impute="col_median",
                                  per_feature_overrides={"genres": {"method": "row_l2"},
                                                         "years": {"method": "col_zscore"}}, device=dev)
    assert np.array_equal(d["genres"].cpu().numpy(), g["dict_genres"])
    assert np.array_equal(d["years"].cpu().numpy(), g["dict_years"]) and tuple(d["years"].shape) == (50, 1)
    with pytest.raises(ValueError, match="Unknown method"):
        F.normalize_feature_device(g["clean"], "l2", device=dev)
    with pytest.raises(ValueError, match="Unknown impute"):
        F.normalize_feature_device(g["clean"], "none", impute="mean", device=dev)
    with pytest.raises(ValueError, match="NaN/Inf"):
        F.normalize_feature_device(g["X"], "row_l2", device=dev)
    with pytest.raises(TypeError):
        F.normalize_feature_device(g["clean"].astype(np.float32), "row_l2", device=dev)


def test_device_normaliser_at_catalogue_size():
    """10^6 items x 20 columns (the BASELINE configs[4] item count): equal to the host form (numpy) bit for bit."""
    from collaborative_filtering_amd import features as F
    dev = _dev()
    rng = np.random.default_rng(9)
    X = (rng.random(size=(1_000_000, 20)) < 0.15).astype(np.float64) + rng.normal(size=(1_000_000, 20)) * 1e-3
    for method in ("row_l2", "col_zscore"):
        got = F.normalize_feature_device(X, method, device=dev).cpu().numpy()
        assert np.array_equal(got, F.normalize_feature(X, method))


def test_fit_takes_device_normalised_features():
    """The normaliser's device tensors go straight into ALS.fit: same fit as with the host-normalised arrays
    (fixture g4: genres + years projections), bit for bit."""
    from collaborative_filtering_amd import features as F
    from tests.common import Golden
    from tests.test_gpu_parity import _model_for
    dev = _dev()
    g = Golden("g4_feat_uw2")
    raw = {name: np.asarray(X, dtype=np.float64) for name, X in g.features.items()}
    over = {"genres": {"method": "row_l2"}, "years": {"method": "col_zscore"}}
    host = F.normalize_features_dict(raw, per_feature_overrides=over)
    devf = F.normalize_features_dict(raw, per_feature_overrides=over, device=dev)
    assert all(np.array_equal(devf[n].cpu().numpy(), host[n]) for n in raw)
    r, c, v = g.train
    a = _model_for(g).fit_coo(r, c, v, (g.m, g.n), features=host, tol=None, verbose=0)
    b = _model_for(g).fit_coo(r, c, v, (g.m, g.n), features=devf, tol=None, verbose=0)
    assert np.array_equal(a.U, b.U) and np.array_equal(a.V, b.V)
    assert all(np.array_equal(a.W[n], b.W[n]) for n in a.W)
    assert np.array_equal(a.predict_at(g.val_flat(), host), b.predict_at(g.val_flat(), devf))


def test_device_median_imputation_against_numpy_at_size():
    """als_impute_col_median (radix select): 200 001 x 7 with NaN / +-inf holes, an all-missing column, columns with an
    even and an odd number of finite entries, duplicates and signed zeros - equal to numpy's nanmedian fill."""
    import torch
    from collaborative_filtering_amd import features as F
    dev = _dev()
    rng = np.random.default_rng(12)
    n, d = 200_001, 7
    X = np.round(rng.normal(size=(n, d)) * 50.0) / 8.0            # many duplicates
    X[rng.random(size=(n, d)) < 0.2] = np.nan
    X[::97, 1] = np.inf
    X[5::89, 1] = -np.inf
    X[:, 3] = np.nan                                             # all missing -> 0
    X[:1000, 4] = -0.0
    X[np.isnan(X[:, 5]), 5] = 1.0
    X[0, 5] = np.nan                                             # exactly one hole: n - 1 = even count
    want = F.normalize_feature(X, "none", impute="col_median", dtype="float64")
    got = F.normalize_feature_device(X, "none", impute="col_median", device=dev)
    assert np.array_equal(got.cpu().numpy(), want.astype(np.float32))
    assert np.isnan(X).any()                                     # the caller's array is left alone
    Xt = torch.from_numpy(X).to(dev)
    F.normalize_feature_device(Xt, "col_zscore", impute="col_median", device=dev)
    assert bool(torch.isnan(Xt).any())                           # ... also when it already lives on the device
