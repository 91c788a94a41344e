"""MI355X-native ALS matrix-factorisation solver (drop-in for the `ALS` class of
zhukovanadezhda/collaborative-filtering, scripts/als.py).

    from collaborative_filtering_amd import ALS, ALSConfig, CoreConfig, ...

The directory is named `collaborative-filtering_amd`; the importable alias
`collaborative_filtering_amd` (repo root) points here.
"""
from .als_config import ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig
from .als import ALS
from .helpers import (ES_MIN_ITERS, ES_TOL, DEFAULT_RANDOM_STATE, cholesky_solve, make_config,
                      normalize_params, rmse_on_indices)

from . import cv                     # sparse CV / ablation harness
from . import features               # feature normaliser
from . import sweep                  # resident-data hyper-parameter sweep driver

__all__ = ["ALS", "cv", "features", "sweep", "ALSConfig", "BiasesConfig", "CoreConfig", "GraphConfig", "GraphSimConfig",
           "cholesky_solve", "make_config", "normalize_params", "rmse_on_indices",
           "ES_TOL", "ES_MIN_ITERS", "DEFAULT_RANDOM_STATE"]
