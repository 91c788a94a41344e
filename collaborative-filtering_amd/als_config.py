"""Configuration dataclasses of the ALS solver.

Drop-in for the reference's `scripts/als_config.py:57-95`: same class names,
same field names, same defaults, same nesting, no validation at construction
(the reference validates nothing here; `pop_reg_mode` is checked in `fit`,
`config is None` in `ALS.__init__`).  Only the documentation differs.
"""
from dataclasses import dataclass, field
from typing import Literal, Optional


@dataclass
class CoreConfig:
    n_factors: int                     # k, latent dimension
    n_iters: int                       # outer ALS iterations (upper bound)
    lambda_u: float                    # ridge on user factors
    lambda_v: float                    # ridge on item factors (base value)
    pop_reg_mode: Optional[Literal["inverse_sqrt"]] = None   # lambda_v / sqrt(count+1)
    random_state: int = 42             # seed of numpy's default_rng for the init
    update_w_every: int = 5            # W-step period (plus the last iteration)


@dataclass
class BiasesConfig:
    lambda_bu: Optional[float] = None  # None or 0.0 -> lambda_u  (reference `or` rule)
    lambda_bi: Optional[float] = None  # None or 0.0 -> lambda_v


@dataclass
class GraphSimConfig:
    source: Literal["feature", "precomputed"] = "feature"
    feature_name: str = "genres"
    metric: Literal["cosine"] = "cosine"
    topk: Optional[int] = 50
    eps: float = 1e-8


@dataclass
class GraphConfig:
    alpha: float = 0.0                         # Laplacian strength; 0 disables
    sim: Optional[GraphSimConfig] = None       # None disables


@dataclass
class ALSConfig:
    core: CoreConfig
    biases: BiasesConfig = field(default_factory=BiasesConfig)
    graph: GraphConfig = field(default_factory=GraphConfig)
