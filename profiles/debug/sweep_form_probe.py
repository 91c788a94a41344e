#!/usr/bin/env python3
"""Final train RMSE of a small graph fit at several ranks - run once per sweep form (ALS_GS_FORM=image / stream) and
compare: the two forms share dependency semantics and summation order and must agree to the last bit."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig, layout   # noqa: E402
from tests.synth import make_features, make_ratings                                                                      # noqa: E402

for k in [int(x) for x in sys.argv[1:]] or [100, 112, 128, 130, 144, 150, 160]:
    m, n = 400, 300
    r, c, v = make_ratings(m, n, 9000, seed=500 + k)
    G, _ = make_features(n, 9)
    S_csr = layout.dense_graph_to_csr(layout.build_similarity_dense(G, 8, 1e-8))
    cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=4, lambda_u=3.0, lambda_v=4.0, pop_reg_mode="inverse_sqrt"),
                    biases=BiasesConfig(lambda_bu=2.0, lambda_bi=1.5),
                    graph=GraphConfig(alpha=0.8, sim=GraphSimConfig(source="precomputed")))
    model = ALS(cfg)
    if os.environ.get("ALS_PROBE_LEVELS") == "1":
        model._dataflow_sweep = False                 # per-level launches (k_gs_level: the same substitutions)
    model.fit_coo(r, c, v, (m, n), tol=None, verbose=0, S=S_csr)
    print(k, os.environ.get("ALS_GS_FORM", "auto"), repr(model.history["train_rmse"][-1]), float(np.abs(model.V).sum()), flush=True)
