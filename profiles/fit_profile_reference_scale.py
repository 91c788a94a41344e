#!/usr/bin/env python3
"""Where a fit + predict goes at the reference's own scale (610 x 4980, 100K ratings, k = 64): the named
caller times `ALS(...).fit(...)` + `predict` together (scripts/evaluate_models.py:245-255), so host-side set-up
counts.  Two cases: features only, and the full model with the genre graph (top-k 50, built once and memoised).
Prints wall-clock per stage for three repetitions each and a cProfile top-15 of the engine construction."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig, _hip, layout  # noqa: E402
from tests.synth import make_features, make_ratings  # noqa: E402

m, n, nnz = 610, 4980, 100000
rows, cols, vals = make_ratings(m, n, nnz, 5)
G, Y = make_features(n, 6)
feats = {"genres": G, "years": Y}
lib = _hip.load()


def cfg_for(graph):
    return ALSConfig(core=CoreConfig(n_factors=64, n_iters=20, lambda_u=5.0, lambda_v=6.0, random_state=42),
                     biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0),
                     graph=GraphConfig(alpha=0.5, sim=GraphSimConfig(feature_name="genres", topk=50)) if graph else GraphConfig())


for graph in (False, True):
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        csr, csc = layout.coo_to_sides_native(lib, rows, cols, vals, (m, n))
        t1 = time.perf_counter()
        model = ALS(cfg_for(graph), lambda_w={"genres": 5.0, "years": 10.0}, device="cuda:0")
        prof = cProfile.Profile() if rep == 3 else None
        if prof:
            prof.enable()
        model._fit_sides(csr, csc, feats, None, 0, 0, None, run=False)
        if prof:
            prof.disable()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        model._eng.run(None, 0, 0)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        model._eng.export(model)
        t4 = time.perf_counter()
        idx = np.arange(0, m * n, 37)
        p = model.predict_at(idx, features=feats)
        t5 = time.perf_counter()
        print(f"graph={graph} rep {rep}: coo_to_sides {1e3 * (t1 - t0):.2f} ms, engine init {1e3 * (t2 - t1):.2f}, "
              f"20 iterations {1e3 * (t3 - t2):.2f}, export {1e3 * (t4 - t3):.2f}, predict_at({idx.size}) {1e3 * (t5 - t4):.2f}, "
              f"total {1e3 * (t5 - t0):.2f}", flush=True)
        if prof:
            s = io.StringIO()
            pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(18)
            print(s.getvalue()[:3500])
