#!/usr/bin/env python3
"""Observed errors of one fixture fit against the reference (what the test tolerances are set from)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tests.common import Golden
from tests.test_gpu_parity import _model_for
for name in sys.argv[1].split(","):
    for kw in ({}, {"gram": "f32"}, {"solve_dtype": "float64"}):
        g = Golden(name); d = g.d
        r, c, v = g.train
        m = _model_for(g, **kw).fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"], min_iters=g.cfg["min_iters"], verbose=0)
        U, V = (m.U[d["sel_u"]], m.V[d["sel_i"]]) if "sel_u" in d.files else (m.U, m.V)
        rel = lambda a, b: float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
        pred = m.predict_at(g.val_flat(), g.features or None)
        rm = float(np.sqrt(np.mean((g.val_truth() - pred) ** 2)))
        print(name, kw, "hist", float(np.max(np.abs(np.array(m.history["train_rmse"]) - d["hist_train_rmse"]))), "U", rel(U, d["U"]), "V", rel(V, d["V"]),
              "W", {f: rel(m.W[f], d["W_" + f]) for f in g.cfg["feats"]}, "pred", float(np.max(np.abs(pred - d["pred_val"]))),
              "test_rmse", abs(rm - float(d["test_rmse"][0])), "redone rows (last call)", int(m._eng.be._redo_count.item()), flush=True)
