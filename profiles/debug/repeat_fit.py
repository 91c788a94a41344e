#!/usr/bin/env python3
"""Debug probe: repeat one fixture fit N times (no synchronisation between calls), report completion."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.common import Golden
from tests.test_gpu_parity import _model_for
name, n = sys.argv[1], int(sys.argv[2])
kw = {"solve_dtype": sys.argv[3]} if len(sys.argv) > 3 else {}
g = Golden(name)
r, c, v = g.train
for i in range(n):
    model = _model_for(g, device="cuda:0", **kw)
    model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"], min_iters=g.cfg["min_iters"], verbose=0)
    torch.cuda.synchronize()
    print("fit", i, "ok", model.history["train_rmse"][-1], "redo count", int(model._eng.be._redo_count.item()), flush=True)
