#!/usr/bin/env python3
"""Debug probe: one fit of a fixture with lambda_w missing (fp64 V-step by-products), every backend call synchronised."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.common import Golden
from tests.test_gpu_parity import _model_for, _check_against_fixture, TOL
from collaborative_filtering_amd.backend import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "g4_feat_uw1"
for meth in ("row_solve", "w_item_vectors", "w_accumulate", "spd_solve", "item_stats", "gs_level", "gs_levels", "gs_dataflow",
             "sum_pairs", "history_row", "compose_z"):
    orig = getattr(HipBackend, meth)
    def wrap(self, *a, _o=orig, _m=meth, **kw):
        print("->", _m, {k: (str(v.dtype) + str(tuple(v.shape)) if torch.is_tensor(v) else v) for k, v in kw.items()
                         if k in ("f64", "gram", "gram_out", "H", "W", "rhs", "factor", "factor_out")}, flush=True)
        r = _o(self, *a, **kw)
        torch.cuda.synchronize()
        print("<-", _m, flush=True)
        return r
    setattr(HipBackend, meth, wrap)
g = Golden(name)
r, c, v = g.train
model = _model_for(g, device="cuda:0", **({"solve_dtype": sys.argv[2]} if len(sys.argv) > 2 else {}))
model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"], min_iters=g.cfg["min_iters"], verbose=0)
print("v_f64", model._eng.v_f64)
_check_against_fixture(model, g, TOL)
print("fixture ok")
