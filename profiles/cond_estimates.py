#!/usr/bin/env python3
"""Distribution of the per-row condition estimate that solve_dtype="auto" decides on (als_row_solve_params::cond_limit,
row_solve.hip::row_needs_f64): kappa = max( (max L_ii / min L_ii)^2,  (trace(G) / rank + lambda) / min L_ii^2,
mean eigenvalue / lambda for rows of fewer than 4 k ratings ).  The headline workload (cfg4, rows of both half-steps
after a few iterations, by row length) and the small-lambda / ill-posed fixtures.  The limit (backend.COND_LIMIT = 300)
must sit above the former (cost of auto at the BASELINE shapes = 0) and below the rows whose fp32 solve leaves the budget.
    python profiles/cond_estimates.py [size] > profiles/r03_cond_estimates.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                                  # noqa: E402
from tests.common import Golden                                                               # noqa: E402
from tests.test_gpu_parity import _model_for                                                  # noqa: E402

os.environ["ALS_COND_LIMIT"] = "1e30"          # estimate every row, redo none


def quantiles(x):
    x = x[np.isfinite(x) & (x > 0)]
    q = np.quantile(x, [0.5, 0.9, 0.99, 0.999, 1.0]) if x.size else [np.nan] * 5
    return "rows %8d  median %.3g  p90 %.3g  p99 %.3g  p99.9 %.3g  max %.3g" % ((x.size,) + tuple(q))


def probe_engine(eng, iters):
    be = eng.be
    for it in range(iters):
        be.cond_probe = torch.zeros(eng.m, dtype=torch.float32, device=eng.dev)
        eng.user_step()
        ku = be.cond_probe.cpu().numpy()
        be.cond_probe = torch.zeros(eng.n, dtype=torch.float32, device=eng.dev)
        eng.b_i_prev = eng.b_i.clone()
        eng.item_step(want_gram=bool(eng.feat_names))
        kv = be.cond_probe.cpu().numpy()
        be.cond_probe = None
        if eng.feat_names:
            eng.w_step(eng.b_i_prev)
        eng.stats_step(None)
    return ku, kv


def main():
    dev = torch.device("cuda", 0)
    size = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    inp = bench.make_inputs(size, dev, 0, False)
    res = bench.run_case(size, inp, dev, steps=1, warmup=0)
    eng = res["eng"]
    ku, kv = probe_engine(eng, 3)
    lens_u = (eng.csr.indptr[1:] - eng.csr.indptr[:-1]).cpu().numpy()
    lens_i = (eng.csc.indptr[1:] - eng.csc.indptr[:-1]).cpu().numpy()
    print(f"{size} (lambda_u 5, lambda_v 6, alpha 0.5), after 4 iterations:")
    print("  U-step", quantiles(ku))
    print("  V-step", quantiles(kv))
    for lo, hi in ((1, 63), (64, 255), (256, 4095), (4096, 10 ** 9)):
        for name, k_, l_ in (("U", ku, lens_u), ("V", kv, lens_i)):
            sel = (l_ >= lo) & (l_ <= hi)
            if sel.any():
                print(f"    {name} rows of {lo}-{hi} ratings:", quantiles(k_[sel]))
    del eng, res, inp
    torch.cuda.empty_cache()
    for name in ("g2_bias_pop", "g9_k64_mid", "g10_full_k64", "g11_lam_tuned_k64", "g11_lam1e-2_k64", "g11_lam1e-4_k64",
                 "g12_wlam0_dup_k64"):
        g = Golden(name)
        r, c, v = g.train
        model = _model_for(g, device=dev)
        model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=None, verbose=0)
        e = model._eng
        ku, kv = probe_engine(e, 1)
        print(f"{name} (lambda_u {g.cfg['lambda_u']}, lambda_v {g.cfg['lambda_v']}, pop {g.cfg['pop_reg_mode']}):")
        print("  U-step", quantiles(ku))
        print("  V-step", quantiles(kv))


if __name__ == "__main__":
    main()
