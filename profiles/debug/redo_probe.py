#!/usr/bin/env python3
"""How many rows does solve_dtype="auto" redo in fp64 per half-step as a fit goes on?  (bench workload, per iteration:
rows redone in the U-step / V-step, time of the iteration, and the quantiles of the condition estimate at the end.)
    python profiles/debug/redo_probe.py [size] [iterations]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench                                                                                  # noqa: E402
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig   # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "cfg5-small"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
inputs = bench.make_inputs(size, dev, 0, False, "sampled", False)
m, n, _, k = bench.SIZES[size]
csr, csc, S, features = inputs[:4]
cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=iters, lambda_u=5.0, lambda_v=6.0, random_state=42,
                                pop_reg_mode="inverse_sqrt" if size in ("cfg5-small", "cfg5") else None),
                biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0),
                graph=(GraphConfig(alpha=0.5, sim=GraphSimConfig(source="precomputed", topk=50)) if S is not None else GraphConfig()))
model = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0} if features else None, device=dev)
eng = model.prepare_csr(csr, csc, (m, n), features=features, S=S)
if features:
    eng.be.compose_z(eng.V, eng.Xcat, eng.Wcat, eng.Z)
be = eng.be
for it in range(iters):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.user_step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ru = int(be._redo_count.item())
    do_w = bool(eng.feat_names) and ((it % model.update_w_every == 0) or (it == iters - 1))
    b_i_old = eng.b_i.clone() if (do_w or eng.fused_feat_stats) else None
    eng.b_i_prev = b_i_old
    eng.item_step(want_gram=do_w or eng.fused_feat_stats)
    torch.cuda.synchronize()
    rv = int(be._redo_count.item())
    if do_w:
        eng.w_step(b_i_old)
    eng.stats_step(it)
    eng.iters_run = it + 1
    torch.cuda.synchronize()
    if it < 6 or it % 5 == 4:
        print(f"iteration {it + 1:3d}: U-step {1e3 * (t1 - t0):7.2f} ms, rows redone in fp64: U-step {ru:7d} of {m}, V-step {rv:6d} of {n};"
              f"  max|Z| {float(eng.Z.abs().max()):.3g}  max|U| {float(eng.U.abs().max()):.3g}", flush=True)
be.cond_probe = torch.zeros(m, dtype=torch.float32, device=dev)
eng.user_step()
torch.cuda.synchronize()
c = be.cond_probe.cpu().numpy()
c = c[np.isfinite(c) & (c > 0)]
print("U-step condition estimates after", iters, "iterations:", {q: float(np.quantile(c, q)) for q in (0.5, 0.9, 0.99, 0.999, 1.0)})
