#!/usr/bin/env python3
"""Size-independent optimality checks of one U-step and one V-step + Gauss-Seidel sweep on a bench.py workload
that is too large for the test suite (default: BASELINE configs[4] at full size, `--size cfg5`, ~170 GB of HBM).
Same properties as tests/test_gpu_fullsize.py: for sampled rows the stored factor row solves the row's normal
equations built in float64 from the state it saw (relative residual of an fp32 solve), the bias equals its closed
form, and the Laplacian term uses the live values of the sweep order.  Prints one JSON line.
Usage (GPU box): python3 profiles/check_full_size_properties.py [--size cfg5]"""
import argparse, json, os, sys, time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig  # noqa: E402

EPS = 1e-10


def rel_residual(A, x, b):
    return float(np.linalg.norm(A @ x - b) / (np.linalg.norm(A) * np.linalg.norm(x) + np.linalg.norm(b)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="cfg5")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    m, n, nnz, k = bench.SIZES[args.size]
    t0 = time.perf_counter()
    features = bench.gen_features(n, 3004) if args.size in ("cfg3", "cfg5-small", "cfg5") else None
    csr, csc = bench.gen_ratings(dev, m, n, nnz, seed=1004)
    S = bench.gen_graph(dev, n, seed=2004) if args.size not in ("cfg2", "cfg3") else None
    cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=3, lambda_u=5.0, lambda_v=6.0, random_state=42,
                                    pop_reg_mode="inverse_sqrt" if args.size in ("cfg5-small", "cfg5") else None),
                    biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0),
                    graph=(GraphConfig(alpha=0.5, sim=GraphSimConfig(source="precomputed", topk=50))
                           if S is not None else GraphConfig()))
    model = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0} if features else None, device=dev)
    eng = model.prepare_csr(csr, csc, (m, n), features=features, S=S)
    if features:
        eng.be.compose_z(eng.V, eng.Xcat, eng.Wcat, eng.Z)
    eng.iteration(0, 3)                                   # one full iteration: non-trivial state
    torch.cuda.synchronize()
    eng._check_status()
    setup_s = time.perf_counter() - t0
    rng = np.random.default_rng(7)
    out = {"size": args.size, "m": m, "n": n, "nnz": int(eng.nnz), "k": k, "setup_plus_one_iteration_s": setup_s}

    # ---- U-step ---------------------------------------------------------------------------------
    b_u_old = eng.b_u.clone()
    eng.user_step()
    torch.cuda.synchronize()
    eng._check_status()
    uptr = eng.csr.indptr.cpu().numpy()
    lens = np.diff(uptr)
    picks = [int(lens.argmax()), int(lens.argmin())]
    for lo_len, hi_len in ((1, 16), (17, 64), (65, 96), (97, 4096), (4097, 1 << 40)):   # every K1 row class
        cand = np.nonzero((lens >= lo_len) & (lens <= hi_len))[0]
        if cand.size:
            picks += list(rng.choice(cand, size=min(10, cand.size), replace=False))
    users = np.unique(np.array(picks, dtype=np.int64))
    mu = float(eng.mu.item())
    worst, worst_b = 0.0, 0.0
    for u in users:
        lo, hi = int(uptr[u]), int(uptr[u + 1])
        cols = eng.csr.indices[lo:hi].long()
        Zr = eng.Z[cols, :k].double().cpu().numpy()
        r = eng.csr.vals[lo:hi].double().cpu().numpy()
        bi = eng.b_i[cols].double().cpu().numpy()
        rho = r - mu - float(b_u_old[u]) - bi
        A = Zr.T @ Zr + (model.lambda_u + EPS) * np.eye(k)
        x = eng.U[u, :k].double().cpu().numpy()
        worst = max(worst, rel_residual(A, x, Zr.T @ rho))
        b_new = (r - mu - bi - Zr @ x).sum() / ((hi - lo) + model.lambda_bu + EPS)
        worst_b = max(worst_b, abs(float(eng.b_u[u]) - b_new))
    out["user_step"] = {"rows_checked": int(users.size), "longest_row": int(lens.max()),
                        "worst_relative_residual": worst, "worst_bias_error": worst_b}

    # ---- V-step + Gauss-Seidel sweep --------------------------------------------------------------
    V_old, b_i_old = eng.V.clone(), eng.b_i.clone()
    eng.item_step(False)
    torch.cuda.synchronize()
    eng._check_status()
    iptr = eng.csc.indptr.cpu().numpy()
    ilen = np.diff(iptr)
    picks = [0, n - 1, int(ilen.argmax()), int(ilen.argmin())]
    if S is not None:
        picks.append(int(np.argmax(eng.sched.level)))
    for lo_len, hi_len in ((1, 64), (65, 96), (97, 4096), (4097, 1 << 40)):
        cand = np.nonzero((ilen >= lo_len) & (ilen <= hi_len))[0]
        if cand.size:
            picks += list(rng.choice(cand, size=min(10, cand.size), replace=False))
    items = np.unique(np.array(picks, dtype=np.int64))
    lam_row = eng.lam_v_row.cpu().numpy()
    dex = eng.diag_extra.cpu().numpy() if S is not None else np.zeros(n)
    sp = eng.S_ptr.cpu().numpy() if S is not None else None
    worst, worst_b = 0.0, 0.0
    for i in items:
        lo, hi = int(iptr[i]), int(iptr[i + 1])
        rows = eng.csc.indices[lo:hi].long()
        Ur = eng.U[rows, :k].double().cpu().numpy()
        r = eng.csc.vals[lo:hi].double().cpu().numpy()
        bu = eng.b_u[rows].double().cpu().numpy()
        rho = r - mu - bu - float(b_i_old[i])
        A = Ur.T @ Ur + (float(lam_row[i]) + EPS + float(dex[i])) * np.eye(k)
        rhs = Ur.T @ rho
        if S is not None:
            nb = eng.S_idx[int(sp[i]):int(sp[i + 1])].long()
            sv = eng.S_val[int(sp[i]):int(sp[i + 1])].double().cpu().numpy()
            nb_h = nb.cpu().numpy()
            live = np.where((nb_h < i)[:, None], eng.V[nb, :k].double().cpu().numpy(),
                            V_old[nb, :k].double().cpu().numpy())
            rhs = rhs + model.alpha * (sv @ live)
        x = eng.V[i, :k].double().cpu().numpy()
        worst = max(worst, rel_residual(A, x, rhs))
        b_new = (r - mu - bu - Ur @ x).sum() / ((hi - lo) + model.lambda_bi + EPS)
        worst_b = max(worst_b, abs(float(eng.b_i[i]) - b_new))
    out["item_step"] = {"rows_checked": int(items.size), "longest_row": int(ilen.max()),
                        "worst_relative_residual": worst, "worst_bias_error": worst_b}
    out["pass"] = bool(out["user_step"]["worst_relative_residual"] < 2e-6 and out["user_step"]["worst_bias_error"] < 2e-5
                       and out["item_step"]["worst_relative_residual"] < 2e-6 and out["item_step"]["worst_bias_error"] < 2e-5)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
