#!/usr/bin/env python3
"""Run-to-run bitwise comparison of one full ALS iteration on a bench.py workload (default cfg5 at full size):
two engines built from the same generated inputs, one after the other; reports which state arrays differ.
Usage (GPU box): python3 profiles/check_determinism.py [--size cfg5]"""
import argparse, gc, json, os, sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="cfg5")
    ap.add_argument("--iters", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    m, n, nnz, k = bench.SIZES[args.size]
    features = bench.gen_features(n, 3004) if args.size in ("cfg3", "cfg5-small", "cfg5") else None
    csr, csc = bench.gen_ratings(dev, m, n, nnz, seed=1004)
    S = bench.gen_graph(dev, n, seed=2004) if args.size not in ("cfg2", "cfg3") else None
    outs = []
    for run in range(2):
        cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=3, lambda_u=5.0, lambda_v=6.0, random_state=42,
                                        pop_reg_mode="inverse_sqrt" if args.size in ("cfg5-small", "cfg5") else None),
                        biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0),
                        graph=(GraphConfig(alpha=0.5, sim=GraphSimConfig(source="precomputed", topk=50))
                               if S is not None else GraphConfig()))
        model = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0} if features else None, device=dev)
        eng = model.prepare_csr(csr, csc, (m, n), features=features, S=S)
        if features:
            eng.be.compose_z(eng.V, eng.Xcat, eng.Wcat, eng.Z)
        state = {}
        eng.user_step()
        torch.cuda.synchronize()
        state["U_after_first_user_step"] = eng.U.clone()
        state["b_u_after_first_user_step"] = eng.b_u.clone()
        for it in range(args.iters):
            eng.iteration(it, 3)
        torch.cuda.synchronize()
        eng._check_status()
        state.update(U=eng.U.clone(), V=eng.V.clone(), b_u=eng.b_u.clone(), b_i=eng.b_i.clone(),
                     hist=eng.hist[:args.iters].clone(), mu=eng.mu.clone())
        if features:
            state["Wcat"] = eng.Wcat.clone()
        outs.append(state)
        del model, eng
        gc.collect()
        torch.cuda.empty_cache()
    rep = {"size": args.size, "gs_form": os.environ.get("ALS_GS_FORM", "auto"), "iterations": args.iters}
    for key in outs[0]:
        a, b = outs[0][key], outs[1][key]
        eq = bool(torch.equal(a, b))
        rep[key] = "bitwise equal" if eq else {
            "max_abs_diff": float((a.double() - b.double()).abs().max()),
            "n_different": int((a != b).sum()), "numel": int(a.numel())}
    print(json.dumps(rep))


if __name__ == "__main__":
    main()
