#!/usr/bin/env python3
"""Debug probe: run single half-steps of a bench size and report NaNs / status words / factor scale after each."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

size = sys.argv[1] if len(sys.argv) > 1 else "cfg5-small"
dev = torch.device("cuda", 0)
inp = bench.make_inputs(size, dev, 0, False)
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig
m, n, _, k = bench.SIZES[size]
csr, csc, S, features = inp[:4]
cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=3, lambda_u=5.0, lambda_v=6.0, random_state=42,
                                pop_reg_mode="inverse_sqrt" if size in ("cfg5-small", "cfg5") else None),
                biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0),
                graph=(GraphConfig(alpha=0.5, sim=GraphSimConfig(source="precomputed", topk=50)) if S is not None else GraphConfig()))
model = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0} if features else None, device=dev,
            solve_dtype=os.environ.get("PROBE_DTYPE", "auto"), gram=os.environ.get("PROBE_GRAM", "f16x2"))
eng = model.prepare_csr(csr, csc, (m, n), features=features, S=S)
if features:
    eng.be.compose_z(eng.V, eng.Xcat, eng.Wcat, eng.Z)

def rep(tag):
    torch.cuda.synchronize()
    w = eng.ctrl[64:76].view(torch.int32).cpu().tolist()
    print(f"{tag}: status {w} fscale {eng.be._fscale.cpu().tolist()} U nan {int(torch.isnan(eng.U).sum())} max {float(eng.U.abs().max()):.4g} "
          f"V nan {int(torch.isnan(eng.V).sum())} max {float(eng.V.abs().max()):.4g} Z max {float(eng.Z.abs().max()):.4g}", flush=True)

rep("init")
for it in range(2):
    eng.user_step(); rep(f"it{it} user")
    b_old = eng.b_i.clone(); eng.b_i_prev = b_old
    eng.item_step(want_gram=True); rep(f"it{it} item")
    if eng.feat_names:
        if eng.gram is not None:
            g = eng.gram.base if hasattr(eng.gram, "base") else eng.gram
            print("   gram nan", int(torch.isnan(g).sum()), "max", float(torch.nan_to_num(g).abs().max()), flush=True)
        eng.w_step(b_old); rep(f"it{it} w")
    eng.stats_step(it); rep(f"it{it} stats")
    print("   hist", eng.hist[it].cpu().tolist(), flush=True)

# ---- which rows differ between this mode and the f32-Gram / float32 reference run of the U-step from the same state?
if os.environ.get("PROBE_COMPARE"):
    import numpy as np
    def ustep(gram, dtype):
        mdl = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0} if features else None, device=dev, solve_dtype=dtype, gram=gram)
        e = mdl.prepare_csr(csr, csc, (m, n), features=features, S=S)
        if features:
            e.be.compose_z(e.V, e.Xcat, e.Wcat, e.Z)
        e.user_step()
        torch.cuda.synchronize()
        return e.U[: m, :k].double().cpu().numpy(), e
    Ua, ea = ustep("f16x2", "float32")
    Ub, eb = ustep("f32", "float32")
    Uc, ec = ustep("f16x2", "float64")
    lens = (ea.csr.indptr[1:] - ea.csr.indptr[:-1]).cpu().numpy()
    for name, X in (("f16x2/f32", Ua), ("f32/f32", Ub)):
        bad = ~np.isfinite(X).all(axis=1)
        err = np.where(bad, np.inf, np.abs(X - Uc).max(axis=1) / np.abs(Uc).max())
        order = np.argsort(-err)[:40]
        print(name, "rows with NaN:", int(bad.sum()), "lens of NaN rows:", sorted(lens[bad].tolist())[:40])
        print(name, "worst rows (len, err):", [(int(lens[r]), float(f"{err[r]:.2e}")) for r in order])
        for lo, hi in ((1, 16), (17, 32), (33, 48), (49, 64), (65, 80), (81, 96), (97, 128), (129, 512), (513, 10**9)):
            sel = (lens >= lo) & (lens <= hi) & ~bad
            if sel.any():
                print(f"   len {lo}-{hi}: rows {int(sel.sum())} max err {err[sel].max():.2e} median {np.median(err[sel]):.2e}")
