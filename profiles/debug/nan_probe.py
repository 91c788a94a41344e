#!/usr/bin/env python3
"""Debug probe: fit a fixture with every backend call synchronised and its tensor arguments scanned for NaN / inf."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests.common import Golden
from tests.test_gpu_parity import _model_for
from collaborative_filtering_amd.backend import HipBackend
name = sys.argv[1]
kw = {"solve_dtype": sys.argv[2]} if len(sys.argv) > 2 else {}
def bad(v):
    t = getattr(v, "base", v)
    return torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t).all())
for meth in ("row_solve", "w_item_vectors", "w_accumulate", "spd_solve", "item_stats", "gs_level", "gs_levels", "gs_dataflow",
             "sum_pairs", "history_row", "compose_z"):
    orig = getattr(HipBackend, meth)
    def wrap(self, *a, _o=orig, _m=meth, **kw):
        pre = [k for k, v in kw.items() if bad(v)]
        if _m == "spd_solve":
            A, b, dadd = a[0], a[1], a[2]
            ev = torch.linalg.eigvalsh(A)
            print("spd_solve input: N", A.shape[0], "asym", float((A - A.T).abs().max()), "eig min/max", float(ev.min()), float(ev.max()),
                  "diag_add", dadd, "|b|", float(b.abs().max()), flush=True)
        r = _o(self, *a, **kw)
        torch.cuda.synchronize()
        if _m == "row_solve" and kw.get("gram_out") is not None:
            import numpy as np
            from collaborative_filtering_amd import layout
            k, ld, side = kw["k"], kw["ld"], kw["side"]
            G = getattr(kw["gram_out"], "base", kw["gram_out"]).double().cpu().numpy().reshape(-1, ld, ld)
            F = kw["F"].double().cpu().numpy()
            ptr = side.indptr.cpu().numpy(); idx = side.indices.cpu().numpy()
            pos = layout.perm_of_col(k)[:k]
            worst = 0.0
            for i in range(min(G.shape[0], 200)):
                cols = idx[ptr[i]:ptr[i + 1]]
                ref = F[cols, :k].T @ F[cols, :k]
                got = G[i][np.ix_(pos, pos)]
                blk = pos // 16
                low = blk[:, None] >= blk[None, :]
                worst = max(worst, float(np.abs(got - ref)[low].max()) / max(float(np.abs(ref).max()), 1e-30))
            print("row_solve gram_out vs F^T F: worst relative error over items", worst, "f64" , kw.get("f64"), flush=True)
        if _m == "w_accumulate":
            import numpy as np
            from tests.cpu_backend import NumpyBackend
            k, ld = kw["k"], kw["ld"]
            gram = getattr(kw["gram"], "base", kw["gram"]).cpu()
            H = getattr(kw["H"], "base", kw["H"]).cpu()
            Aref, Bref = NumpyBackend().w_accumulate(k=k, ld=ld, item_begin=kw["item_begin"], item_end=kw["item_end"], gram=gram,
                                                     X=kw["X"].cpu(), H=H, feat_index=kw["feat_index"], feat_col0=kw["feat_col0"],
                                                     feat_d=kw["feat_d"])
            A, B = r
            try:
                g64 = getattr(kw["gram"], "base", kw["gram"]).double()
                h64 = getattr(kw["H"], "base", kw["H"]).double()
                kw2 = dict(kw); kw2.update(gram=g64, H=h64, f64=True)
                A2, B2 = _o(self, *a, **kw2)
                torch.cuda.synchronize()
                print("   same call with double copies and f64=True: max|A2 - ref|/max|ref|", float((A2.cpu() - Aref).abs().max() / Aref.abs().max()), flush=True)
            except Exception as e:
                print("   f64 variant raised", e)
            print("w_accumulate feat", kw["feat_index"], "d", kw["feat_d"], "max|A - ref|/max|ref|",
                  float((A.cpu() - Aref).abs().max() / Aref.abs().max()), "B:", float((B.cpu() - Bref).abs().max() / Bref.abs().max().clamp_min(1e-30)),
                  "| A[0,:4]", A[0, :4].cpu().tolist(), "ref", Aref[0, :4].tolist(), "| B[:4]", B[:4].cpu().tolist(), "ref", Bref[:4].tolist(),
                  "| median ratio A/ref", float((A.cpu() / Aref)[Aref.abs() > 1e-3 * Aref.abs().max()].median()),
                  "gram dtype", gram.dtype, "H dtype", H.dtype, "gram finite", bool(torch.isfinite(torch.nan_to_num(gram, nan=0.0)).all()), flush=True)
        if _m == "spd_solve":
            print("   status", int(a[3].item()), "x finite", bool(torch.isfinite(r).all()), flush=True)
        post = [k for k, v in kw.items() if bad(v)]
        outs = r if isinstance(r, tuple) else (r,)
        rb = [i for i, v in enumerate(outs) if bad(v)]
        if pre or post or rb:
            print(_m, "non-finite before:", pre, "after:", post, "returned:", rb, "status", self and None, flush=True)
        return r
    setattr(HipBackend, meth, wrap)
g = Golden(name)
r, c, v = g.train
model = _model_for(g, device="cuda:0", **kw)
try:
    model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"], min_iters=g.cfg["min_iters"], verbose=0)
    print("fit ok", model.history["train_rmse"])
except Exception as e:
    print("fit raised", type(e).__name__, e)
