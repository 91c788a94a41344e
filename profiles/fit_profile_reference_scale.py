import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from tests.synth import make_ratings, make_features
from collaborative_filtering_amd import ALS, ALSConfig, CoreConfig, BiasesConfig, GraphConfig, GraphSimConfig, layout
import collaborative_filtering_amd.als as A
m, n, nnz = 610, 4980, 100000
rows, cols, vals = make_ratings(m, n, nnz, 5)
G, Y = make_features(n, 6); feats = {"genres": G, "years": Y}
cfg = ALSConfig(core=CoreConfig(n_factors=64, n_iters=20, lambda_u=5.0, lambda_v=6.0, random_state=42), biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    csr, csc = layout.coo_to_sides(rows, cols, vals, (m, n))
    t1 = time.perf_counter()
    model = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0}, device="cuda:0")
    model._fit_sides(csr, csc, feats, None, 0, 0, None, run=False)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    model._eng.run(None, 0, 0)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    model._eng.export(model)
    t4 = time.perf_counter()
    idx = np.arange(0, m * n, 37)
    p = model.predict_at(idx, features=feats)
    t5 = time.perf_counter()
    print(f"rep {rep}: coo_to_sides {1e3*(t1-t0):.1f} ms, engine init {1e3*(t2-t1):.1f}, 20 iterations {1e3*(t3-t2):.1f}, export {1e3*(t4-t3):.1f}, predict_at({idx.size}) {1e3*(t5-t4):.1f}")
