#!/usr/bin/env python3
"""Host cost of one iteration at a launch-bound size (cfg 2: 6040 x 3706, 1M ratings, k = 32): wall time per
iteration over 200 iterations against the GPU time of its launches, and a cProfile of the iteration body."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig  # noqa: E402

dev = torch.device("cuda", 0)
m, n, nnz, k = bench.SIZES[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
csr, csc = bench.gen_ratings(dev, m, n, nnz, seed=1004)
N = 220
cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=N, lambda_u=5.0, lambda_v=6.0), biases=BiasesConfig(3.0, 2.0))
model = ALS(cfg, device=dev)
eng = model.prepare_csr(csr, csc, (m, n))
for it in range(20):
    eng.iteration(it, N)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for it in range(20, 120):
    eng.iteration(it, N)
t_host = time.perf_counter() - t0
b.record()
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f"100 iterations: host enqueue {1e3 * t_host / 100:.3f} ms/iter, wall {1e3 * t_wall / 100:.3f} ms/iter, "
      f"GPU span {a.elapsed_time(b) / 100:.3f} ms/iter")
pr = cProfile.Profile()
pr.enable()
for it in range(120, 220):
    eng.iteration(it, N)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3000])
