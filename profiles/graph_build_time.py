#!/usr/bin/env python3
"""Build time of the similarity graph by the product's kernels (csrc/graph_build.hip) at the BASELINE item counts:
genre-like binary features (19 columns at the shipped column rates) + one small continuous column that breaks the
massive ties of binary features at random (bench.py's workload), top-k = 50.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from collaborative_filtering_amd import _hip, layout  # noqa: E402

dev = torch.device("cuda", 0)
lib = _hip.load()
out = {}
for n in [int(a) for a in (sys.argv[1:] or ["4980", "100000", "1000000"])]:
    X = bench.graph_features(dev, n, seed=2004)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ptr, idx, val, D = layout.build_similarity_kernel(lib, X, 50, 1e-8, dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ptr, idx, val, D = layout.build_similarity_kernel(lib, X, 50, 1e-8, dev)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    deg = (ptr[1:] - ptr[:-1])
    out[str(n)] = {"first_call_s": t1 - t0, "second_call_s": t2 - t1, "graph_nnz": int(idx.numel()),
                   "mean_row": float(deg.float().mean()), "max_row": int(deg.max()),
                   "pairs_per_s": n * n / (t2 - t1)}
    print(n, out[str(n)], flush=True)
print(json.dumps(out))
