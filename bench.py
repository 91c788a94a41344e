#!/usr/bin/env python3
"""Headline benchmark: ratings/sec per ALS iteration at k=64 (BASELINE.json).

Workload (`config.workload`): BASELINE.json configs[3] - synthetic 1M users x
100K items, 100M ratings, k=64, bias terms + graph-Laplacian regularisation
(sparse S, 50 neighbours per item before symmetrisation, alpha = 0.5); inputs
are generated on the GPU following SURVEY.md section 8(d) and are resident in
HBM before the timed region.  A "step" is one full ALS iteration: U-step,
V-step (Gram + factor, Gauss-Seidel sweep), mu / RMSE / norm statistics.
The problem size is fixed as GPUs are added (users sharded for the U-step,
items for the V-step, RCCL all-gathers in between): "scaling": "strong".

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement), with
`roofline` for the dominant kernel (als_row_solve's k_row_tasks, timed with
events on its stream) and `cpu_baseline` (the numpy oracle - a port of the
reference's CPU path - on a bounded row sample, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GENRE_RATES = [0.1869, 0.1307, 0.0516, 0.0719, 0.3847, 0.1217, 0.0394, 0.4524, 0.0779, 0.01,
               0.099, 0.0133, 0.0378, 0.0568, 0.1749, 0.0956, 0.199, 0.0408, 0.0177]

SIZES = {   # name: (m, n, nnz, k)   - cfg4 is the headline; the others are diagnostics
    "cfg2": (6_040, 3_706, 1_000_000, 32),                # BASELINE configs[1]: +bias, no graph
    "cfg3": (138_493, 26_744, 20_000_000, 64),            # BASELINE configs[2]: +W_f (genres, years), no graph
    "cfg4": (1_000_000, 100_000, 100_000_000, 64),
    "cfg4-small": (100_000, 10_000, 5_000_000, 64),       # rehearsal size
    "cfg5-small": (200_000, 20_000, 10_000_000, 128),     # BASELINE configs[4] shape / 50: full model, k = 128
    "cfg5": (10_000_000, 1_000_000, 1_000_000_000, 128),  # BASELINE configs[4] at full size on ONE GPU (~170 GB)
    "k48": (1_000_000, 100_000, 100_000_000, 48),         # cfg4 shape at KB = 3 (occupancy experiments)
    "k80": (100_000, 10_000, 5_000_000, 80),              # rehearsal size at other ranks (KB = 5, 6, 10)
    "k96": (100_000, 10_000, 5_000_000, 96),
    "k160": (100_000, 10_000, 5_000_000, 160),
    "tiny": (4_000, 1_500, 150_000, 64),
}


def gen_ratings(dev, m, n, nnz, seed):
    """Power-law sparsity, planted rank-8 model, half-star values; CSR + CSC on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    cu = torch.cumsum(torch.arange(1, m + 1, device=dev, dtype=torch.float64) ** -0.6, 0)
    ci = torch.cumsum(torch.arange(1, n + 1, device=dev, dtype=torch.float64) ** -0.9, 0)
    cu /= cu[-1].clone()
    ci /= ci[-1].clone()
    uperm = torch.randperm(m, device=dev, generator=g)
    iperm = torch.randperm(n, device=dev, generator=g)
    # every user and every item gets at least one rating
    au = torch.arange(m, device=dev)
    ai = torch.arange(n, device=dev)
    diag = torch.unique(torch.cat([au * n + (au % n), (ai % m) * n + ai]))
    keys = diag
    while keys.numel() < nnz:
        draw = max(int(1.15 * (nnz - keys.numel())), 1 << 16)
        u = uperm[torch.searchsorted(cu, torch.rand(draw, device=dev, dtype=torch.float64, generator=g)).clamp_(max=m - 1)]
        i = iperm[torch.searchsorted(ci, torch.rand(draw, device=dev, dtype=torch.float64, generator=g)).clamp_(max=n - 1)]
        keys = torch.unique(torch.cat([keys, u * n + i]))
        del u, i
    excess = keys.numel() - nnz
    if excess > 0:
        is_diag = torch.isin(keys, diag)
        cand = torch.nonzero(~is_diag).squeeze(1)
        drop = cand[torch.randperm(cand.numel(), device=dev, generator=g)[:excess]]
        keep = torch.ones(keys.numel(), dtype=torch.bool, device=dev)
        keep[drop] = False
        keys = keys[keep]
        del is_diag, cand, drop, keep
    u = torch.div(keys, n, rounding_mode="floor")
    i = keys - u * n
    us = torch.randn(m, 8, device=dev, generator=g) * 0.5
    vs = torch.randn(n, 8, device=dev, generator=g) * 0.5
    raw = 3.5 + (us[u] * vs[i]).sum(1) + 0.5 * torch.randn(keys.numel(), device=dev, generator=g)
    vals = (torch.round(raw * 2.0) / 2.0).clamp_(0.5, 5.0).to(torch.float32)
    del raw, us, vs
    uptr = torch.zeros(m + 1, dtype=torch.int64, device=dev)
    uptr[1:] = torch.cumsum(torch.bincount(u, minlength=m), 0)
    csr = (uptr, i.to(torch.int32), vals)
    order = torch.argsort(i * m + u)
    iptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    iptr[1:] = torch.cumsum(torch.bincount(i, minlength=n), 0)
    csc = (iptr, u[order].to(torch.int32), vals[order])
    return csr, csc


def graph_features(dev, n, seed):
    """Item features the benchmark graph is built on: 19 genre-like binary columns at the shipped file's column
    rates plus one small continuous column (N(0, 0.05^2)) - binary genres alone tie massively at the top-k
    boundary; the extra column breaks those ties at random, which is what the reference's
    argpartition does in effect.  float32 [n, 20] on the device."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rates = torch.tensor(GENRE_RATES, device=dev)
    G = (torch.rand(n, 19, device=dev, generator=g) < rates).to(torch.float32)
    jitter = 0.05 * torch.randn(n, 1, device=dev, generator=g)
    return torch.cat([G, jitter], dim=1)


def gen_graph_product(dev, n, seed, topk=50):
    """The benchmark's item graph built by the PRODUCT's own kernels (csrc/graph_build.hip through
    layout.build_similarity_kernel): exact top-`topk` cosine neighbours of every item over all n items,
    symmetrised by max.  Returns ((ptr, idx, val), seconds)."""
    from collaborative_filtering_amd import _hip, layout
    X = graph_features(dev, n, seed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ptr, idx, val, _ = layout.build_similarity_kernel(_hip.load(), X, topk, 1e-8, dev)
    torch.cuda.synchronize()
    return (ptr, idx, val), time.perf_counter() - t0


def gen_graph(dev, n, seed, topk=50, ncand=512):
    """Sparse item graph: each item keeps its `topk` genre-cosine neighbours among `ncand`
    random candidates; symmetrised by max; zero similarities are not edges."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    rates = torch.tensor(GENRE_RATES, device=dev)
    G = (torch.rand(n, 19, device=dev, generator=g) < rates).to(torch.float32)
    G = G / torch.sqrt((G * G).sum(1, keepdim=True)).clamp_min(1e-8)
    ncand = min(ncand, n - 1)
    topk = min(topk, ncand)
    rows, cols, vals = [], [], []
    step = 8192
    for b in range(0, n, step):
        e = min(b + step, n)
        cand = torch.randint(0, n, (e - b, ncand), device=dev, generator=g)
        sims = torch.einsum("bd,bcd->bc", G[b:e], G[cand])
        sims[cand == torch.arange(b, e, device=dev)[:, None]] = 0.0
        tv, ti = torch.topk(sims, topk, dim=1)
        rows.append(torch.arange(b, e, device=dev)[:, None].expand(-1, topk).reshape(-1))
        cols.append(torch.gather(cand, 1, ti).reshape(-1))
        vals.append(tv.reshape(-1))
    r, c, v = torch.cat(rows), torch.cat(cols), torch.cat(vals)
    ok = v > 0
    r, c, v = r[ok], c[ok], v[ok]
    key = torch.cat([r * n + c, c * n + r])
    val = torch.cat([v, v])
    uk, inv = torch.unique(key, return_inverse=True)
    sv = torch.zeros(uk.numel(), device=dev).scatter_reduce_(0, inv, val, reduce="amax")
    sr = torch.div(uk, n, rounding_mode="floor")
    sc = uk - sr * n
    ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(torch.bincount(sr, minlength=n), 0)
    return ptr, sc.to(torch.int32), sv.to(torch.float32)


def gen_features(n, seed):
    """genres: n x 19 Bernoulli at the shipped file's column rates, row-L2; years: n x 1 z-scored."""
    rng = np.random.default_rng(seed)
    G = (rng.random((n, 19)) < np.asarray(GENRE_RATES)[None, :]).astype(np.float64)
    G = (G / np.maximum(np.sqrt((G * G).sum(1, keepdims=True)), 1e-8)).astype(np.float32)
    y = rng.normal(size=(n, 1))
    return {"genres": G, "years": ((y - y.mean()) / y.std()).astype(np.float32)}


def _sample_rows(ptr_d, idx_d, val_d, nrows, cap_ratings, seed, dev):
    """A seeded uniform random sample of rows (without replacement, in random order) holding about
    `cap_ratings` ratings: (row ids, compact indptr, indices, values) on the host."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    perm = torch.randperm(nrows, generator=g).to(dev)
    lens = (ptr_d[1:] - ptr_d[:-1])[perm]
    csum = torch.cumsum(lens, 0)
    take = int(torch.searchsorted(csum, torch.tensor([cap_ratings], device=dev)).item()) + 1
    take = max(1, min(take, nrows))
    rows, lens = perm[:take], lens[:take]
    ptr = torch.zeros(take + 1, dtype=torch.int64, device=dev)
    ptr[1:] = torch.cumsum(lens, 0)
    pos = torch.repeat_interleave(ptr_d[rows] - ptr[:-1], lens) + torch.arange(int(ptr[-1]), device=dev)
    return rows.cpu().numpy(), ptr.cpu().numpy(), idx_d[pos].cpu().numpy().astype(np.int64), val_d[pos].double().cpu().numpy()


def cpu_baseline(eng, budget_s=20.0):
    """The oracle (numpy port of the reference's per-row loops, float64) on a bounded RANDOM sample of the same
    workload: seeded uniform samples of users (U-step) and of items (V-step, with the Laplacian term), worked
    through in random order until the time budget is spent.  ratings/s per iteration = 1 / (s per rating of the
    U-step + s per rating of the V-step)."""
    from oracle.als_oracle import OracleALS, OracleConfig, Ratings
    md = eng.model
    k = eng.k
    o = OracleALS(OracleConfig(n_factors=k, n_iters=1, lambda_u=md.lambda_u, lambda_v=md.lambda_v,
                               lambda_bu=md.lambda_bu, lambda_bi=md.lambda_bi, alpha=md.alpha,
                               sim={"feature_name": "genres"} if eng.use_graph else None))
    V = eng.V[: eng.n, :k].double().cpu().numpy()
    b_i = eng.b_i[: eng.n].double().cpu().numpy()
    mu = float(eng.mu.item())
    cap = 12_000_000          # ratings per sample: ~15-20 s of oracle time for both half-steps together
    # ---- users: a random sample, as a compact CSR over all items
    urows, uptr, uidx, uval = _sample_rows(eng.csr.indptr, eng.csr.indices, eng.csr.vals, eng.m, cap, 4001, eng.dev)
    su = urows.size
    rt = Ratings(su, eng.n, None, uidx, uval, uptr, None, None, None)
    o.mu, o.V, o.b_i = mu, V, b_i
    o.U = np.zeros((su, k))
    o.b_u = eng.b_u[torch.from_numpy(urows).to(eng.dev)].double().cpu().numpy()
    t0 = time.perf_counter()
    done_u, rows_u = 0, 0
    for blk in range(0, su, 256):
        o.user_step(rt, V, rows=range(blk, min(blk + 256, su)))
        rows_u = min(blk + 256, su)
        done_u = int(uptr[rows_u])
        if time.perf_counter() - t0 > budget_s / 2:
            break
    tu = time.perf_counter() - t0
    # ---- items: a random sample of columns; users remapped to a compact range, item ids kept (graph rows)
    irows, iptr, iusers, ival = _sample_rows(eng.csc.indptr, eng.csc.indices, eng.csc.vals, eng.n, cap, 4002, eng.dev)
    si = irows.size
    uniq, compact = np.unique(iusers, return_inverse=True)
    o.U = eng.U[torch.from_numpy(uniq).to(eng.dev), :k].double().cpu().numpy()
    o.b_u = eng.b_u[torch.from_numpy(uniq).to(eng.dev)].double().cpu().numpy()
    o.V = V.copy()
    o.b_i = b_i.copy()
    # scatter the sampled columns back to their item ids (empty columns elsewhere): the Laplacian term of item i
    # then reads the real neighbours of i
    cnt = np.zeros(eng.n, dtype=np.int64)
    cnt[irows] = np.diff(iptr)
    full_ptr = np.zeros(eng.n + 1, dtype=np.int64)
    np.cumsum(cnt, out=full_ptr[1:])
    order = np.argsort(irows, kind="stable")
    src = np.concatenate([np.arange(iptr[j], iptr[j + 1]) for j in order]) if si else np.zeros(0, np.int64)
    rt2 = Ratings(uniq.size, eng.n, None, None, None, None, full_ptr, compact[src], ival[src])
    o.lambda_v_i = np.full(eng.n, float(md.lambda_v))
    o.lambda_bi_i = np.full(eng.n, float(md.lambda_bi))
    o.use_graph = eng.use_graph
    if eng.use_graph:
        o.S_csr = (eng.S_ptr.cpu().numpy(), eng.S_idx.cpu().numpy().astype(np.int64), eng.S_val.cpu().numpy())
        o.D = (eng.diag_extra[: eng.n] / np.float32(md.alpha)).cpu().numpy()
    t0 = time.perf_counter()
    done_i, cols_i = 0, 0
    for blk in range(0, si, 32):
        o.item_step(rt2, cols=[int(c) for c in irows[blk:blk + 32]])
        cols_i = min(blk + 32, si)
        done_i = int(iptr[cols_i])
        if time.perf_counter() - t0 > budget_s / 2:
            break
    tv = time.perf_counter() - t0
    per_rating = tu / max(done_u, 1) + tv / max(done_i, 1)
    return {"value": 1.0 / per_rating, "unit": "ratings/s per ALS iteration",
            "cores": 1, "kind": "port",
            "sample": (f"oracle/als_oracle.py (float64 numpy+scipy per-row loop) on seeded uniform random row samples: "
                       f"U-step over {rows_u} random users ({done_u} ratings, {tu:.1f} s) + V-step with Laplacian term "
                       f"over {cols_i} random items ({done_i} ratings, {tv:.1f} s); host has {os.cpu_count()} cores, "
                       f"loop is single-threaded")}


def _sync(dev):
    if dev.type == "cuda":
        torch.cuda.synchronize(dev)


def make_inputs(size, dev, rank, dist_on, graph_kind="sampled", no_graph=False):
    """Synthetic inputs of a named size, resident on `dev`: rank 0 generates, everybody receives the same bytes
    (robust against RNG differences).  Returns (csr, csc, S, features, graph_build_s, seconds spent here)."""
    m, n, nnz, k = SIZES[size]
    t0 = time.perf_counter()
    use_graph = (not no_graph) and size not in ("cfg2", "cfg3")
    features = gen_features(n, 3004) if size in ("cfg3", "cfg5-small", "cfg5") else None
    graph_build_s = None
    csr = csc = S = None
    if rank == 0:
        csr, csc = gen_ratings(dev, m, n, nnz, seed=1004)
        if not use_graph:
            S = None
        elif graph_kind == "product":
            S, graph_build_s = gen_graph_product(dev, n, seed=2004)
        else:
            S = gen_graph(dev, n, seed=2004)
    if dist_on:
        def bc(t, dtype, numel):
            if rank != 0:
                t = torch.empty(numel, dtype=dtype, device=dev)
            dist.broadcast(t, 0)
            return t
        meta = torch.tensor([csr[1].numel() if rank == 0 else 0,
                             (S[1].numel() if (rank == 0 and S is not None) else 0)], device=dev)
        dist.broadcast(meta, 0)
        N, NS = int(meta[0]), int(meta[1])
        csr = tuple(bc(csr[j] if rank == 0 else None, dt, sz) for j, (dt, sz) in
                    enumerate([(torch.int64, m + 1), (torch.int32, N), (torch.float32, N)]))
        csc = tuple(bc(csc[j] if rank == 0 else None, dt, sz) for j, (dt, sz) in
                    enumerate([(torch.int64, n + 1), (torch.int32, N), (torch.float32, N)]))
        if use_graph:
            S = tuple(bc(S[j] if rank == 0 else None, dt, sz) for j, (dt, sz) in
                      enumerate([(torch.int64, n + 1), (torch.int32, NS), (torch.float32, NS)]))
        else:
            S = None
    _sync(dev)
    return csr, csc, S, features, graph_build_s, time.perf_counter() - t0


def run_case(size, inputs, dev, *, steps, warmup, dist_on=False, gs_mode=None, gram="f16x2", hip_graph=False,
             solve_dtype="auto", backend=None):
    """Set the model up on resident inputs and time `steps` iterations after `warmup` (barrier + synchronize on
    both sides, MAX over ranks).  Returns a dict: elapsed seconds, per-phase event times, the engine, set-up time
    of the PRODUCT alone (upload / task lists / schedules / initial factors - no data generation)."""
    from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig
    m, n, _, k = SIZES[size]
    csr, csc, S, features = inputs[:4]
    n_total = warmup + steps
    cfg = ALSConfig(core=CoreConfig(n_factors=k, n_iters=n_total, lambda_u=5.0, lambda_v=6.0, random_state=42,
                                    pop_reg_mode="inverse_sqrt" if size in ("cfg5-small", "cfg5") else None),
                    biases=BiasesConfig(lambda_bu=3.0, lambda_bi=2.0),
                    graph=(GraphConfig(alpha=0.5, sim=GraphSimConfig(source="precomputed", topk=50))
                           if S is not None else GraphConfig()))
    t_setup = time.perf_counter()
    model = ALS(cfg, lambda_w={"genres": 5.0, "years": 10.0} if features else None,
                device=dev, gs_mode=gs_mode, gram=gram, hip_graph=hip_graph, backend=backend,
                solve_dtype=solve_dtype, process_group="world" if dist_on else None)
    eng = model.prepare_csr(csr, csc, (m, n), features=features, S=S)
    if features:
        eng.be.compose_z(eng.V, eng.Xcat, eng.Wcat, eng.Z)
    _sync(dev)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if dist_on:
            dist.barrier()
        _sync(dev)

    for it in range(warmup):
        eng.iteration(it, n_total)
    barrier()
    eng.timers = None if (hip_graph or dev.type != "cuda") else []   # event pairs cannot be recorded inside a captured graph
    t0 = time.perf_counter()
    for it in range(warmup, n_total):
        eng.iteration(it, n_total)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    eng._check_status()
    phase = {}
    for name, a, b in (eng.timers or []):
        phase.setdefault(name, []).append(a.elapsed_time(b))
    eng.timers = None
    return {"model": model, "eng": eng, "elapsed": elapsed, "phase": phase, "setup_s": t_setup,
            "hist": eng.hist[: n_total].cpu().numpy()}


def frac_iter_of(size, nnz, elapsed_per_step):
    """[2 (4k+12) N + (m+n)(4k+12)] / t_iter as a fraction of 8 TB/s."""
    m, n, _, k = SIZES[size]
    return (2 * (4 * k + 12) * nnz + (m + n) * (4 * k + 12)) / elapsed_per_step / 1e9 / 8000.0


def main(argv=None, *, backend=None, device=None):
    """`backend` / `device`: test hook only (tests/test_dist_gloo.py runs this very code path on CPU ranks over gloo
    with the test-only numpy stand-in); the command line always runs the HIP backend on cuda:LOCAL_RANK."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", default="cfg4", choices=sorted(SIZES))
    ap.add_argument("--gs-mode", default=None, choices=[None, "exact", "block", "levels"],
                    help="Laplacian sweep with several ranks: exact (default; shards in rank order), block, levels")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary cases (N = 1, cfg4 only)")
    ap.add_argument("--no-graph", action="store_true", help="debug only; not a valid headline run")
    ap.add_argument("--hip-graph", action="store_true",
                    help="replay iterations as captured HIP graphs (no per-phase timing; for launch-bound sizes)")
    ap.add_argument("--gram", default="f16x2", choices=["f16x2", "f32"],
                    help="how K1 forms the Gram: 2-way fp16 split of the scaled floats on the fp16 matrix cores (fp32 "
                         "accumulate, fp32-level accuracy) or v_mfma_f32 on the raw floats")
    ap.add_argument("--graph", default="sampled", choices=["sampled", "product"],
                    help="item graph: SURVEY 8(d)'s workload definition (default; every item's top-50 genre-cosine "
                         "neighbours among 512 random candidates, symmetrised by max - the graph every number since "
                         "round 1 was measured on), or the exact top-50 over ALL items built by the product's own "
                         "kernels (csrc/graph_build.hip; 34 ms at n = 100K, 2.7 s at n = 1M - hub items with thousands "
                         "of neighbours then lengthen the sweep: 2.7 instead of 1.4 ms at cfg 4)")
    ap.add_argument("--solve-dtype", default="auto", choices=["auto", "float32", "float64"],
                    help="auto (the product's default): fp32 rows, ill-conditioned rows redone in fp64; float64: fp64 "
                         "Gram / Cholesky / substitutions for every row (accuracy mode; NOT the headline dtype)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 code path with several ranks on ONE GPU")
    args = ap.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device is None:
        ngpu = torch.cuda.device_count()
        local = local % max(ngpu, 1)
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    else:
        dev = torch.device(device)
    # ALS_FORCE_COLLECTIVES=1: one-rank rehearsal of the sharded path and its RCCL calls on a single GPU
    dist_on = world > 1 or os.environ.get("ALS_FORCE_COLLECTIVES") == "1"
    own_group = False
    if dist_on and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl" and dev.type == "cuda":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        own_group = True
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)

    m, n, nnz, k = SIZES[args.size]
    inputs = make_inputs(args.size, dev, rank, dist_on, args.graph, args.no_graph)
    csr, csc, S, features, graph_build_s, datagen_s = inputs
    use_graph = S is not None
    nnz = int(csr[1].numel())
    n_total = args.warmup + args.steps
    res = run_case(args.size, inputs, dev, steps=args.steps, warmup=args.warmup, dist_on=dist_on, gs_mode=args.gs_mode,
                   gram=args.gram, hip_graph=args.hip_graph, solve_dtype=args.solve_dtype, backend=backend)
    eng, elapsed, phase, hist, t_setup = res["eng"], res["elapsed"], res["phase"], res["hist"], res["setup_s"]

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        # algorithmic work of the row-solve launches of THIS rank:
        #   flops/rating/half-step = 2k^2 (Gram) + 2k (rhs) + 2k (bias), + k^3/3 + 2k^2 per solved row
        #   bytes/rating/half-step = 4 idx + 4 val + 4k factor row + 4 bias; + (4k + 12) per row
        rs_ms = phase.get("row_solve_user", []) + phase.get("row_solve_item", [])
        nn_u, nn_i = eng.utasks.nnz, eng.itasks.nnz
        rows_u = eng.ue - eng.ub
        rows_i = eng.ie - eng.ib
        fl = (2 * k * k + 4 * k) * (nn_u + nn_i) + (k ** 3 / 3 + 2 * k * k) * (rows_u + rows_i)
        by = (4 * k + 12) * (nn_u + nn_i) + (4 * k + 12) * (rows_u + rows_i)
        t_rs = max(1e-3 * sum(rs_ms) / args.steps, 1e-12)   # seconds per iteration in row-solve launches
        n_launch = 2
        # Primary view: HBM.  With the Gram on the bf16 matrix cores the binding resources of this kernel are
        # the gather of 256-B factor rows (V-step launch: U does not fit the caches) and VALU issue (U-step
        # launch); the fp32-matrix view (k = 64 was fp32-MFMA-bound) is kept as `mfma_view`.
        # the iteration's algorithmic bytes over the whole iteration time
        by_iter = 2 * (4 * k + 12) * nnz + (m + n) * (4 * k + 12)
        roof = {"kernel": f"k_row_tasks<KB={-(-k // 16)}> (als_row_solve; U-step and V-step launches"
                          + ("; short rows of the U-step in k_row_dual" if k > 64 else "") + ")",
                "bound": "hbm", "achieved": by / t_rs / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": by / t_rs / 1e9 / 8000.0,
                "frac_note": "algorithmic bytes per launch / average launch duration of the dominant kernel",
                "frac_iter": by_iter / (elapsed / args.steps) / 1e9 / 8000.0,
                "frac_iter_note": "SURVEY 8(d): [2(4k+12)N + (m+n)(4k+12)] / t_iter / 8 TB/s (whole iteration)",
                "avg_launch_ms": 1e3 * t_rs / n_launch,
                "algorithmic_bytes_per_launch": by / n_launch,
                "per_launch": {nm: {"ms": sum(phase.get(nm, [0.0])) / args.steps,
                                    "algorithmic_GBps": (4 * k + 12) * (nn + rr) / (1e-3 * sum(phase.get(nm, [1e-9])) / args.steps) / 1e9,
                                    "frac": (4 * k + 12) * (nn + rr) / (1e-3 * sum(phase.get(nm, [1e-9])) / args.steps) / 1e9 / 8000.0}
                               for nm, nn, rr in (("row_solve_user", nn_u, rows_u), ("row_solve_item", nn_i, rows_i))},
                "traffic": None}
        # matrix-core view: flops the kernel actually ISSUES.  f16x2 mode: the 10 of 16 lower Gram blocks, three
        # fp16 MFMAs per block (2-way split) -> 3 * (10/16) * 2 k^2 flops per rating, against the dense fp16 / bf16
        # peak; f32 mode: (10/16) * 2 k^2 on v_mfma_f32 against the fp32 matrix peak.  (For k != 64 the block count
        # is KB(KB+1)/2 of KB^2.)
        kb = -(-k // 16)
        sym = (kb * (kb + 1) / 2) / (kb * kb)
        if args.solve_dtype == "float64":
            issued, peak_tf, what = sym * 2 * (16 * kb) ** 2 * (nn_u + nn_i), 78.6, "fp64 MFMA flops issued / dense fp64 matrix peak"
        elif args.gram == "f16x2":
            issued, peak_tf, what = 3 * sym * 2 * (16 * kb) ** 2 * (nn_u + nn_i), 2500.0, "fp16 MFMA flops issued (3 per Gram product) / dense fp16 peak"
            if getattr(eng.be, "_planes", None):
                # U-step from pre-split operands (als_row_solve_params::F_planes): F^T r / F^T 1 on the matrix cores
                # too, two more 16x16x32 instructions per column block and 32 ratings
                issued += 2 * 2 * 16 * (16 * kb) * nn_u
                what += "; U-step incl. the right-hand-side operand (pre-split operands)"
        else:
            issued, peak_tf, what = sym * 2 * (16 * kb) ** 2 * (nn_u + nn_i), 157.3, "fp32 MFMA flops issued / dense fp32 matrix peak"
        roof["mfma_view"] = {"issued_TFLOPs": issued / t_rs / 1e12, "peak_TFLOPs": peak_tf,
                             "frac": issued / t_rs / 1e12 / peak_tf, "what": what,
                             "algorithmic_full_gram_fp32_TFLOPs": fl / t_rs / 1e12}
        # HBM bytes per launch: NOT measured in this run - taken from the committed PMC pass of the same workload
        # (profiles/collect_pmc.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
        import glob
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_k_row_tasks.json")))
        if pm and args.size == "cfg4" and world == 1 and args.solve_dtype != "float64":
            roof["traffic"] = json.load(open(pm[-1]))["traffic_bytes_per_launch_mean"]
            roof["traffic_source"] = "from_profile: " + os.path.relpath(pm[-1], ROOT) + " (not measured in this run)"
        out = {
            "metric": "ratings/sec per ALS iteration at k=64", "value": nnz / (elapsed / args.steps),
            "unit": "ratings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": ("f32 storage; f64 Gram / Cholesky / substitutions (solve_dtype=float64)" if args.solve_dtype == "float64"
                      else "f32 (storage, solve, accumulate); Gram products as 2-way fp16 split (error <= 2^-23) on fp16 MFMA"
                           + ("; rows over the condition limit redone in f64 (solve_dtype=auto)" if args.solve_dtype == "auto" else "")
                      if args.gram == "f16x2" else "f32"), "data": "synthetic",
            "config": {"workload": f"{args.size}: {m} users x {n} items, {nnz} ratings, k={k}, bias + "
                                   f"graph-Laplacian (alpha=0.5, {0 if S is None else int(S[1].numel())} graph nnz)"
                                   + (" [BASELINE.json configs[3]]" if args.size == "cfg4" else ""),
                       "gram": args.gram, "solve_dtype": args.solve_dtype, "hip_graph": bool(args.hip_graph), "gs_mode": getattr(eng, "gs_mode", None), "gs_levels": (len(eng.sched.offsets) - 1)
                       if eng.use_graph else 0,
                       "parallelism": f"users/items sharded x{world}, all-gather of factor blocks",
                       "graph": args.graph if use_graph else None, "graph_build_s": graph_build_s,
                       # set-up of the PRODUCT (upload, task lists, level schedule, initial factors) and the generation
                       # of the synthetic inputs, timed separately; both are outside `value`
                       "setup_s": t_setup, "datagen_s": datagen_s},
            "phase_ms_per_step": {kk: sum(v) / args.steps for kk, v in phase.items()},
            "train_rmse": [float(x) for x in hist[:, 0]],
            "roofline": roof if not args.hip_graph else None,   # per-launch times need the eager path
        }
        if eng.use_graph and getattr(eng, "gs_mode", None) == "exact":
            # what the exact (reference-order) Gauss-Seidel sweep allows at most on N GPUs: the sweep is a dependency
            # chain that takes the same time on any number of GPUs, everything else divides by N (DESIGN.md section 6;
            # from THIS run's one-GPU phase times when N = 1, otherwise stated as unknown)
            if world == 1 and phase.get("gs_sweep"):
                t_sw = sum(phase["gs_sweep"]) / args.steps
                out["config"]["scaling_cap"] = {
                    "mode": "exact", "sweep_ms": t_sw,
                    "speedup_bound_at_8_gpus": ms_step / (t_sw + (ms_step - t_sw) / 8.0),
                    "note": "serial sweep + everything else / N, exchanges not counted; north_star asks >= 6x"}
            else:
                out["config"]["scaling_cap"] = {"mode": "exact", "note": "the sweep's dependency chain does not shrink "
                                                "with N; see the N = 1 line for the bound"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        if (world == 1 and not dist_on and args.size == "cfg4" and not args.no_secondary and not args.hip_graph
                and args.solve_dtype == "auto" and backend is None):
            # other shapes / the reference's arithmetic type, in the driver's record as well (each a few seconds):
            # same code path, fewer steps; ms_per_step and the SURVEY 8(d) iteration fraction
            del res, eng
            sec = {}
            r64 = run_case("cfg4", inputs, dev, steps=3, warmup=1, solve_dtype="float64")
            sec["cfg4_float64"] = {"ms_per_step": 1e3 * r64["elapsed"] / 3, "frac_iter": frac_iter_of("cfg4", nnz, r64["elapsed"] / 3),
                                   "setup_s": r64["setup_s"], "what": "headline workload with solve_dtype=float64 everywhere"}
            del r64, inputs, csr, csc, S
            torch.cuda.empty_cache()
            for sz, what in (("cfg5-small", "BASELINE configs[4] shape / 50: k = 128, bias + W_f + Laplacian, popularity-scaled lambda_v"),
                             ("cfg3", "BASELINE configs[2]: 138K x 27K, 20M ratings, k = 64, + W_f genres / years")):
                inp = make_inputs(sz, dev, 0, False)
                rr = run_case(sz, inp, dev, steps=5, warmup=2)
                nz = int(inp[0][1].numel())
                sec[sz] = {"ms_per_step": 1e3 * rr["elapsed"] / 5, "frac_iter": frac_iter_of(sz, nz, rr["elapsed"] / 5),
                           "ratings_per_s": nz / (rr["elapsed"] / 5), "setup_s": rr["setup_s"], "datagen_s": inp[5],
                           "phase_ms_per_step": {kk: sum(v) / 5 for kk, v in rr["phase"].items()}, "what": what}
                del inp, rr
                torch.cuda.empty_cache()
            out["secondary"] = sec
        print(json.dumps(out))
    else:
        out = None
    if own_group:
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
