"""T5 (CPU, gloo, world_size 2): the N>1 path of the engine - user/item sharding,
in-place all-gathers of the factor blocks, all-reduce of the statistics and of the
W-step normal equations, block / exact Gauss-Seidel modes.  The per-row arithmetic is
supplied by the test-only numpy stand-in (tests/cpu_backend.py); what is under test is
the host logic that the 2/4/8-GPU runs execute unchanged with the HIP backend."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, gs_mode, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.common import Golden
        from tests.cpu_backend import NumpyBackend
        from tests.test_gpu_parity import _model_for
        g = Golden(name)
        r, c, v = g.train
        model = _model_for(g, device="cpu", backend=NumpyBackend(), gs_mode=gs_mode, process_group="world")
        model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"],
                      min_iters=g.cfg["min_iters"], verbose=0)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), U=model.U, V=model.V, b_u=model.b_u,
                 b_i=model.b_i, mu=model.mu, rmse=np.asarray(model.history["train_rmse"]),
                 **{"W_" + f: model.W[f] for f in g.cfg["feats"]})
    finally:
        dist.destroy_process_group()


def _run(name, gs_mode=None, world=2):
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, gs_mode, d), nprocs=world, join=True)
        return [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]


def _single(name):
    from tests.common import Golden
    from tests.cpu_backend import NumpyBackend
    from tests.test_gpu_parity import _model_for
    g = Golden(name)
    r, c, v = g.train
    model = _model_for(g, device="cpu", backend=NumpyBackend())
    model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"],
                  min_iters=g.cfg["min_iters"], verbose=0)
    return g, model


@pytest.mark.parametrize("name", ["g2_bias_pop", "g3_empty", "g4_feat_uw2"])
def test_two_ranks_equal_one_rank(name):
    """Sharding must not change per-row arithmetic: factors identical on both ranks and equal to
    the single-process run (bitwise without features; the W-step all-reduce changes the summation
    order of the (d k)^2 normal equations, hence a tight tolerance there)."""
    g, ref = _single(name)
    outs = _run(name)
    for key in ("U", "V", "b_u", "b_i", "rmse"):
        np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=f"ranks disagree on {key}")
    tol = dict(rtol=0, atol=0) if not g.cfg["feats"] else dict(rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(outs[0]["U"], ref.U, **tol)
    np.testing.assert_allclose(outs[0]["V"], ref.V, **tol)
    np.testing.assert_allclose(outs[0]["rmse"], ref.history["train_rmse"], rtol=1e-9 if g.cfg["feats"] else 1e-12)
    # and the sharded run still reproduces the reference fixture
    np.testing.assert_allclose(outs[0]["rmse"], g.d["hist_train_rmse"], atol=2e-6)


@pytest.mark.parametrize("world", [2, 3])
def test_exact_gauss_seidel_across_ranks(world):
    """gs_mode='exact' (the default): the item shards sweep one after the other in rank order, each
    followed by a broadcast of its rows - the reference's Gauss-Seidel order.  Every item sees the same
    neighbour values as in the single-process sweep and sums them in the same order: bitwise equal."""
    g, ref = _single("g5_graph_a0.5")
    outs = _run("g5_graph_a0.5", world=world)
    for r in range(1, world):
        np.testing.assert_array_equal(outs[0]["V"], outs[r]["V"])
        np.testing.assert_array_equal(outs[0]["b_i"], outs[r]["b_i"])
    np.testing.assert_array_equal(outs[0]["V"], ref.V)
    np.testing.assert_array_equal(outs[0]["U"], ref.U)
    np.testing.assert_allclose(outs[0]["rmse"], ref.history["train_rmse"], rtol=1e-12)
    np.testing.assert_allclose(outs[0]["rmse"], g.d["hist_train_rmse"], atol=2e-6)


def test_level_exchange_gauss_seidel_across_ranks():
    """gs_mode='levels': global level schedule, exchange of the fresh V rows after every level."""
    g, ref = _single("g5_graph_a0.5")
    outs = _run("g5_graph_a0.5", gs_mode="levels")
    np.testing.assert_array_equal(outs[0]["V"], outs[1]["V"])
    np.testing.assert_allclose(outs[0]["V"], ref.V, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(outs[0]["rmse"], g.d["hist_train_rmse"], atol=2e-6)


def test_block_gauss_seidel_is_close_and_consistent():
    """gs_mode='block' (opt-in): Gauss-Seidel inside an item shard, previous-iteration
    values across shards.  Not the reference's order - the deviation is bounded here and stated
    in DESIGN.md; both ranks must still agree bitwise."""
    g, ref = _single("g5_graph_a0.5")
    outs = _run("g5_graph_a0.5", gs_mode="block")
    np.testing.assert_array_equal(outs[0]["V"], outs[1]["V"])
    assert np.max(np.abs(outs[0]["rmse"] - np.asarray(ref.history["train_rmse"]))) < 5e-3
    assert np.max(np.abs(outs[0]["rmse"] - np.asarray(ref.history["train_rmse"]))) > 0   # it is a different sweep


def test_three_ranks_uneven_shards():
    """world_size 3 on 300 x 200: three rating-balanced shards of different row counts."""
    g, ref = _single("g2_bias_pop")
    outs = _run("g2_bias_pop", world=3)
    for r in (1, 2):
        np.testing.assert_array_equal(outs[0]["U"], outs[r]["U"])
    np.testing.assert_array_equal(outs[0]["U"], ref.U)
    np.testing.assert_array_equal(outs[0]["V"], ref.V)


def _skewed_problem():
    """120 x 90 with heavy rows FIRST (ids sorted by descending count): equal-row shards would be badly
    unbalanced, rating-balanced shards have very different row counts (some ranks own a handful of rows)."""
    from collaborative_filtering_amd import layout
    from tests.synth import make_features, make_ratings
    m, n = 120, 90
    r, c, v = make_ratings(m, n, 2600, seed=77, user_exp=1.5, item_exp=1.3)
    ur = np.argsort(np.argsort(-np.bincount(r, minlength=m), kind="stable"), kind="stable")
    ir = np.argsort(np.argsort(-np.bincount(c, minlength=n), kind="stable"), kind="stable")
    r, c = ur[r], ir[c]
    o = np.lexsort((c, r))
    G, _ = make_features(n, 5)
    S = layout.dense_graph_to_csr(layout.build_similarity_dense(G, 6, 1e-8))
    return (r[o], c[o], v[o]), (m, n), (S[0], S[1].astype(np.int64), S[2])


def _skewed_worker(rank, world, port, graph, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from collaborative_filtering_amd import ALS, ALSConfig, BiasesConfig, CoreConfig, GraphConfig, GraphSimConfig
        from tests.cpu_backend import NumpyBackend
        (r, c, v), shape, S = _skewed_problem()
        cfg = ALSConfig(core=CoreConfig(n_factors=6, n_iters=4, lambda_u=2.0, lambda_v=3.0, pop_reg_mode="inverse_sqrt"),
                        biases=BiasesConfig(1.5, 2.5),
                        graph=GraphConfig(alpha=0.7, sim=GraphSimConfig(source="precomputed")) if graph else GraphConfig())
        model = ALS(cfg, device="cpu", backend=NumpyBackend(), process_group="world" if world > 1 else None)
        model.fit_coo(r, c, v, shape, tol=None, verbose=0, S=S if graph else None)
        eng = model._eng
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), U=model.U, V=model.V, b_u=model.b_u, b_i=model.b_i,
                 mu=model.mu, rmse=np.asarray(model.history["train_rmse"]),
                 ubounds=np.asarray(eng.ubounds), ibounds=np.asarray(eng.ibounds))
    finally:
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("graph", [False, True])
def test_four_ranks_rating_balanced_shards_on_a_skewed_input(graph):
    """world_size 4, shards balanced by number of ratings on an input whose heavy rows are
    contiguous: row counts per shard differ by an order of magnitude, the all-gathers are uneven, and with the
    Laplacian the exact sweep goes shard by shard.  Still bitwise the one-rank fit."""
    with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d4:
        _skewed_worker(0, 1, 0, graph, d1)
        ref = np.load(os.path.join(d1, "rank0.npz"))
        mp.spawn(_skewed_worker, args=(4, _free_port(), graph, d4), nprocs=4, join=True)
        outs = [np.load(os.path.join(d4, f"rank{r}.npz")) for r in range(4)]
    ub = outs[0]["ubounds"]
    rows = ub[:, 1] - ub[:, 0]
    assert rows.sum() == 120 and rows.max() >= 4 * max(rows.min(), 1), rows        # genuinely uneven shards
    (r, c, v), _, _ = _skewed_problem()
    per = np.array([np.sum((r >= b) & (r < e)) for b, e in ub])
    assert per.max() <= 1.5 * per.mean() + np.bincount(r).max(), per             # balanced up to one heavy row
    for o in outs:
        for key in ("U", "V", "b_u", "b_i", "rmse"):
            np.testing.assert_array_equal(o[key], ref[key], err_msg=key)
        assert float(o["mu"]) == float(ref["mu"])


def test_fused_statistics_closed_form_on_cpu():
    """Graph on, no features: the engine takes mu / RMSE from the per-item closed form (stand-in
    implements the same formula as the sweep kernel) - must equal the standalone residual pass."""
    from tests.common import Golden
    from tests.cpu_backend import NumpyBackend
    from tests.test_gpu_parity import _model_for

    NoFuse = type("NoFuse", (object,), {k: v for k, v in NumpyBackend.__dict__.items() if k != "sum_pairs"})

    g = Golden("g5_graph_a0.5")
    r, c, v = g.train
    a = _model_for(g, device="cpu", backend=NumpyBackend()).fit_coo(r, c, v, (g.m, g.n), tol=None, verbose=0, S=g.S_csr())
    b = _model_for(g, device="cpu", backend=NoFuse()).fit_coo(r, c, v, (g.m, g.n), tol=None, verbose=0, S=g.S_csr())
    assert a._eng.fused_stats and not b._eng.fused_stats
    np.testing.assert_allclose(a.history["train_rmse"], b.history["train_rmse"], rtol=1e-6)
    np.testing.assert_allclose(a.V, b.V, rtol=1e-5, atol=1e-7)


def _forced_worker(rank, world, port, name, gs_mode, outdir):
    os.environ["ALS_FORCE_COLLECTIVES"] = "1"
    _worker(rank, world, port, name, gs_mode, outdir)


@pytest.mark.parametrize("gs_mode", ["block", "exact", "levels"])
def test_forced_collectives_on_one_rank(gs_mode):
    """ALS_FORCE_COLLECTIVES=1 takes the sharded code path (chunked U-step with async all-gathers,
    per-level exchange in exact mode) on a one-rank group - the rehearsal mode `bench.py` uses to
    exercise the RCCL calls on a single GPU.  Results must equal the plain single-process run."""
    g, ref = _single("g5_graph_a0.5")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_forced_worker, args=(1, _free_port(), "g5_graph_a0.5", gs_mode, d), nprocs=1, join=True)
        out = np.load(os.path.join(d, "rank0.npz"))
    np.testing.assert_array_equal(out["U"], ref.U)
    np.testing.assert_array_equal(out["V"], ref.V)
    np.testing.assert_array_equal(out["rmse"], ref.history["train_rmse"])


def test_four_rank_shards_are_balanced_by_predicted_cost():
    """Shards are cut by ratings + c(k) * rows (layout.shard_bounds_nnz with row_cost): on the skewed input the
    predicted per-rank cost stays within 10 % of the mean (up to the granularity of one row), while shards balanced
    by ratings alone would be far off."""
    from collaborative_filtering_amd import layout
    rng = np.random.default_rng(3)
    lens = np.sort((rng.pareto(1.2, 50_000) * 20).astype(np.int64))[::-1]          # heavy rows first
    ptr = np.concatenate([[0], np.cumsum(lens)])
    c = layout.row_cost_weight(64)
    for world in (2, 4, 8):
        bounds, chunks = layout.shard_bounds_nnz(ptr, world, 2, c)
        assert bounds[0][0] == 0 and bounds[-1][1] == lens.size
        assert all(bounds[r][1] == bounds[r + 1][0] for r in range(world - 1))
        cost = layout.shard_costs(ptr, bounds, c)
        assert cost.max() <= 1.10 * cost.mean() + lens.max() + c, (world, cost)
        for (b, e), ch in zip(bounds, chunks):                                      # sub-ranges tile the shard
            assert ch[0][0] == b and ch[-1][1] == e and ch[0][1] == ch[1][0]
            cc = layout.shard_costs(ptr, ch, c)
            assert cc.max() <= 1.10 * cc.mean() + lens.max() + c
        by_ratings, _ = layout.shard_bounds_nnz(ptr, world)
        assert layout.shard_costs(ptr, by_ratings, c).max() > 1.4 * cost.mean()      # what the weight repairs
    (r, c2, v), _, _ = _skewed_problem()
    ptr = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=120))])
    b4, _ = layout.shard_bounds_nnz(ptr, 4, 1, layout.row_cost_weight(6))
    cost = layout.shard_costs(ptr, b4, layout.row_cost_weight(6))
    assert cost.max() <= 1.10 * cost.mean() + np.diff(ptr).max() + layout.row_cost_weight(6)


def _status_worker(rank, world, port, outdir):
    """Rank 1's sweep raises the error word once (as a timed-out dependency wait would); the status words are
    reduced over the group, so BOTH ranks must see SweepNotResident in the same iteration and refit together."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import collaborative_filtering_amd.als as A
        from tests.common import Golden
        from tests.cpu_backend import NumpyBackend
        from tests.test_gpu_parity import _model_for

        class FlakyDataflow(NumpyBackend):
            """Stand-in with a `gs_dataflow` entry (level by level underneath)."""
            fired = 0

            def gs_dataflow(self, *, items, S_idx_wait, publish, err, nondep=None, **kw):
                sched = self.engine().sched
                for lv in range(len(sched.offsets) - 1):
                    self.gs_level(items=items[sched.offsets[lv]:sched.offsets[lv + 1]], **kw)
                if rank == 1 and FlakyDataflow.fired == 0:
                    err.fill_(1)
                    FlakyDataflow.fired = 1

        g = Golden("g5_graph_a0.5")
        r, c, v = g.train
        be = FlakyDataflow()
        model = _model_for(g, device="cpu", backend=be, process_group="world")
        be.engine = lambda: model._eng
        raised = []
        orig_init = A._Engine.__init__

        def counting_init(self, *a, **kw):
            raised.append(1)
            orig_init(self, *a, **kw)
        A._Engine.__init__ = counting_init
        model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"], min_iters=g.cfg["min_iters"],
                      verbose=0)
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), V=model.V, U=model.U, engines=len(raised),
                 dataflow=int(model._eng.gs_dataflow), rmse=np.asarray(model.history["train_rmse"]))
    finally:
        dist.destroy_process_group()


def test_sweep_error_on_one_rank_makes_every_rank_refit():
    g, ref = _single("g5_graph_a0.5")
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_status_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        outs = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(2)]
    for o in outs:
        assert int(o["engines"]) == 2 and int(o["dataflow"]) == 0        # both ranks built a second engine
        np.testing.assert_array_equal(o["V"], ref.V)
        np.testing.assert_array_equal(o["U"], ref.U)
        assert len(o["rmse"]) == len(ref.history["train_rmse"])            # the failed run left no history behind


def _bench_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import json

    import bench
    from tests.cpu_backend import NumpyBackend
    out = bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--size", "tiny", "--no-cpu-baseline",
                      "--dist-backend", "gloo"], backend=NumpyBackend(), device="cpu")
    if rank == 0:
        with open(os.path.join(outdir, f"bench_w{world}.json"), "w") as fh:
            json.dump(out, fh)
    else:
        assert out is None


def test_bench_torchrun_path_end_to_end_on_four_cpu_ranks():
    """`bench.py`'s own code path as the driver launches it for N > 1 - rank 0 generates and broadcasts, sharded
    engine with the two-chunk U-step and its overlapped all-gathers, exact shard-ordered sweep, barrier + MAX over
    ranks, one JSON line on rank 0 - at `tiny` size on 4 CPU ranks over gloo (numpy stand-in for the kernels): an
    argument or shape bug surfaces here, not on the 8-GPU node.  The history equals the one-rank run bitwise."""
    import json
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_bench_worker, args=(4, _free_port(), d), nprocs=4, join=True)
        mp.spawn(_bench_worker, args=(1, _free_port(), d), nprocs=1, join=True)
        o4 = json.load(open(os.path.join(d, "bench_w4.json")))
        o1 = json.load(open(os.path.join(d, "bench_w1.json")))
    assert o4["n_gpus"] == 4 and o4["steps"] == 2 and o4["value"] > 0 and o4["scaling"] == "strong"
    assert o4["config"]["gs_mode"] == "exact" and o4["config"]["gs_levels"] > 0 and "scaling_cap" in o4["config"]
    assert o4["metric"].startswith("ratings/sec") and o4["unit"] == "ratings/s"
    assert np.all(np.isfinite(o4["train_rmse"])) and o4["train_rmse"] == o1["train_rmse"]
