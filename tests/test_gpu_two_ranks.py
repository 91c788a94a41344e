"""T5' (GPU): the sharded code path with the REAL kernels - two ranks sharing cuda:0, collectives over
gloo (RCCL refuses two ranks on one device; the 8-GPU node is the driver's).  Every rank must end with the
same factors, and those must match the one-rank HIP fit and the reference fixture."""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
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
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests.common import Golden
        from tests.test_gpu_parity import _model_for
        g = Golden(name)
        r, c, v = g.train
        model = _model_for(g, device="cuda:0", gs_mode=gs_mode, process_group="world")
        model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"],
                      min_iters=g.cfg["min_iters"], verbose=0)
        assert model._eng.world == world and model._eng.multi
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), U=model.U, V=model.V, b_u=model.b_u, b_i=model.b_i,
                 mu=model.mu, rmse=np.asarray(model.history["train_rmse"]),
                 **{"W_" + f: model.W[f] for f in g.cfg["feats"]})
    finally:
        dist.destroy_process_group()


def _run(name, gs_mode=None, world=2):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), name, gs_mode, d), nprocs=world, join=True)
        return [dict(np.load(os.path.join(d, f"rank{r}.npz"))) for r in range(world)]


def _single(name):
    from tests.common import Golden
    from tests.test_gpu_parity import _model_for
    g = Golden(name)
    r, c, v = g.train
    model = _model_for(g, device="cuda:0")
    model.fit_coo(r, c, v, (g.m, g.n), features=g.features or None, tol=g.cfg["tol"],
                  min_iters=g.cfg["min_iters"], verbose=0)
    return g, model


@pytest.mark.parametrize("name", ["g2_bias_pop", "g4_feat_uw2", "g5_graph_a0.5"])
def test_two_ranks_on_one_gpu_match_one_rank(name):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no ROCm device is visible")
    g, ref = _single(name)
    outs = _run(name)
    for key in ("U", "V", "b_u", "b_i", "rmse"):
        np.testing.assert_array_equal(outs[0][key], outs[1][key], err_msg=f"ranks disagree on {key}")
    if name == "g2_bias_pop":           # no sweep, no W-step: sharding leaves every row's arithmetic alone
        np.testing.assert_array_equal(outs[0]["U"], ref.U)
        np.testing.assert_array_equal(outs[0]["V"], ref.V)
    scale = max(np.abs(ref.V).max(), np.abs(ref.U).max())
    np.testing.assert_allclose(outs[0]["U"], ref.U, rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(outs[0]["V"], ref.V, rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(outs[0]["rmse"], ref.history["train_rmse"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[0]["rmse"], g.d["hist_train_rmse"], rtol=0, atol=2e-5)   # and the reference
    for f in g.cfg["feats"]:
        np.testing.assert_allclose(outs[0]["W_" + f], ref.W[f], rtol=0, atol=2e-5 * max(np.abs(ref.W[f]).max(), 1e-3))


def test_block_mode_is_an_approximation_that_both_ranks_agree_on():
    outs = _run("g5_graph_a0.5", gs_mode="block")
    g, ref = _single("g5_graph_a0.5")
    np.testing.assert_array_equal(outs[0]["V"], outs[1]["V"])
    d = np.max(np.abs(outs[0]["rmse"] - np.asarray(ref.history["train_rmse"])))
    assert 0 < d < 5e-3
