"""`ALS` - the reference's fit / predict surface on MI355X.

Mirror of `scripts/als.py` (class ALS, :104-574) of
zhukovanadezhda/collaborative-filtering: same constructor, `fit`, `predict`,
attributes (`U V W b_u b_i mu S history`) and error behaviour, so it stands in
behind `scripts/evaluate_models.py:246-254` and `scripts/tune_params.py:376-391`.
The arithmetic is not numpy: ratings live as CSR + CSC in HBM and every
per-row / per-rating step runs in the HIP kernels behind include/als_hip.h.

Build-only additions (keyword arguments with defaults, so the reference
signature is unchanged):
  ALS(..., device=, backend=, gs_mode=, process_group=, gram=, solve_dtype=, graph_build=, hip_graph=)
  fit(..., S=)              precomputed similarity graph as CSR (ptr, idx, val)
  fit_coo(rows, cols, vals, shape, ...)   sparse-native entry for large inputs
  predict_at(flat_idx, ...) predictions at flat indices u*n+i without the
                            dense m x n matrix
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

from . import layout
from .als_config import ALSConfig

SCALE_FACTOR = 0.1      # scripts/als.py:93
EPS = 1e-10             # scripts/als.py:94
W_F64_BELOW = 1e-2      # solve_dtype="auto": fp64 V-step by-products when some lambda_w (+ 1e-10) is below this

logger = logging.getLogger(__name__)

# The similarity graph depends only on (feature matrix, topk, eps); CV harnesses refit with the same
# features fold after fold (scripts/evaluate_models.py:242), so the O(n^2 d) host build is memoised by
# content.  Two entries: a dense n x n float32 S is 99 MB at n = 4980.
_SIM_CACHE: "dict" = {}
_SIM_CACHE_MAX = 2

# Devices on which the persistent one-launch sweep has given up once (SweepNotResident: something else holds compute
# units there): later fits in this process start with the per-level launches instead of paying a failed attempt
# each - a sweep driver makes a new ALS per fit.
_DATAFLOW_GAVE_UP: "set" = set()


def _similarity_cached(X: np.ndarray, topk, eps):
    import hashlib
    Xc = np.ascontiguousarray(X)
    key = (hashlib.blake2b(Xc.view(np.uint8).reshape(-1), digest_size=16).hexdigest(), str(Xc.dtype), Xc.shape,
           topk, float(eps))
    hit = _SIM_CACHE.pop(key, None)
    if hit is None:
        Sd = layout.build_similarity_dense(X, topk, eps)
        ptr, idx, val = layout.dense_graph_to_csr(Sd)
        # D = S.sum(axis=1) exactly as the reference forms it (scripts/als.py:357)
        hit = (Sd, (ptr, idx, val.astype(np.float32), Sd.sum(axis=1).astype(np.float32)))
    _SIM_CACHE[key] = hit                      # most recently used last
    while len(_SIM_CACHE) > _SIM_CACHE_MAX:
        _SIM_CACHE.pop(next(iter(_SIM_CACHE)))
    return hit


@dataclass
class _SideDev:
    nrows: int
    ncols: int
    indptr: torch.Tensor
    indices: torch.Tensor
    vals: torch.Tensor


@dataclass
class _TasksDev:
    tasks: torch.Tensor
    long_rows: torch.Tensor
    ntasks: int
    nlong: int
    nslots: int
    nnz: int
    ndual: int = 0
    nmid: int = 0


def _side_to_dev(s, device) -> _SideDev:
    if isinstance(s, _SideDev):
        return _SideDev(s.nrows, s.ncols, s.indptr.to(device), s.indices.to(device), s.vals.to(device))
    return _SideDev(s.nrows, s.ncols,
                    torch.from_numpy(s.indptr).to(device),
                    torch.from_numpy(s.indices).to(device),
                    torch.from_numpy(s.vals).to(device))


def _check_csr(indptr: torch.Tensor, indices: torch.Tensor, nrows: int, ncols: int, what: str):
    """Structural validation of a caller-supplied CSR (one device reduction each): an index outside
    [0, ncols) would be an out-of-bounds gather inside the kernels."""
    if indptr.numel() != nrows + 1:
        raise ValueError(f"{what}: indptr has {indptr.numel()} entries, expected {nrows + 1}")
    nnz = indices.numel()
    if int(indptr[0]) != 0 or int(indptr[-1]) != nnz or (nrows and bool((indptr[1:] < indptr[:-1]).any())):
        raise ValueError(f"{what}: indptr must rise monotonically from 0 to nnz = {nnz}")
    if nnz and (int(indices.min()) < 0 or int(indices.max()) >= ncols):
        raise ValueError(f"{what}: index outside [0, {ncols})")


def _as_side(triple, nrows: int, ncols: int):
    indptr, indices, vals = triple
    if isinstance(indptr, torch.Tensor):
        if indptr.dtype != torch.int64 or indices.dtype != torch.int32 or vals.dtype != torch.float32:
            raise ValueError("device CSR needs int64 indptr, int32 indices, float32 vals")
        if indptr.numel() != nrows + 1 or indices.numel() != vals.numel():
            raise ValueError("inconsistent CSR sizes")
        _check_csr(indptr, indices, nrows, ncols, "ratings CSR")
        return _SideDev(nrows, ncols, indptr.contiguous(), indices.contiguous(), vals.contiguous())
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    _check_csr(torch.from_numpy(indptr), torch.from_numpy(indices), nrows, ncols, "ratings CSR")
    return layout.SparseSide(nrows, ncols, indptr, indices, np.ascontiguousarray(vals, dtype=np.float32))


def _validate_graph(ptr: torch.Tensor, idx: torch.Tensor, val: torch.Tensor, n: int):
    """A caller-supplied similarity graph must be what the reference would have built (scripts/als.py:224-240):
    indices inside [0, n), ascending inside every row, no diagonal, and SYMMETRIC in pattern and value
    (S = max(S, S^T)).  The level schedule of the Gauss-Seidel sweep relies on the symmetry: a neighbour
    j > i must sit on a later level so that it still holds its previous value when i is solved."""
    _check_csr(ptr, idx, n, n, "similarity graph S")
    if val.numel() != idx.numel():
        raise ValueError("similarity graph S: values and indices differ in length")
    if idx.numel() == 0:
        return
    rows = torch.repeat_interleave(torch.arange(n, device=ptr.device), ptr[1:] - ptr[:-1])
    cols = idx.to(torch.int64)
    key = rows * n + cols
    if bool((key[1:] <= key[:-1]).any()):
        raise ValueError("similarity graph S: column indices must be strictly ascending inside every row")
    if bool((rows == cols).any()):
        raise ValueError("similarity graph S: diagonal entries are not allowed (the reference zeroes them)")
    tkey, order = torch.sort(cols * n + rows)
    if not torch.equal(tkey, key) or not torch.equal(val[order], val):
        raise ValueError("similarity graph S must be symmetric in pattern and value (S == S^T); "
                         "symmetrise with max(S, S^T) as the reference does")


def _on(device):
    """Make `device` the current HIP device for the enclosed calls: the C-ABI library launches on the current
    device (and sizes its persistent grids from that device's occupancy), torch only hands it a stream."""
    import contextlib
    return torch.cuda.device(device) if device.type == "cuda" else contextlib.nullcontext()


class _RowShift:
    """A by-product array that exists for the rank's own rows [row0, row0 + rows) only, addressed by the kernels
    with ABSOLUTE row ids: data_ptr() is moved back by row0 rows (never dereferenced outside the local rows; the
    C ABI takes plain pointers).  Everything else is the underlying tensor's."""

    def __init__(self, t: torch.Tensor, row0: int, row_elems: int):
        self.t, self.row0, self.row_elems = t, int(row0), int(row_elems)
        self.base = t                   # the tensor that exists (an attribute no torch.Tensor has)

    def data_ptr(self) -> int:
        return self.t.data_ptr() - self.row0 * self.row_elems * self.t.element_size()

    def __getattr__(self, name):
        return getattr(self.t, name)


def _to_dev(a, device, dtype) -> torch.Tensor:
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


def _tasks_to_dev(t: layout.RowTasks, device) -> _TasksDev:
    return _TasksDev(torch.from_numpy(t.tasks).to(device), torch.from_numpy(t.long_rows).to(device),
                     int(t.tasks.shape[0]), int(t.long_rows.shape[0]), t.nslots, t.nnz, t.ndual, t.nmid)


class FitCache:
    """Set-up products shared by many fits on the SAME resident inputs (sweep.SweepDriver: the tuner's
    150 x 3 fits, scripts/tune_params.py:341-421): ratings already in HBM, host copies of the row pointers, task
    lists, the similarity graph and its level schedule, uploaded features, and the (seed, shape, k)-determined
    initial factors.  Keys carry everything a value depends on; objects keyed by identity are pinned so that
    their id cannot be recycled."""

    def __init__(self):
        self._d = {}
        self._pins = []
        self.hits = 0
        self.misses = 0

    def pin(self, obj):
        self._pins.append(obj)
        return id(obj)

    def has(self, key) -> bool:
        return key in self._d

    def get(self, key, build):
        if key in self._d:
            self.hits += 1
            return self._d[key]
        self.misses += 1
        v = self._d[key] = build()
        return v


class _NoCache:
    def pin(self, obj):
        return id(obj)

    def has(self, key) -> bool:
        return False

    def get(self, key, build):
        return build()


class SweepNotResident(RuntimeError):
    """The persistent one-launch form of the Laplacian sweep (als_gs_sweep_dataflow) found some of its waves not
    running - another kernel or process held compute units - and gave up; `ALS.fit` then refits with the
    per-level launches, which have no residency requirement."""


def _host_features(features):
    """The features dict with device tensors (e.g. the output of features.normalize_feature_device) brought to the
    host: the fit keeps float32 / float64 copies of its own in HBM and validates on the host, as the reference."""
    if not features:
        return {}
    return {name: (X.detach().cpu().numpy() if torch.is_tensor(X) else X) for name, X in features.items()}


class ALS:
    """Alternating least squares with biases, feature projections and a graph
    Laplacian:  R ~ U (V + sum_f X_f W_f)^T + mu + b_u + b_i."""

    def __init__(self, config: ALSConfig, lambda_w: Optional[Dict[str, float]] = None, *,
                 device=None, backend=None, gs_mode: Optional[str] = None, process_group=None,
                 gram: Optional[str] = None, graph_build: str = "host", hip_graph: bool = False,
                 solve_dtype: str = "auto", fit_cache: Optional["FitCache"] = None) -> None:
        if config is None:                                   # scripts/als.py:146-147
            raise ValueError("ALSConfig must be provided.")
        self.cfg = config
        self.W: Dict[str, np.ndarray] = {}
        self.lambda_w: Dict[str, float] = dict(lambda_w or {})
        core = config.core
        self.n_factors = core.n_factors
        self.n_iters = core.n_iters
        self.lambda_u = core.lambda_u
        self.lambda_v = core.lambda_v
        self.random_state = core.random_state
        self.update_w_every = core.update_w_every
        self.pop_reg_mode = core.pop_reg_mode
        # reference `or` rule: None and 0.0 both fall back (scripts/als.py:166-167)
        self.lambda_bu = config.biases.lambda_bu or self.lambda_u
        self.lambda_bi = config.biases.lambda_bi or self.lambda_v
        self.alpha = config.graph.alpha
        sim = config.graph.sim
        self.S_topk = sim.topk if sim is not None else None
        self.S_eps = sim.eps if sim is not None else EPS
        self.U = self.V = self.b_u = self.b_i = None
        self.mu: float = 0.0
        self.S = None
        self.history: Dict[str, list] = {"train_rmse": [], "U_norm": [], "V_norm": [],
                                         "bu_norm": [], "bi_norm": []}
        # build-only state
        self._device = torch.device(device) if device is not None else None
        self._backend = backend
        self._gs_mode = gs_mode
        self._pg = process_group
        self._gram = gram               # "f16x2" (default) or "f32": how K1 forms the Gram on the matrix cores
        # "float64": every row's normal equations are accumulated, factorised and solved in fp64 (the reference's
        # arithmetic type) - for the small-lambda corner of the tuner's search space, where rank-deficient rows
        # have cond ~ 1/lambda; factors stay fp32 in HBM.  Several times slower than fp32.
        if solve_dtype not in ("auto", "float32", "float64"):
            raise ValueError("solve_dtype must be 'auto', 'float32' or 'float64'")
        self._solve_dtype = solve_dtype
        if graph_build not in ("host", "device"):
            raise ValueError("graph_build must be 'host' (reference-identical, dense n x n) or 'device'")
        self._graph_build = graph_build
        self._hip_graph = bool(hip_graph)          # replay iterations as captured HIP graphs (one rank only)
        self._validate_S = False
        self._fit_cache = fit_cache      # set-up products shared across fits on the same resident inputs
        self._dataflow_sweep = os.environ.get("ALS_GS_DATAFLOW", "1") != "0"     # persistent one-launch sweep
        self._eng: Optional[_Engine] = None

    # ------------------------------------------------------------------ fit
    def fit(self, R: np.ndarray, features: Optional[Dict[str, np.ndarray]] = None,
            tol: Optional[float] = 1e-3, min_iters: int = 5, verbose: int = 1, *, S=None) -> "ALS":
        """Fit on a dense (m, n) matrix with NaN for missing (scripts/als.py:300-529)."""
        R = np.asarray(R)
        ru, ri, rv = layout.dense_to_coo(R)
        return self.fit_coo(ru, ri, rv, R.shape, features=features, tol=tol, min_iters=min_iters,
                            verbose=verbose, S=S)

    def fit_coo(self, rows, cols, vals, shape, features: Optional[Dict[str, np.ndarray]] = None,
                tol: Optional[float] = 1e-3, min_iters: int = 5, verbose: int = 1, *, S=None) -> "ALS":
        """Same as `fit` on COO triplets; nothing dense m x n is ever formed."""
        shape = (int(shape[0]), int(shape[1]))
        if self._backend is None:           # HIP backend: the set-up passes of the C-ABI library (host code)
            from . import _hip
            csr, csc = layout.coo_to_sides_native(_hip.load(), rows, cols, vals, shape)
        else:
            csr, csc = layout.coo_to_sides(rows, cols, vals, shape)
        return self._fit_sides(csr, csc, features, tol, min_iters, verbose, S)

    def fit_csr(self, csr, csc, shape, features: Optional[Dict[str, np.ndarray]] = None,
                tol: Optional[float] = 1e-3, min_iters: int = 5, verbose: int = 1, *, S=None) -> "ALS":
        """Fit on ratings that are already CSR (by user) + CSC (by item).

        `csr` / `csc` are (indptr int64, indices int32, vals float32) triples of
        numpy arrays or torch tensors (device tensors are used in place: this is
        the entry for inputs that are generated or loaded straight into HBM).
        Indices must be ascending inside every row / column.
        """
        m, n = int(shape[0]), int(shape[1])
        return self._fit_sides(_as_side(csr, m, n), _as_side(csc, n, m), features, tol, min_iters,
                               verbose, S)

    def prepare_csr(self, csr, csc, shape, features: Optional[Dict[str, np.ndarray]] = None, *, S=None):
        """Upload / lay out everything `fit_csr` would and return the engine without
        iterating (bench.py times `engine.iteration(it, n_iters)` itself)."""
        m, n = int(shape[0]), int(shape[1])
        self._fit_sides(_as_side(csr, m, n), _as_side(csc, n, m), features, None, 0, 0, S, run=False)
        return self._eng

    def _fit_sides(self, csr, csc, features, tol, min_iters, verbose, S, run: bool = True,
                   S_trusted: bool = False) -> "ALS":
        m, n = csr.nrows, csc.nrows
        features = _host_features(features)
        for name, X in features.items():                     # scripts/als.py:346-351
            if X.shape[0] != n:
                raise ValueError(f"Feature '{name}' has {X.shape[0]} rows; "
                                 f"expected {n} (number of items).")
            if not np.isfinite(X).all():
                raise ValueError(f"Feature '{name}' contains infinite values.")
        if self.pop_reg_mode and self.pop_reg_mode != "inverse_sqrt":   # scripts/als.py:259
            raise ValueError(f"Unknown pop_reg_mode '{self.pop_reg_mode}'")

        # graph (scripts/als.py:354-357): on iff alpha > 0, sim configured, graph available
        S_csr = None
        self.S = None
        if (self.alpha > 0.0) and (self.cfg.graph.sim is not None):
            if S is not None:
                S_csr = tuple(S)
                self.S = S_csr
                self._validate_S = not S_trusted      # (the sweep driver passes graphs this package built)
            else:
                X = features.get(self.cfg.graph.sim.feature_name)
                if X is None:                                # scripts/als.py:219-222
                    logger.warning("GraphSim feature '%s' not found in features dict. "
                                   "Graph regularization disabled.", self.cfg.graph.sim.feature_name)
                elif self._graph_build == "device":
                    dev = self._device or torch.device("cuda", torch.cuda.current_device())
                    with _on(dev):
                        from . import _hip
                        S_csr = layout.build_similarity_device(X, self.S_topk, self.S_eps, dev,
                                                               lib=_hip.load() if self._backend is None else None)
                    self.S = S_csr[:3]
                else:
                    self.S, S_csr = _similarity_cached(X, self.S_topk, self.S_eps)

        device = self._device or torch.device("cuda", torch.cuda.current_device()
                                              if torch.cuda.is_available() else 0)
        backend = self._backend
        if backend is None:
            from .backend import HipBackend
            backend = HipBackend(device, gram=self._gram or "f16x2", solve_dtype=self._solve_dtype)
        with _on(device):
            self._eng = _Engine(self, csr, csc, features, S_csr, device, backend, self._pg, self._gs_mode)
        if not run:                         # prepare(): the caller drives the iterations
            return self
        if verbose > 0:
            logger.info("Starting ALS training: n_factors=%d, n_iters=%d, lambda_u=%s, lambda_v=%s, "
                        "pop_reg_mode=%s, features=%s, lambda_w=%s, random_state=%s, graph_alpha=%s, "
                        "update_w_every=%s, world=%d", self.n_factors, self.n_iters, self.lambda_u,
                        self.lambda_v, self.pop_reg_mode, list(features), self.lambda_w,
                        self.random_state, self.alpha, self.update_w_every, self._eng.world)
        with _on(device):
            try:
                self._eng.run(tol, min_iters, verbose)
            except SweepNotResident as e:
                # results of the failed sweep are invalid: refit from the same initial state with one launch per
                # dependency level (stream order is the only synchronisation they need)
                logger.warning("%s; refitting with per-level sweep launches", e)
                self._dataflow_sweep = False
                _DATAFLOW_GAVE_UP.add(str(device))
                self._eng = _Engine(self, csr, csc, features, S_csr, device, backend, self._pg, self._gs_mode)
                self._eng.run(tol, min_iters, verbose)
            self._eng.export(self)
        if verbose > 0 and self.history["train_rmse"]:
            logger.info("ALS training finished. Final train RMSE: %.4f", self.history["train_rmse"][-1])
        return self

    # -------------------------------------------------------------- predict
    def _check_predict(self, features):
        features = _host_features(features)
        if self.U is None or self.V is None:                 # scripts/als.py:554-555
            raise RuntimeError("Model must be fitted before prediction.")
        n = self.V.shape[0]
        features = features or {}
        for name, X in features.items():                     # scripts/als.py:560-565
            if X.shape[0] != n:
                raise ValueError(f"Feature '{name}' has {X.shape[0]} rows. "
                                 f"Expected number of rows: {n}.")
            if not np.isfinite(X).all():
                raise ValueError(f"Feature '{name}' contains infinite values.")
        return features

    def predict(self, features: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
        """Completed matrix U Z^T + mu + b_u + b_i, (m, n) float64 (scripts/als.py:532-574)."""
        features = self._check_predict(features)
        with _on(self._eng.dev):
            return self._eng.predict_dense(features)

    def predict_at(self, flat_idx, features: Optional[Dict[str, np.ndarray]] = None) -> np.ndarray:
        """Predictions at flat indices u*n+i (what scripts/tune_params.py:165-166 reads)."""
        features = self._check_predict(features)
        with _on(self._eng.dev):
            return self._eng.predict_at(np.asarray(flat_idx, dtype=np.int64), features)


class _Engine:
    """Device state and the iteration loop of one `fit`."""

    U_CHUNKS = 2        # sub-ranges of a rank's user shard (multi-rank runs only; ALS_U_CHUNKS overrides)

    def __init__(self, model: ALS, csr, csc, features, S_csr, device, backend, pg, gs_mode):
        import weakref
        # no reference cycle with the model (which owns this engine): both then die by reference count, at a
        # deterministic point - a cyclic-GC pass destroying captured HIP graphs in the middle of ANOTHER engine's
        # stream capture is an error ("operation not permitted when stream is capturing")
        self.model = weakref.proxy(model)
        self.dev = device
        self.be = backend
        # sharding is opt-in: process_group="world" (the default group) or a ProcessGroup object; None fits on
        # this rank alone even when torch.distributed happens to be initialised
        if isinstance(pg, str):
            if pg != "world":
                raise ValueError("process_group must be None, 'world' or a torch.distributed ProcessGroup")
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("process_group='world' needs an initialised torch.distributed")
            self.pg, self.dist_on = None, True
        else:
            self.pg, self.dist_on = pg, pg is not None
        pg = self.pg
        if self.dist_on:
            self.world, self.rank = dist.get_world_size(pg), dist.get_rank(pg)
        else:
            self.world, self.rank = 1, 0
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                logger.info("torch.distributed is initialised (world %d) but no process_group was given: "
                            "fitting on this rank alone", dist.get_world_size())
        # `multi`: take the sharded code path with its collectives.  ALS_FORCE_COLLECTIVES=1 takes it on a
        # one-rank group too, to rehearse the RCCL calls on a single GPU (bench.py / tests only).
        self.multi = self.world > 1 or (os.environ.get("ALS_FORCE_COLLECTIVES") == "1" and self.dist_on)
        k = int(model.n_factors)
        self.k, self.ld = k, layout.padded_k(k)
        self.m, self.n = csr.nrows, csc.nrows
        self.nnz = int(csr.vals.shape[0])
        self.timers = None              # set to [] to record (name, start, end) events per phase
        self.perm = torch.from_numpy(layout.perm_of_col(k)).to(device)     # storage col -> perm pos
        f32, f64 = torch.float32, torch.float64

        cache = model._fit_cache if model._fit_cache is not None else _NoCache()
        ku, ki = cache.pin(csr), cache.pin(csc)

        # --- initial factors (scripts/als.py:329,360-376): numpy's Generator on the host, same seed and draw order
        # as the reference.  At 1M x 64 that is 0.8 s of Generator.normal (the reference pays the same); it runs in
        # a worker thread (numpy releases the GIL) underneath the rest of the set-up.
        self.feat_names = list(features)
        self.feat_dims = [int(features[f].shape[1]) for f in self.feat_names]
        init_key = ("init", model.random_state, self.m, self.n, k, tuple(self.feat_dims))
        # The V-step in fp64 with fp64 by-products (item Grams, right-hand sides, Cholesky factors, column sums) for
        # the W-step / the sweep / the item statistics: always under solve_dtype="float64"; under "auto" when a
        # feature's ridge parameter is (nearly) absent - lambda_w missing means 0 (scripts/als.py:497) - because the
        # (d k)^2 system is then singular up to the 1e-10 the reference adds and normal equations assembled from
        # fp32 Grams (rounding ~3e-7 |A|) are no longer positive definite, let alone accurate.
        lw_min = min((float(model.lambda_w.get(f, 0.0)) for f in self.feat_names), default=float("inf"))
        mode = getattr(backend, "solve_dtype", "float32")
        self.v_f64 = bool(getattr(backend, "lib", None) is not None and
                          (mode == "float64" or (mode == "auto" and lw_min + EPS < W_F64_BELOW)))

        def draw_init_host():
            rng = np.random.default_rng(model.random_state)
            U0 = self._padded_host(rng.normal(scale=SCALE_FACTOR, size=(self.m, k)))
            V0 = self._padded_host(rng.normal(scale=SCALE_FACTOR, size=(self.n, k)))
            W0 = [rng.normal(scale=SCALE_FACTOR, size=(d, k)) for d in self.feat_dims]
            return U0, V0, W0

        init_future = None
        if not cache.has(init_key):
            import concurrent.futures
            pool = concurrent.futures.ThreadPoolExecutor(max_workers=1)
            init_future = pool.submit(draw_init_host)
            pool.shutdown(wait=False)

        # --- ratings in HBM
        self.csr = cache.get(("side_dev", ku), lambda: _side_to_dev(csr, device))
        self.csc = cache.get(("side_dev", ki), lambda: _side_to_dev(csc, device))
        uptr_h = cache.get(("indptr_h", ku), lambda: self.csr.indptr.cpu().numpy())
        iptr_h = cache.get(("indptr_h", ki), lambda: self.csc.indptr.cpu().numpy())

        # --- shards: contiguous row ranges balanced by predicted cost = ratings + c(k) * rows (about
        # half of a U-step row's time at cfg 4 is per row, not per rating, so rating-balanced shards of a skewed input
        # would be time-imbalanced); every rank derives the same tables from the row pointers.  The user shard is solved in u_chunks sub-ranges whose all-gathers overlap
        # the next sub-range's solve.  Factor storage is exactly [rows + 1, ld] (the extra row is the zero row).
        self.u_chunks = max(1, int(os.environ.get("ALS_U_CHUNKS", self.U_CHUNKS))) if self.multi else 1
        self.row_cost = layout.row_cost_weight(k)
        self.ubounds, self.uchunks = layout.shard_bounds_nnz(uptr_h, self.world, self.u_chunks, self.row_cost)
        self.ibounds, _ = layout.shard_bounds_nnz(iptr_h, self.world, 1, self.row_cost)
        self.ub, self.ue = self.ubounds[self.rank]
        self.ib, self.ie = self.ibounds[self.rank]
        m_pad, n_pad = self.m, self.n
        self.m_pad, self.n_pad = m_pad, n_pad
        dl = layout.dual_max_len(k)          # rows this short are solved in the dual form (k_row_dual)
        dm = layout.dual_mid_len(k)          # ... and rows up to this length above k = 96 (k_row_dual_mid)
        lib = getattr(backend, "lib", None)  # HIP backend: set-up passes in the library (csrc/host_setup.cpp)

        def row_tasks(ptr, lo, hi):
            if lib is not None:
                return layout.build_row_tasks_native(lib, ptr, lo, hi, dual_len=dl, mid_len=dm)
            return layout.build_row_tasks(ptr, lo, hi, dual_len=dl, mid_len=dm)

        def tasks_dev(key_side, ptr, lo, hi):
            return cache.get(("tasks", key_side, lo, hi, dl, dm), lambda: _tasks_to_dev(row_tasks(ptr, lo, hi), device))

        self.utasks = tasks_dev(ku, uptr_h, self.ub, self.ue)
        self.itasks = tasks_dev(ki, iptr_h, self.ib, self.ie)
        if self.u_chunks > 1:
            self.utasks_c = [tasks_dev(ku, uptr_h, b, e) for b, e in self.uchunks[self.rank]]
        nslots = max(self.utasks.nslots, self.itasks.nslots)
        slot_bytes = max(backend.slot_bytes(k), backend.slot_bytes(k, True) if self.v_f64 else 0)
        self.workspace = (torch.empty(nslots * slot_bytes // 4, dtype=f32, device=device) if nslots else None)
        # Everything the host reads back per iteration lives in ONE 128-byte block: the history row (6 doubles at
        # byte 0) and the three status words (int32 at byte 64: row-solve status, sweep error, W-step status) - early
        # stopping then costs a single contiguous device-to-host copy per iteration instead of four small ones.
        self.ctrl = torch.zeros(128, dtype=torch.uint8, device=device)
        self.hist_row = self.ctrl[0:48].view(f64)
        words = self.ctrl[64:128].view(torch.int32)
        self.status, self.gs_err_word, self.w_bad = words[0:1], words[1:2], words[2:3]
        self.status_words = words[0:4]

        # --- parameters (scripts/als.py:329,360-376): numpy Generator on the host, same draw order
        mean0 = cache.get(("mean", ku), lambda: float(self.csr.vals.to(f64).mean().item()) if self.nnz
                          else float("nan"))                                                 # :360
        self.mu = torch.tensor([mean0], dtype=f64, device=device)

        def upload_init():
            U0, V0, W0 = init_future.result()
            return (torch.from_numpy(U0).to(device), torch.from_numpy(V0).to(device),
                    [torch.from_numpy(w).to(device) for w in W0])

        U0d, V0d, W0d = cache.get(init_key, upload_init)
        if model._fit_cache is not None:        # the cached initial state stays pristine
            U0d, V0d, W0d = U0d.clone(), V0d.clone(), [w.clone() for w in W0d]
        self.U, self.V = U0d[:m_pad], V0d[:n_pad]
        self.b_u = torch.zeros(m_pad, dtype=f32, device=device)
        self.b_i = torch.zeros(n_pad, dtype=f32, device=device)
        self.W64 = dict(zip(self.feat_names, W0d))
        if self.feat_names:
            def upload_features():
                Xcat = np.concatenate([np.asarray(features[f], dtype=np.float32) for f in self.feat_names], axis=1)
                Xp = np.zeros((n_pad, Xcat.shape[1]), dtype=np.float32)
                Xp[: self.n] = Xcat
                return (torch.from_numpy(Xp).to(device),
                        {f: torch.from_numpy(np.asarray(features[f], dtype=np.float64)).to(device)
                         for f in self.feat_names})
            fkey = ("features", tuple((f, cache.pin(features[f])) for f in self.feat_names), n_pad)
            self.Xcat, self.X64 = cache.get(fkey, upload_features)
            self.Wcat = torch.zeros(self.Xcat.shape[1], self.ld, dtype=f32, device=device)
            self.Z = torch.zeros(n_pad + 1, self.ld, dtype=f32, device=device)[:n_pad]
            self._sync_wcat()
        else:
            self.Xcat = self.Wcat = None
            self.Z = self.V                       # Z == V when there are no features

        # --- per-item regularisation (scripts/als.py:379-384, 243-259)
        counts = np.diff(iptr_h).astype(np.float64)
        if not model.pop_reg_mode:
            lam_v = np.full(self.n, float(model.lambda_v))
        else:
            lam_v = model.lambda_v / np.sqrt(counts + 1.0)
        lv = np.zeros(n_pad, dtype=np.float32)
        lv[: self.n] = lam_v
        self.lam_v_row = torch.from_numpy(lv).to(device)

        # --- graph
        self.use_graph = S_csr is not None
        if self.use_graph:
            kg = cache.pin(S_csr)

            def upload_graph():
                S_ptr = _to_dev(S_csr[0], device, torch.int64)
                S_idx = _to_dev(S_csr[1], device, torch.int32)
                S_val = _to_dev(S_csr[2], device, f32)
                if model._validate_S:                 # caller-supplied graph: checked once, on the device
                    if S_ptr.numel() != self.n + 1:
                        raise ValueError(f"similarity graph S has {S_ptr.numel() - 1} rows; expected {self.n}")
                    _validate_graph(S_ptr, S_idx, S_val, self.n)
                if len(S_csr) > 3:
                    D = _to_dev(S_csr[3], device, f32)
                else:                       # D = S.sum(axis=1) (scripts/als.py:357); segment sums from an fp64
                    # prefix sum: deterministic (index_add_ would use float atomics)
                    csum = torch.zeros(S_val.numel() + 1, dtype=f64, device=device)
                    csum[1:] = torch.cumsum(S_val.to(f64), 0)
                    D = (csum[S_ptr[1:]] - csum[S_ptr[:-1]]).to(f32)
                return S_ptr, S_idx, S_val, D, S_ptr.cpu().numpy(), S_idx.cpu().numpy()

            self.S_ptr, self.S_idx, self.S_val, D, ptr, idx = cache.get(("graph", kg, bool(model._validate_S)),
                                                                        upload_graph)
            self.diag_extra = torch.zeros(n_pad, dtype=f32, device=device)
            self.diag_extra[: self.n] = np.float32(model.alpha) * D
            # Sweep modes with several ranks (one rank: all the same thing):
            #   "exact"  (default) the item shards sweep ONE AFTER THE OTHER in rank order, each followed by a
            #            broadcast of its rows: shard r sees the new rows of shards < r and the old rows of
            #            shards > r - the reference's Gauss-Seidel order, at the cost of a sweep that does not
            #            scale with the number of ranks;
            #   "block"  all shards sweep at once, rows of other shards are the previous iteration's
            #            (Gauss-Seidel inside a shard, Jacobi across shards: approximate, scales);
            #   "levels" global level schedule with an exchange after every level (exact; one collective per level).
            self.gs_mode = gs_mode or "exact"
            if self.gs_mode not in ("exact", "block", "levels"):
                raise ValueError(f"unknown gs_mode '{self.gs_mode}'")
            active = counts > 0
            # persistent dataflow sweep (one launch, no level barriers) when the backend has it
            self.gs_dataflow = (hasattr(backend, "gs_dataflow") and model._dataflow_sweep and not self.v_f64
                                and str(device) not in _DATAFLOW_GAVE_UP
                                and not (self.multi and self.gs_mode == "levels"))
            lo, hi = (0, self.n) if (self.multi and self.gs_mode == "levels") else (self.ib, self.ie)

            def schedule():
                if lib is not None:
                    sched, wait = layout.build_level_schedule_native(lib, ptr, idx, active, lo, hi,
                                                                     want_wait=self.gs_dataflow)
                else:
                    sched = layout.build_level_schedule(ptr, idx, active, lo, hi)
                    wait = layout.wait_edges(ptr, idx, sched.level) if self.gs_dataflow else None
                return (sched, torch.from_numpy(sched.items).to(device),
                        torch.from_numpy(wait).to(device) if wait is not None else None)

            self.sched, self.sched_items, wait_dev = cache.get(("sched", kg, ki, lo, hi, bool(self.gs_dataflow)), schedule)
            if self.gs_dataflow:
                self.S_idx_wait = wait_dev
                self.gs_publish = torch.empty(n_pad, self.ld, dtype=f32, device=device)       # same shape as V
                self.gs_err = self.gs_err_word
                # Neighbour sums that do not depend on the sweep can be formed for all items by a parallel launch
                # before it (same sums, same order).  On graphs without hubs the in-sweep gather hides behind the
                # dependency waits and the extra launch only costs (cfg 4, sampled graph: 1.79 vs 1.46 ms); with hub
                # rows it pays (exact top-50 graph, rows of up to 3700 neighbours: 2.43 vs 2.75 ms).  Default: on
                # when the longest row of S has 1024 neighbours or more; ALS_GS_NONDEP=0 / 1 forces it.
                nd_env = os.environ.get("ALS_GS_NONDEP")
                hubs = bool(ptr.size > 1 and int(np.diff(ptr).max()) >= 1024)
                self.gs_nondep = (torch.empty(n_pad, self.ld, dtype=f32, device=device)
                                  if (nd_env == "1" or (nd_env is None and hubs)) else None)
        # By-products of the V-step exist for this rank's items only ([ib, ie): n / world rows instead of n - the
        # item Grams and Cholesky factors are n * ld^2 floats each, 65 GB at BASELINE configs[4]); the kernels index
        # them with absolute item ids through a shifted base pointer (_RowShift).  The numpy stand-in of the CPU
        # tests indexes tensors directly, so it keeps full-size arrays.
        self.local_rows = (self.ib, self.ie - self.ib) if lib is not None else (0, n_pad)
        r0, nloc = self.local_rows

        byp = f64 if self.v_f64 else f32           # dtype of the V-step by-products

        def local(*row_shape, dtype=None):
            t = torch.zeros((nloc,) + row_shape, dtype=dtype or byp, device=device)
            return _RowShift(t, r0, int(np.prod(row_shape)) if row_shape else 1) if lib is not None else t

        if self.use_graph:
            self.factor = local(self.ld * self.ld)
            self.sumr = local()
        need_byproducts = self.use_graph or bool(self.feat_names)
        self.rhs_out = local(self.ld) if need_byproducts else None
        self.colsum_out = local(self.ld) if need_byproducts else None
        self.gram = local(self.ld, self.ld) if self.feat_names else None
        # Rank-local by-products scale with the rank's ITEM COUNT (factor / gram: ld^2 per item), the shards are cut by
        # cost (ratings + c rows): when item ids correlate with popularity the tail shard holds more than n / world
        # items.  Say so instead of letting the "n / world rows" memory statement fail silently.
        elt = 8 if self.v_f64 else 4
        self.byproduct_bytes = nloc * elt * ((self.ld * self.ld if self.use_graph else 0)
                                             + (self.ld * self.ld if self.feat_names else 0)
                                             + (2 * self.ld if need_byproducts else 0))
        if self.multi and lib is not None:
            even = -(-self.n // self.world)
            logger.info("rank %d: %d of %d items, %.2f GB of V-step by-products", self.rank, nloc, self.n,
                        self.byproduct_bytes / 1e9)
            if nloc > 1.5 * even + 64:
                logger.warning("rank %d holds %d items (even share %d): its V-step by-products take %.2f GB, %.1fx the "
                               "even share - item ids correlate with popularity; shards are balanced by cost, not by "
                               "item count", self.rank, nloc, even, self.byproduct_bytes / 1e9, nloc / max(even, 1))

        # --- fused statistics (DESIGN.md "Statistics"): without features Z == V, so the residual
        #     sums of an iteration follow in closed form from what the V-step already holds
        self.fused_stats = ((not self.feat_names) and hasattr(backend, "sum_pairs")
                            and (self.ld <= 64 or not self.use_graph or getattr(self, "gs_dataflow", False) or self.v_f64))
        # with features Z != V: the same sums follow per item from the V-step's Gram / rhs / column sums
        # and the final Z (als_item_stats), again without a pass over the ratings
        self.fused_feat_stats = (bool(self.feat_names) and hasattr(backend, "item_stats")
                                 and hasattr(backend, "sum_pairs"))
        if self.fused_stats or self.fused_feat_stats:
            self.stat_rows = local(2, dtype=f32)
            if self.use_graph or self.fused_feat_stats:
                self.sumr2 = local()
            if self.use_graph:
                self.lam_eff = (self.lam_v_row + np.float32(EPS) + self.diag_extra).contiguous()
        if self.fused_feat_stats and not self.use_graph:
            self.sumr = local()
        # --- stats scratch
        self.stats = torch.zeros(2, dtype=f64, device=device)
        self.ss = torch.zeros(4, dtype=f64, device=device)
        self.hist = torch.zeros(max(model.n_iters, 1), 6, dtype=f64, device=device)
        self._graphs = {}
        self._stage = {}                # receive buffers of _allgather_rows, one per (tensor, rows per rank)
        self.graphs_captured = 0
        self.replay_ok = True           # (diagnostics: set False to force eager iterations)
        self.iters_run = 0

    # ------------------------------------------------------------- helpers
    def _padded_host(self, A64: np.ndarray) -> np.ndarray:
        """[rows + 1, ld] fp32 array whose first rows are the factor matrix: the extra last row stays zero for
        ever - als_row_solve points ratings past the end of a row at it (F_zero_row)."""
        out = np.zeros((A64.shape[0] + 1, self.ld), dtype=np.float32)
        out[: A64.shape[0], : self.k] = A64
        return out

    def _sync_wcat(self):
        off = 0
        for f, d in zip(self.feat_names, self.feat_dims):
            self.Wcat[off:off + d, : self.k] = self.W64[f].to(torch.float32)
            off += d

    def _allgather_rows(self, t: torch.Tensor, bounds, async_op: bool = False, tag=None):
        """All-gather of the contiguous, unevenly sized row shards `bounds[r] = (begin, end)` of `t`: every rank
        contributes `per` rows (the longest shard) in one all_gather_into_tensor - RCCL and gloo both need equal
        sizes -, `finish()` copies the other ranks' rows into place (`tag` keeps exchanges that are in flight
        together - the sub-ranges of the U-step - on separate receive buffers).  No staging copy of the own rows: rank r sends
        the window t[s_r : s_r + per] with s_r = min(begin_r, rows - per), which contains its shard (what else is
        in the window is ignored by the receivers).  The receive buffer is allocated once per (tensor, per).
        Returns finish (called at once unless async_op)."""
        if not self.multi:
            return None
        rows = t.shape[0]
        per = min(max(max(hi - lo for lo, hi in bounds), 1), rows)
        starts = [min(lo, rows - per) for lo, _ in bounds]
        tail = tuple(t.shape[1:])
        key = (tag, t.data_ptr(), per, tail, t.dtype)
        stage = self._stage.get(key)
        if stage is None:
            stage = self._stage[key] = torch.empty((self.world, per) + tail, dtype=t.dtype, device=t.device)
        s0 = starts[self.rank]
        work = dist.all_gather_into_tensor(stage.view(-1), t[s0:s0 + per].reshape(-1), group=self.pg,
                                           async_op=async_op)

        def finish():
            if work is not None:
                work.wait()
            for r, (lo, hi) in enumerate(bounds):
                if r != self.rank and hi > lo:
                    t[lo:hi] = stage[r, lo - starts[r]: hi - starts[r]]
        if async_op:
            return finish
        finish()
        return None

    def _global_rank(self, r: int) -> int:
        return r if self.pg is None else dist.get_global_rank(self.pg, r)

    def _allreduce(self, t: torch.Tensor):
        if self.multi:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg)

    def _read_ctrl(self):
        """(history row, status words) of the last iteration: one contiguous 128-byte device-to-host copy.
        With several ranks the status words are first reduced (MAX) over the group, so that every rank takes the
        same decision - all raise LinAlgError / SweepNotResident (and refit) together, or none does; a rank that
        left the iteration loop alone would leave the others waiting in their next collective."""
        if self.multi:
            dist.all_reduce(self.status_words, op=dist.ReduceOp.MAX, group=self.pg)
        c = self.ctrl.cpu()
        return c[0:48].view(torch.float64).numpy(), c[64:76].view(torch.int32).numpy()

    def _check_status(self, words=None):
        if words is None:
            words = self._read_ctrl()[1]
        # the sweep's error word first: it is sticky, every later sweep bails out at its first wait, and whatever
        # else went wrong afterwards (NaN factors -> "not positive definite") is a consequence
        if getattr(self, "gs_dataflow", False) and int(words[1]):
            raise SweepNotResident("Gauss-Seidel dataflow sweep: a dependency wait exceeded its bound (the persistent "
                                   "launch was not resident as a whole)")
        if int(words[2]):
            raise np.linalg.LinAlgError("W-step normal equations of a feature are not positive definite")
        bad = int(words[0])
        if bad:
            self.status.zero_()
            raise np.linalg.LinAlgError(
                f"normal equations of row {bad - 1} are not positive definite")   # scripts/helpers.py:19


    def _tick(self, name):
        """Context manager recording a (name, start, end) event pair on the current stream."""
        eng = self

        class _T:
            def __enter__(self_inner):
                if eng.timers is not None:
                    self_inner.a = torch.cuda.Event(enable_timing=True)
                    self_inner.b = torch.cuda.Event(enable_timing=True)
                    self_inner.a.record()

            def __exit__(self_inner, *exc):
                if eng.timers is not None:
                    self_inner.b.record()
                    eng.timers.append((name, self_inner.a, self_inner.b))
        return _T()

    # ---------------------------------------------------------- half steps
    def user_step(self):
        """scripts/als.py:414-433 on this rank's user shard, then all-gather.

        With several ranks the shard is solved in U_CHUNKS contiguous sub-ranges; the all-gather of a
        finished sub-range (async, on the collective stream) overlaps the solve of the next one.
        """
        md = self.model
        kw = dict(k=self.k, ld=self.ld, side=self.csr, F=self.Z, zero_row=self.n_pad, bias_self=self.b_u,
                  bias_other=self.b_i, mu=self.mu, lam=md.lambda_u, lam_row=None, lam_b=md.lambda_bu,
                  lam_b_row=None, rhs_extra=None, diag_extra=None, X_out=self.U, bias_out=self.b_u,
                  gram_out=None, factor_out=None, rhs_out=None, colsum_out=None, sumr_out=None,
                  status=self.status, workspace=self.workspace)
        if self.u_chunks == 1:
            with self._tick("row_solve_user"):
                self.be.row_solve(tasks=self.utasks, **kw)
            self._allgather_rows(self.U, self.ubounds)
            self._allgather_rows(self.b_u, self.ubounds)
            return
        pending = []
        with self._tick("row_solve_user"):
            for c in range(self.u_chunks):
                if self.utasks_c[c].ntasks or self.utasks_c[c].nlong:
                    self.be.row_solve(tasks=self.utasks_c[c], **kw)
                cb = [self.uchunks[r][c] for r in range(self.world)]        # sub-range c of every rank's shard
                pending.append(self._allgather_rows(self.U, cb, async_op=True, tag=c))
                pending.append(self._allgather_rows(self.b_u, cb, async_op=True, tag=c))
        with self._tick("allgather_user_wait"):
            for finish in pending:
                finish()

    def item_step(self, want_gram: bool):
        """scripts/als.py:436-466 on this rank's item shard, then all-gather.

        The feature part of Z is deliberately not used here (reference quirk,
        :447,:465): F = U, and the bias update uses V, not Z.
        """
        md = self.model
        common = dict(k=self.k, ld=self.ld, side=self.csc, F=self.U, zero_row=self.m_pad, bias_self=self.b_i,
                      bias_other=self.b_u, mu=self.mu, lam=0.0, lam_row=self.lam_v_row,
                      lam_b=md.lambda_bi, lam_b_row=None, rhs_extra=None,
                      gram_out=self.gram if want_gram else None, status=self.status,
                      tasks=self.itasks, workspace=self.workspace, **({"f64": True} if self.v_f64 else {}))
        if not self.use_graph:
            with self._tick("row_solve_item"):
                self.be.row_solve(diag_extra=None, X_out=self.V, bias_out=self.b_i, factor_out=None,
                                  rhs_out=self.rhs_out if want_gram else None,
                                  colsum_out=self.colsum_out if want_gram else None,
                                  sumr_out=self.sumr if self.fused_feat_stats else None,
                                  sumr2_out=self.sumr2 if self.fused_feat_stats else None,
                                  stat_out=self.stat_rows if self.fused_stats else None, **common)
        else:
            # phase A (parallel): Gram, rhs, Cholesky factor of every item of the shard
            with self._tick("row_solve_item"):
                self.be.row_solve(diag_extra=self.diag_extra, X_out=None, bias_out=None,
                                  factor_out=self.factor, rhs_out=self.rhs_out, colsum_out=self.colsum_out,
                                  sumr_out=self.sumr,
                                  sumr2_out=self.sumr2 if (self.fused_stats or self.fused_feat_stats) else None,
                                  **common)
            # phase B (sequential in levels): Gauss-Seidel sweep with live V (:458)
            if self.multi and self.gs_mode == "exact":
                # shards in rank order; the broadcast hands shard r's new rows to everybody before shard
                # r + 1 starts (stream-ordered on every rank)
                with self._tick("gs_sweep"):
                    for r in range(self.world):
                        if r == self.rank:
                            self._gs_sweep()
                        lo, hi = self.ibounds[r]
                        if hi > lo:
                            dist.broadcast(self.V[lo:hi], src=self._global_rank(r), group=self.pg)
                self._allgather_rows(self.b_i, self.ibounds)
                return
            with self._tick("gs_sweep"):
                self._gs_sweep()
        self._allgather_rows(self.V, self.ibounds)
        self._allgather_rows(self.b_i, self.ibounds)

    def _gs_sweep(self):
        md = self.model
        off = self.sched.offsets
        exact_multi = self.multi and self.gs_mode == "levels"
        kw = dict(k=self.k, ld=self.ld, S_ptr=self.S_ptr, S_idx=self.S_idx, S_val=self.S_val,
                  alpha=md.alpha, factor=self.factor, rhs=self.rhs_out, colsum=self.colsum_out,
                  sumr=self.sumr, indptr=self.csc.indptr, lam_b=md.lambda_bi, lam_b_row=None,
                  V=self.V, bias=self.b_i)
        if self.fused_stats:
            kw.update(sumr2=self.sumr2, lambda_eff=self.lam_eff, stat_out=self.stat_rows)
        if self.v_f64:
            kw.update(f64=True)
        if self.gs_dataflow:
            self.be.gs_dataflow(items=self.sched_items, S_idx_wait=self.S_idx_wait, publish=self.gs_publish,
                                err=self.gs_err, nondep=self.gs_nondep, **kw)
            return
        if not exact_multi and hasattr(self.be, "gs_levels"):
            # no collective between levels: the whole sweep is one C call (one launch per level)
            self.be.gs_levels(offsets=np.ascontiguousarray(off, dtype=np.int64), items=self.sched_items, **kw)
            return
        for lv in range(len(off) - 1):
            items = self.sched_items[off[lv]:off[lv + 1]]
            if exact_multi:
                it_np = self.sched.items[off[lv]:off[lv + 1]]
                lo = int(np.searchsorted(it_np, self.ib))
                hi = int(np.searchsorted(it_np, self.ie))
                mine = items[lo:hi]
            else:
                mine = items
            if mine.numel():
                self.be.gs_level(items=mine, **kw)
            if exact_multi:
                self._exchange_level(items, it_np)

    def _exchange_level(self, items: torch.Tensor, it_np: np.ndarray):
        """Exact multi-GPU sweep: publish this level's freshly solved V rows."""
        ends = np.array([hi for _, hi in self.ibounds], dtype=np.int64)
        owner = np.searchsorted(ends, it_np, side="right")          # rank whose [begin, end) holds the item
        cnt = np.bincount(owner, minlength=self.world)
        cmax = int(cnt.max())
        if cmax == 0:
            return
        lo = int(np.searchsorted(it_np, self.ib))
        buf = torch.zeros(cmax, self.ld, dtype=torch.float32, device=self.dev)
        n_mine = int(cnt[self.rank])
        if n_mine:
            buf[:n_mine] = self.V[items[lo:lo + n_mine].long()]
        allb = torch.empty(self.world * cmax, self.ld, dtype=torch.float32, device=self.dev)
        dist.all_gather_into_tensor(allb.view(-1), buf.view(-1), group=self.pg)
        start = 0
        for r in range(self.world):
            c = int(cnt[r])
            if c and r != self.rank:
                self.V[items[start:start + c].long()] = allb[r * cmax:r * cmax + c]
            start += c

    # --------------------------------------------------------------- W step
    def w_step(self, b_i_old: torch.Tensor):
        """scripts/als.py:468-501 without the N_obs x (d k) design matrix.

        With G_i = U_i^T U_i (the item Gram of this iteration's V-step) the
        reference's normal equations are
            A_f = sum_i (x_i x_i^T) (x) G_i + (lambda_f + 1e-10) I
            b_f = sum_i x_i (x) (g_i + G_i xw_{f,i}),   g_i = U_i^T residual_i
        and g_i = U_i^T rho_i - G_i z_i follows from the V-step by-products
        (rhs, column sums), so no extra pass over the ratings is needed.  The
        Jacobi-across-features quirk (:474-489) and the lambda=0-for-missing
        quirk (:497) are kept.  A_f / b_f come from the HIP kernels of
        w_step.hip (als_w_normal_equations) in fp64 and are solved by the blocked
        fp64 Cholesky of spd_solve.hip (als_spd_solve_f64).
        """
        md = self.model
        k, ld = self.k, self.ld
        if not hasattr(self, "H"):
            r0, nloc = self.local_rows
            H = torch.zeros(len(self.feat_names), nloc, ld, dtype=torch.float64 if self.v_f64 else torch.float32,
                            device=self.dev)
            self.H = _RowShift(H, r0, ld) if r0 or nloc != self.n_pad else H
            offs = np.concatenate([[0], np.cumsum(self.feat_dims)]).astype(np.int32)
            self.feat_off_host = offs
            self.feat_off = torch.from_numpy(offs).to(self.dev)
            self.w_status = torch.zeros(1, dtype=torch.int32, device=self.dev)    # (w_bad: sticky, in self.ctrl)
        f64kw = {"f64": True} if self.v_f64 else {}
        Wold = torch.cat([self.W64[f] for f in self.feat_names], dim=0).contiguous() if self.v_f64 else self.Wcat
        self.be.w_item_vectors(k=k, ld=ld, item_begin=self.ib, item_end=self.ie, gram=self.gram,
                               rhs=self.rhs_out, colsum=self.colsum_out, V=self.V, b_new=self.b_i,
                               b_old=b_i_old, X=self.Xcat, feat_off=self.feat_off, W=Wold, H=self.H, **f64kw)
        newW = {}
        for fi, (f, d) in enumerate(zip(self.feat_names, self.feat_dims)):
            A_full, B_full = self.be.w_accumulate(k=k, ld=ld, item_begin=self.ib, item_end=self.ie,
                                                  gram=self.gram, X=self.Xcat, H=self.H, feat_index=fi,
                                                  feat_col0=int(self.feat_off_host[fi]), feat_d=d, **f64kw)
            if self.multi:
                self._allreduce(A_full)
                self._allreduce(B_full)
            x = self.be.spd_solve(A_full, B_full, float(md.lambda_w.get(f, 0.0)) + EPS, self.w_status)
            torch.maximum(self.w_bad, self.w_status, out=self.w_bad)     # no host round trip inside an iteration
            newW[f] = x.reshape(d, k)
        for f in self.feat_names:           # into the persistent buffers (Jacobi across features: all solves first);
            self.W64[f].copy_(newW[f])      # a tensor born inside a captured iteration lives in the graph's pool
        self._sync_wcat()

    # ---------------------------------------------------------------- stats
    def stats_step(self, it: int):
        """scripts/als.py:503-517: mu update and the five history series."""
        if self.feat_names:
            self.be.compose_z(self.V, self.Xcat, self.Wcat, self.Z)              # :504
        with self._tick("residual_stats"):
            if self.fused_stats:        # per-item (sum d, sum d^2) written by the V-step / the sweep
                self.be.sum_pairs(self.stat_rows, self.stats)
            elif self.fused_feat_stats:  # per-item closed form with the final Z (features present)
                self.be.item_stats(k=self.k, ld=self.ld, item_begin=self.ib, item_end=self.ie, gram=self.gram,
                                   rhs=self.rhs_out, colsum=self.colsum_out, sumr=self.sumr, sumr2=self.sumr2,
                                   indptr=self.csc.indptr, Z=self.Z, b_new=self.b_i, b_old=self.b_i_prev,
                                   stat_out=self.stat_rows, **({"f64": True} if self.v_f64 else {}))
                self.be.sum_pairs(self.stat_rows, self.stats)
            else:
                self.be.residual_stats(k=self.k, ld=self.ld, side=self.csr, U=self.U, Z=self.Z, b_u=self.b_u,
                                       b_i=self.b_i, mu=self.mu, tasks=self.utasks, out=self.stats)
        self._allreduce(self.stats)
        h = self.hist_row                 # fixed address: the iteration can be replayed as a captured graph
        if hasattr(self.be, "history_row"):
            self.be.history_row(U=self.U, V=self.V, b_u=self.b_u, b_i=self.b_i, stats=self.stats, nnz=self.nnz,
                                mu=self.mu, row=h)
        else:
            for j, t in enumerate((self.U, self.V, self.b_u, self.b_i)):
                self.be.sumsq(t, self.ss[j:j + 1])
            mean_d = self.stats[0] / self.nnz
            self.mu += mean_d
            h[0] = torch.sqrt(torch.clamp(self.stats[1] / self.nnz - mean_d * mean_d, min=0.0))
            h[1:5] = torch.sqrt(self.ss)
            h[5] = self.mu[0]
        if it is not None:
            self.hist[it].copy_(h)

    # ------------------------------------------------------------------ run
    def iteration(self, it: int, n_iters: int):
        """One full ALS iteration (scripts/als.py:408-517), asynchronous on the stream."""
        md = self.model
        if self.dev.type == "cuda" and torch.cuda.current_device() != self.dev.index and self.dev.index is not None:
            with torch.cuda.device(self.dev):          # callers that drive iterations themselves (bench.py)
                return self.iteration(it, n_iters)
        do_w = bool(self.feat_names) and ((it % md.update_w_every == 0) or (it == n_iters - 1))   # :468
        if self.model._hip_graph and self.replay_ok and not self.multi and self.timers is None and it > 0:
            self._replay(do_w)
            self.hist[it].copy_(self.hist_row)
        else:
            self._iteration_body(do_w, it)
        self.iters_run = it + 1

    def _iteration_body(self, do_w: bool, it):
        # Z is current here: stats_step recomposes it after every V / W update
        self.user_step()
        b_i_old = self.b_i.clone() if (do_w or self.fused_feat_stats) else None
        self.b_i_prev = b_i_old
        self.item_step(want_gram=do_w or self.fused_feat_stats)
        if do_w:
            with self._tick("w_step"):
                self.w_step(b_i_old)
        self.stats_step(it)

    def _replay(self, do_w: bool):
        """The iteration as a captured HIP graph (one per W-step / no-W-step variant): every launch of the
        body - kernels of libals_hip.so on the capture stream, memsets, the few torch tensor ops - becomes a
        graph node; later iterations replay it with one submission.  The first iteration always runs eagerly
        (allocations, lazy initialisation, occupancy queries)."""
        g = self._graphs.get(do_w)
        if g is None:
            torch.cuda.synchronize(self.dev)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._iteration_body(do_w, None)
            self._graphs[do_w] = g
            self.graphs_captured += 1
        g.replay()

    def run(self, tol, min_iters, verbose):
        md = self.model
        n_iters = int(md.n_iters)
        rm = md.history["train_rmse"]
        base_len = len(rm)
        # Captured-graph replay (hip_graph=True) also serves fits with early stopping: the host reads ONE contiguous
        # 128-byte block per iteration between the replays.  (Round 1 / early round 2 had replay limited to tol=None
        # because "host read-backs corrupt later replays"; the real cause was the sweep's hipMemsetD32Async: with the
        # two iteration graphs alive - W-step / no-W-step variant - each holding a memset node, replays went wrong
        # from the first switch back, reads or no reads.  The reset is a kernel now (gs_sweep.hip) and replay is
        # bitwise the eager fit over 120 iterations with per-iteration reads: profiles/graph_early_stop_stress.py,
        # tests/test_gpu_parity.py::test_hip_graph_replay_with_early_stopping_reads.)
        self.replay_ok = True
        if self.feat_names:
            self.be.compose_z(self.V, self.Xcat, self.Wcat, self.Z)              # :411
        seen = list(rm)                 # train RMSE so far (earlier fits of the same model + this one)
        for it in range(n_iters):
            self.iteration(it, n_iters)
            if tol is not None:
                # one read-back per iteration: the history row (-> the stopping rule) and the status words
                row, words = self._read_ctrl()
                seen.append(float(row[0]))
                if it + 1 >= min_iters:                                           # :520-523
                    self._check_status(words)
                    if len(seen) >= 3 and seen[-3] - seen[-1] <= tol:
                        if verbose > 0:
                            logger.info("Early stopping at iter %d; dRMSE <= %.3g", it + 1, tol)
                        break
        self._check_status()
        self._graphs.clear()            # captured iteration graphs are not needed past the fit

    # --------------------------------------------------------------- export
    def export(self, model: ALS):
        k = self.k
        model.U = self.U[: self.m, :k].to(torch.float64).cpu().numpy()
        model.V = self.V[: self.n, :k].to(torch.float64).cpu().numpy()
        model.b_u = self.b_u[: self.m].to(torch.float64).cpu().numpy()
        model.b_i = self.b_i[: self.n].to(torch.float64).cpu().numpy()
        model.mu = float(self.mu.item())
        for f in self.feat_names:
            model.W[f] = self.W64[f].cpu().numpy()
        h = self.hist[: self.iters_run].cpu().numpy()
        for j, key in enumerate(("train_rmse", "U_norm", "V_norm", "bu_norm", "bi_norm")):
            model.history[key].extend(float(x) for x in h[:, j])

    # -------------------------------------------------------------- predict
    def _compose_for(self, features, features_of_fit: bool = False):
        """Z for `features` as passed to predict (scripts/als.py:568-572): composed from whatever is passed.
        `features_of_fit`: the caller vouches that these are the unchanged arrays of the fit (sweep.SweepDriver,
        which owns them) - the fit's own Z = V + sum_f X_f W_f is then current and nothing is uploaded.  (Round 2
        inferred that from object identity, which says nothing about the contents and can be recycled.)"""
        names = [f for f in features if f in self.W64]
        if not names:
            return self.V
        if features_of_fit and self.iters_run > 0 and names == self.feat_names:
            return self.Z
        Xcat = np.concatenate([np.asarray(features[f], dtype=np.float32) for f in names], axis=1)
        Xp = np.zeros((self.n_pad, Xcat.shape[1]), dtype=np.float32)
        Xp[: self.n] = Xcat
        W = torch.zeros(Xcat.shape[1], self.ld, dtype=torch.float32, device=self.dev)
        off = 0
        for f in names:
            d = features[f].shape[1]
            W[off:off + d, : self.k] = self.W64[f].to(torch.float32)
            off += d
        Z = torch.empty_like(self.V)
        self.be.compose_z(self.V, torch.from_numpy(Xp).to(self.dev), W, Z)
        return Z

    def predict_dense(self, features) -> np.ndarray:
        Z = self._compose_for(features)
        out = torch.empty(self.m, self.n, dtype=torch.float32, device=self.dev)
        self.be.predict_dense(k=self.k, ld=self.ld, m=self.m, n=self.n, U=self.U, Z=Z, b_u=self.b_u,
                              b_i=self.b_i, mu=self.mu, out=out)
        return out.cpu().numpy().astype(np.float64)

    def predict_at(self, flat_idx: np.ndarray, features) -> np.ndarray:
        u, i = np.divmod(flat_idx, self.n)
        us = torch.from_numpy(u.astype(np.int32)).to(self.dev)
        is_ = torch.from_numpy(i.astype(np.int32)).to(self.dev)
        return self.predict_pairs(us, is_, features).cpu().numpy().astype(np.float64)

    def predict_pairs(self, us: torch.Tensor, is_: torch.Tensor, features, features_of_fit: bool = False) -> torch.Tensor:
        """Predictions at (user, item) index tensors already on the device (int32); fp32 device tensor."""
        Z = self._compose_for(features, features_of_fit)
        out = torch.empty(us.numel(), dtype=torch.float32, device=self.dev)
        self.be.predict_at(k=self.k, ld=self.ld, us=us, is_=is_, U=self.U, Z=Z, b_u=self.b_u,
                           b_i=self.b_i, mu=self.mu, out=out)
        return out
