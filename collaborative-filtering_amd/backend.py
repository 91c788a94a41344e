"""HIP backend: torch tensors in, C-ABI calls out (include/als_hip.h).

torch is used for what it is good at here - device memory, streams, and the
process group.  All arithmetic of the per-rating / per-row hot path happens in
the hand-written kernels behind the C ABI.  The engine (als.py) talks to this
object only through the methods below, so that the sharding / collective logic
can be exercised on CPU under gloo with a stand-in solver living in tests/.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _hip


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class HipBackend:
    name = "hip"

    GRAM_MODES = {"f32": 0, "f16x2": 1, "f64": 2}
    COND_LIMIT = 300.0
    PLANES_MAX_FLOATS = 16 << 20       # 64 MiB table: V / Z of every BASELINE shape, never U
    PLANES_MAX_FLOATS_K128 = (1 << 31) - 1     # k = 128: any table als_row_solve accepts (element offsets < 2^31)

    def __init__(self, device: torch.device, gram: str = "f16x2", solve_dtype: str = "auto"):
        if gram not in ("f32", "f16x2"):
            raise ValueError("gram must be 'f16x2' or 'f32'")
        if solve_dtype not in ("auto", "float32", "float64"):
            raise ValueError("solve_dtype must be 'auto', 'float32' or 'float64'")
        # solve_dtype="float64": Gram, Cholesky and substitutions of every row in fp64 (ALS_GRAM_F64); `gram` then
        # has no effect
        self.solve_dtype = solve_dtype
        self.gram_mode = self.GRAM_MODES["f64" if solve_dtype == "float64" else gram]
        if device.type != "cuda":
            raise RuntimeError("HipBackend needs a ROCm device (torch device type 'cuda'); "
                               "there is no CPU path in the product")
        self.lib = _hip.load()
        self.device = device
        self._sumsq_partials = torch.empty(self.lib.als_sumsq_partials(), dtype=torch.float64,
                                           device=device)
        self._stats_partials: Optional[torch.Tensor] = None
        self._spd_ws: Optional[torch.Tensor] = None
        # operand scale of the f16x2 Gram ({S, 1 / S^2} of the gathered factor matrix + two working words that stay
        # zero between calls: als_factor_scale); one per backend - calls on one stream run in order
        self._fscale = torch.zeros(4, dtype=torch.float32, device=device)
        self.ablate = int(os.environ.get("ALS_ABLATE", "0"))   # phase ablation of als_row_solve (profiles/ablate.sh)
        # scratch for the pre-split operands of the Gram (row_solve; keyed by table size) and the largest gathered table
        # (floats) it is used for; ALS_PLANES=0 switches the path off
        self._planes: dict = {}
        self.planes_max_floats = 0 if os.environ.get("ALS_PLANES", "1") == "0" else self.PLANES_MAX_FLOATS
        self.planes_max_floats_k128 = self.PLANES_MAX_FLOATS_K128 if os.environ.get("ALS_PLANES_K128") == "1" else 0
        # solve_dtype="auto": rows whose condition estimate (two lower bounds of cond_2 from their own fp32
        # factorisation: the pivot ratio (max L_ii / min L_ii)^2 and (trace(G) / rank + lambda) / min L_ii^2) exceeds
        # COND_LIMIT are redone in fp64 by the same call, as are rows whose closed-form residual statistics cancel to
        # fewer than three digits.  Calibration (profiles/r03_cond_estimates.txt): rows of cfg 4 / cfg 3 stay below
        # 5, of cfg 5 (k = 128, |z| up to 3) below 170 - no row of the BASELINE workloads is redone -, the
        # lambda = 1e-2 fixture 14 ... 1300, lambda = 1e-4 10^3 ... 3 10^5.  A row that stays fp32 has a relative
        # error of at most about COND_LIMIT * 3e-7 = 1e-4 (typically 1e-5).  ALS_COND_LIMIT overrides.
        self.cond_limit = float(os.environ.get("ALS_COND_LIMIT", self.COND_LIMIT)) if solve_dtype == "auto" else 0.0
        self._redo_count = torch.zeros(1, dtype=torch.int32, device=device)
        self._redo_rows: dict = {}         # rows of the orientation -> int32 list buffer
        self.cond_probe: Optional[torch.Tensor] = None     # diagnostics: float32 [nrows], receives every row's estimate

    # -- helpers -------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _check(rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed with status {rc} (see ALS_E_* in include/als_hip.h)")

    def slot_bytes(self, k: int, f64: bool = False) -> int:
        """Bytes of one partial slot of a split row (fp64 calls keep an fp64 image per segment)."""
        if f64 or self.gram_mode == self.GRAM_MODES["f64"]:
            return int(self.lib.als_partial_slot_bytes_f64(k))
        return int(self.lib.als_partial_slot_bytes(k))

    # -- K1 ------------------------------------------------------------------
    def row_solve(self, *, k, ld, side, F, zero_row, bias_self, bias_other, mu, lam, lam_row, lam_b,
                  lam_b_row, rhs_extra, diag_extra, X_out, bias_out, gram_out, factor_out,
                  rhs_out, colsum_out, sumr_out, status, tasks, workspace, sumr2_out=None, stat_out=None,
                  f64=False):
        """`f64`: this call in fp64 whatever the backend's mode, by-product arrays (gram_out, factor_out, rhs_out,
        colsum_out, sumr_out, sumr2_out) being DOUBLE tensors (the engine's fp64 V-step)."""
        p = _hip.RowSolveParams()
        p.k, p.ld, p.nrows, p.F_zero_row = k, ld, side.nrows, int(zero_row)
        p.reserved0 = self.ablate          # 0 in production; profiling builds of bench.py set it
        p.gram_mode = self.GRAM_MODES["f64"] if f64 else self.gram_mode
        p.byproducts_f64 = int(bool(f64))
        p.indptr, p.indices, p.vals = _p(side.indptr), _p(side.indices), _p(side.vals)
        p.F, p.bias_self, p.bias_other, p.mu = _p(F), _p(bias_self), _p(bias_other), _p(mu)
        p.lambda_scalar, p.lambda_row = float(lam), _p(lam_row)
        p.lambda_bias_scalar, p.lambda_bias_row = float(lam_b), _p(lam_b_row)
        p.rhs_extra, p.diag_extra = _p(rhs_extra), _p(diag_extra)
        p.X_out, p.bias_out, p.gram_out, p.factor_out = _p(X_out), _p(bias_out), _p(gram_out), _p(factor_out)
        p.rhs_out, p.colsum_out, p.sumr_out, p.status = _p(rhs_out), _p(colsum_out), _p(sumr_out), _p(status)
        p.sumr2_out, p.stat_out = _p(sumr2_out), _p(stat_out)
        p.tasks, p.ntasks = _p(tasks.tasks), tasks.ntasks
        p.ndual_tail = int(getattr(tasks, "ndual", 0))      # honoured by the library for plain solves
        p.ndual_mid = int(getattr(tasks, "nmid", 0))
        p.long_rows, p.nlong = _p(tasks.long_rows), tasks.nlong
        p.workspace = _p(workspace)
        p.F_scale, p.F_scale_ready = _p(self._fscale), 0
        # pre-split operands (als_row_solve_params::F_planes): k = 49 ... 64, f16x2 Gram, a gathered table small enough
        # that the extra pass over it is noise - in practice the U-step (its table is V / Z); the V-step's table (U) is
        # large and that launch is bound by the gather, not by vector issue
        # (the zero row is the table's last row - right behind the view the engine passes as F)
        # k = 113 ... 128: the kernel has the path too (tested), but it measures +-0 there - 124 instead of 108 matrix
        # instructions per 32 ratings cost what the vector instructions save (cfg 5 / 50: V-step 1.93 -> 1.89 ms,
        # U-step unchanged) - so it is off unless ALS_PLANES_K128=1
        nwords = (int(zero_row) + 1) * ld
        limit = self.planes_max_floats if ld == 64 else (self.planes_max_floats_k128 if ld == 128 else 0)
        if (not f64 and p.gram_mode == self.GRAM_MODES["f16x2"] and 0 < nwords <= limit
                and int(zero_row) >= F.shape[0] - 1 and F.is_contiguous()):
            buf = self._planes.get(nwords)
            if buf is None:
                buf = self._planes[nwords] = torch.empty(nwords, dtype=torch.int32, device=self.device)
            p.F_planes = _p(buf)
        if self.cond_limit > 0.0 and not self.ablate and not f64:
            if side.nrows not in self._redo_rows:
                self._redo_rows[side.nrows] = torch.empty(max(side.nrows, 1), dtype=torch.int32, device=self.device)
            p.cond_limit = self.cond_limit
            p.redo_count, p.redo_rows = _p(self._redo_count), _p(self._redo_rows[side.nrows])
            p.cond_out = _p(self.cond_probe) if (self.cond_probe is not None and self.cond_probe.numel() >= side.nrows) else None
        self._check(self.lib.als_row_solve(C.byref(p), self._stream()), "als_row_solve")

    # -- K2 ------------------------------------------------------------------
    def gs_level(self, *, k, ld, items, S_ptr, S_idx, S_val, alpha, factor, rhs, colsum, sumr,
                 indptr, lam_b, lam_b_row, V, bias, sumr2=None, lambda_eff=None, stat_out=None, f64=False):
        p = _hip.GsSweepParams()
        p.f64 = int(bool(f64))
        p.sumr2, p.lambda_eff, p.stat_out = _p(sumr2), _p(lambda_eff), _p(stat_out)
        p.k, p.ld = k, ld
        p.items, p.nitems = _p(items), items.numel()
        p.S_ptr, p.S_idx, p.S_val, p.alpha = _p(S_ptr), _p(S_idx), _p(S_val), float(alpha)
        p.factor, p.rhs, p.colsum, p.sumr = _p(factor), _p(rhs), _p(colsum), _p(sumr)
        p.indptr, p.lambda_bias_scalar, p.lambda_bias_row = _p(indptr), float(lam_b), _p(lam_b_row)
        p.V, p.bias = _p(V), _p(bias)
        self._check(self.lib.als_gs_sweep(C.byref(p), self._stream()), "als_gs_sweep")

    def _gs_params(self, items, kw):
        p = _hip.GsSweepParams()
        p.k, p.ld = kw["k"], kw["ld"]
        p.items, p.nitems = _p(items), items.numel()
        p.S_ptr, p.S_idx, p.S_val, p.alpha = _p(kw["S_ptr"]), _p(kw["S_idx"]), _p(kw["S_val"]), float(kw["alpha"])
        p.factor, p.rhs, p.colsum, p.sumr = _p(kw["factor"]), _p(kw["rhs"]), _p(kw["colsum"]), _p(kw["sumr"])
        p.indptr, p.lambda_bias_scalar, p.lambda_bias_row = _p(kw["indptr"]), float(kw["lam_b"]), _p(kw["lam_b_row"])
        p.V, p.bias = _p(kw["V"]), _p(kw["bias"])
        p.sumr2, p.lambda_eff, p.stat_out = _p(kw.get("sumr2")), _p(kw.get("lambda_eff")), _p(kw.get("stat_out"))
        p.f64 = int(bool(kw.get("f64", False)))
        return p

    def gs_dataflow(self, *, items, S_idx_wait, publish, err, nondep=None, **kw):
        """Whole sweep as one persistent launch; see als_gs_sweep_dataflow."""
        p = self._gs_params(items, kw)
        assert publish.shape == kw["V"].shape and publish.is_contiguous()
        assert nondep is None or (nondep.shape == kw["V"].shape and nondep.is_contiguous())
        self._check(self.lib.als_gs_sweep_dataflow(C.byref(p), _p(S_idx_wait), _p(publish), publish.shape[0], _p(nondep),
                                                   _p(err), self._stream()), "als_gs_sweep_dataflow")

    def gs_levels(self, *, offsets, **kw):
        """All levels of the sweep with one C call (offsets: host int64 numpy array, nlevels+1)."""
        items = kw.pop("items")
        p = _hip.GsSweepParams()
        p.k, p.ld = kw["k"], kw["ld"]
        p.items, p.nitems = _p(items), 0
        p.S_ptr, p.S_idx, p.S_val, p.alpha = _p(kw["S_ptr"]), _p(kw["S_idx"]), _p(kw["S_val"]), float(kw["alpha"])
        p.factor, p.rhs, p.colsum, p.sumr = _p(kw["factor"]), _p(kw["rhs"]), _p(kw["colsum"]), _p(kw["sumr"])
        p.indptr, p.lambda_bias_scalar, p.lambda_bias_row = _p(kw["indptr"]), float(kw["lam_b"]), _p(kw["lam_b_row"])
        p.V, p.bias = _p(kw["V"]), _p(kw["bias"])
        p.sumr2, p.lambda_eff, p.stat_out = _p(kw.get("sumr2")), _p(kw.get("lambda_eff")), _p(kw.get("stat_out"))
        p.f64 = int(bool(kw.get("f64", False)))
        off = offsets.ctypes.data_as(C.c_void_p)
        self._check(self.lib.als_gs_sweep_levels(C.byref(p), off, len(offsets) - 1, self._stream()),
                    "als_gs_sweep_levels")

    # -- K3/K4 ---------------------------------------------------------------
    def w_item_vectors(self, *, k, ld, item_begin, item_end, gram, rhs, colsum, V, b_new, b_old, X, feat_off,
                       W, H, f64=False):
        """`f64`: gram / rhs / colsum / H are double tensors and W is the fp64 projection matrix [D, k]."""
        p = _hip.WParams()
        p.f64 = int(bool(f64))
        p.k, p.ld, p.phase, p.nfeat = k, ld, 0, feat_off.numel() - 1
        p.item_begin, p.item_end = int(item_begin), int(item_end)
        p.gram, p.rhs, p.colsum, p.V, p.b_new, p.b_old = _p(gram), _p(rhs), _p(colsum), _p(V), _p(b_new), _p(b_old)
        p.D, p.X, p.feat_off, p.W, p.H, p.nrows_h = X.shape[1], _p(X), _p(feat_off), _p(W), _p(H), H.shape[1]
        self._check(self.lib.als_w_normal_equations(C.byref(p), self._stream()), "als_w_normal_equations(0)")

    def w_accumulate(self, *, k, ld, item_begin, item_end, gram, X, H, feat_index, feat_col0, feat_d, f64=False):
        """(A [(d k)^2], B [d k]) fp64, storage order, for one feature over items [item_begin, item_end)."""
        npairs = feat_d * (feat_d + 1) // 2
        nit = max(int(item_end) - int(item_begin), 1)
        nchunks = max(1, min(512, 4096 // npairs, -(-nit // 64)))
        kb = ld // 16
        dbl = torch.float64
        partA = torch.empty(npairs * nchunks * (kb * (kb + 1) // 2) * 256, dtype=dbl, device=self.device)
        partB = torch.empty(feat_d * nchunks * ld, dtype=dbl, device=self.device)
        A = torch.empty(feat_d * k, feat_d * k, dtype=dbl, device=self.device)
        B = torch.empty(feat_d * k, dtype=dbl, device=self.device)
        p = _hip.WParams()
        p.f64 = int(bool(f64))
        p.k, p.ld, p.phase = k, ld, 1
        p.item_begin, p.item_end = int(item_begin), int(item_end)
        p.gram, p.D, p.X, p.H, p.nrows_h = _p(gram), X.shape[1], _p(X), _p(H), H.shape[1]
        p.feat_index, p.feat_col0, p.feat_d, p.nchunks = feat_index, feat_col0, feat_d, nchunks
        p.partA, p.partB, p.A_out, p.B_out = _p(partA), _p(partB), _p(A), _p(B)
        self._check(self.lib.als_w_normal_equations(C.byref(p), self._stream()), "als_w_normal_equations(1)")
        return A, B

    def spd_solve(self, A: torch.Tensor, b: torch.Tensor, diag_add: float, status: torch.Tensor) -> torch.Tensor:
        """x = (A + diag_add I)^-1 b, dense fp64 Cholesky (scripts/als.py:497-500).  status: device int32[1]."""
        N = A.shape[0]
        assert A.dtype == torch.float64 and A.is_contiguous() and A.shape == (N, N) and b.numel() == N
        nbytes = int(self.lib.als_spd_solve_workspace_bytes(N))
        if nbytes == 0:
            raise ValueError(f"W-step system of order {N} exceeds ALS_SPD_MAX_N")
        if self._spd_ws is None or self._spd_ws.numel() < nbytes:
            self._spd_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        x = torch.empty(N, dtype=torch.float64, device=self.device)
        self._check(self.lib.als_spd_solve_f64(N, _p(A), N, _p(b), float(diag_add), _p(x), _p(self._spd_ws),
                                               _p(status), self._stream()), "als_spd_solve_f64")
        return x

    # -- K6 ------------------------------------------------------------------
    def residual_stats(self, *, k, ld, side, U, Z, b_u, b_i, mu, tasks, out):
        if tasks.ntasks == 0:
            out.zero_()
            return
        need = 2 * ((tasks.ntasks + 3) // 4)
        if self._stats_partials is None or self._stats_partials.numel() < need:
            self._stats_partials = torch.empty(need, dtype=torch.float64, device=self.device)
        self._check(self.lib.als_residual_stats(
            k, ld, _p(side.indptr), _p(side.indices), _p(side.vals), _p(U), _p(Z), _p(b_u), _p(b_i),
            _p(mu), _p(tasks.tasks), tasks.ntasks, _p(self._stats_partials), _p(out), self._stream()),
            "als_residual_stats")

    def item_stats(self, *, k, ld, item_begin, item_end, gram, rhs, colsum, sumr, sumr2, indptr, Z, b_new, b_old,
                   stat_out, f64=False):
        """Per-item closed-form residual sums when Z != V (als_item_stats / als_item_stats_f64)."""
        fn = self.lib.als_item_stats_f64 if f64 else self.lib.als_item_stats
        self._check(fn(k, ld, int(item_begin), int(item_end), _p(gram), _p(rhs), _p(colsum),
                       _p(sumr), _p(sumr2), _p(indptr), _p(Z), _p(b_new), _p(b_old),
                       _p(stat_out), self._stream()), "als_item_stats")

    def sum_pairs(self, x: torch.Tensor, out: torch.Tensor):
        """out[0:2] = column sums of x viewed as [n, 2] (fp64, deterministic)."""
        x = getattr(x, "base", x)           # a rank-local by-product array (als._RowShift): reduce what exists
        if getattr(self, "_pair_partials", None) is None:
            self._pair_partials = torch.empty(2 * self._sumsq_partials.numel(), dtype=torch.float64, device=self.device)
        part = self._pair_partials
        self._check(self.lib.als_sum_pairs(_p(x), x.numel() // 2, _p(part), _p(out), self._stream()),
                    "als_sum_pairs")

    def history_row(self, *, U, V, b_u, b_i, stats, nnz, mu, row):
        """mu update + the five history values of an iteration in two launches (als_history_row)."""
        if getattr(self, "_hist_partials", None) is None:
            self._hist_partials = torch.empty(4 * self._sumsq_partials.numel(), dtype=torch.float64, device=self.device)
        self._check(self.lib.als_history_row(_p(U), U.numel(), _p(V), V.numel(), _p(b_u), b_u.numel(), _p(b_i), b_i.numel(),
                                             _p(stats), int(nnz), _p(mu), _p(self._hist_partials), _p(row), self._stream()),
                    "als_history_row")

    def sumsq(self, x: torch.Tensor, out: torch.Tensor):
        self._check(self.lib.als_sumsq(_p(x), x.numel(), _p(self._sumsq_partials), _p(out),
                                       self._stream()), "als_sumsq")

    # -- K0 / K7 ---------------------------------------------------------------
    def compose_z(self, V, X, W, Z):
        D = 0 if X is None else X.shape[1]
        self._check(self.lib.als_compose_z(V.shape[0], V.shape[1], D, _p(V), _p(X), _p(W), _p(Z),
                                           self._stream()), "als_compose_z")

    def predict_at(self, *, k, ld, us, is_, U, Z, b_u, b_i, mu, out):
        self._check(self.lib.als_predict_at(k, ld, us.numel(), _p(us), _p(is_), _p(U), _p(Z), _p(b_u),
                                            _p(b_i), _p(mu), _p(out), self._stream()), "als_predict_at")

    def predict_dense(self, *, k, ld, m, n, U, Z, b_u, b_i, mu, out):
        self._check(self.lib.als_predict_dense(k, ld, m, n, _p(U), _p(Z), _p(b_u), _p(b_i), _p(mu),
                                               _p(out), self._stream()), "als_predict_dense")
