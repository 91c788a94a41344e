"""TEST-ONLY stand-in for `collaborative_filtering_amd.backend.HipBackend`.

Implements the backend interface of the engine (als.py) with numpy on CPU
tensors, using the same per-row formulas as the oracle, and the same output
conventions as the C ABI (perm space, lower-block Gram, factor completion).
It exists so that the host-side logic - sharding, all-gathers, the Gauss-Seidel
level schedule, the W-step algebra, early stopping - can be tested without a
GPU (world_size-2 gloo tests).  It is never imported by the product package.
"""
from __future__ import annotations

import numpy as np
import torch

from collaborative_filtering_amd import layout

EPS = 1e-10
CHUNK = layout.SPLIT_CHUNK


def _np(t):
    return None if t is None else t.detach().numpy()


class NumpyBackend:
    name = "numpy-test"

    def __init__(self, dtype=np.float64):
        self.dtype = dtype

    def slot_bytes(self, k):
        kb = layout.padded_k(k) // 16
        return (kb * (kb + 1) // 2 * 4 + 2 * kb + 2) * 64 * 4

    @staticmethod
    def _rows(tasks):
        t = _np(tasks.tasks)
        return np.unique(t[:, 0]) if t.size else np.zeros(0, np.int64)

    def row_solve(self, *, k, ld, side, F, zero_row, bias_self, bias_other, mu, lam, lam_row, lam_b, lam_b_row,
                  rhs_extra, diag_extra, X_out, bias_out, gram_out, factor_out, rhs_out, colsum_out,
                  sumr_out, status, tasks, workspace, sumr2_out=None, stat_out=None):
        ptr, idx, vals = _np(side.indptr), _np(side.indices), _np(side.vals)
        # zero_row indexes the extra all-zero row behind the [rows_pad, ld] view; unused here
        Fn = _np(F)[:, :k].astype(self.dtype)
        bs, bo = _np(bias_self), _np(bias_other)
        pos = layout.perm_of_col(k)
        mu = float(mu.item())
        blk = np.arange(ld) // 16
        for r in self._rows(tasks):
            lo, hi = ptr[r], ptr[r + 1]
            cols = idx[lo:hi]
            Fr = Fn[cols]
            v = vals[lo:hi].astype(self.dtype)
            base = v - mu - bo[cols]
            rr = base - bs[r]
            G = Fr.T @ Fr
            b = Fr.T @ rr
            cs = Fr.sum(axis=0)
            lam_r = (float(lam_row[r]) if lam_row is not None else float(lam)) + EPS \
                + (float(diag_extra[r]) if diag_extra is not None else 0.0)
            if gram_out is not None:
                Gp = np.zeros((ld, ld), dtype=np.float32)
                Gp[np.ix_(pos[:k], pos[:k])] = G
                Gp[blk[:, None] < blk[None, :]] = np.nan      # upper blocks: not written by the kernel
                gram_out[r] = torch.from_numpy(Gp)
            if rhs_out is not None:
                t = np.zeros(ld, dtype=np.float32); t[pos[:k]] = b; rhs_out[r] = torch.from_numpy(t)
            if colsum_out is not None:
                t = np.zeros(ld, dtype=np.float32); t[pos[:k]] = cs; colsum_out[r] = torch.from_numpy(t)
            if sumr_out is not None:
                sumr_out[r] = float(base.sum())
            if sumr2_out is not None:
                sumr2_out[r] = float((base * base).sum())
            A = G + lam_r * np.eye(k)
            try:
                L = np.linalg.cholesky(A)
            except np.linalg.LinAlgError:
                status[0] = max(int(status[0]), int(r) + 1)
                continue
            if factor_out is not None:
                Ap = np.eye(ld)
                Ap[np.ix_(pos[:k], pos[:k])] = A
                Lp = np.linalg.cholesky(Ap)
                M = Lp + Lp.T
                M[np.diag_indices(ld)] = 1.0 / np.diag(Lp)
                factor_out.view(-1, ld, ld)[r] = torch.from_numpy(M.astype(np.float32))
                continue
            if rhs_extra is not None:
                b = b + _np(rhs_extra)[r, :k]
            x = np.linalg.solve(L.T, np.linalg.solve(L, b))
            xo = np.zeros(ld, dtype=np.float32); xo[:k] = x
            X_out[r] = torch.from_numpy(xo)
            lb = float(lam_b_row[r]) if lam_b_row is not None else float(lam_b)
            bnew = float((base.sum() - cs @ x) / ((hi - lo) + lb + EPS))
            bias_out[r] = bnew
            if stat_out is not None:          # residuals with the new x / bias, computed directly
                d = base - Fr @ x - bnew
                stat_out[r, 0] = float(d.sum()); stat_out[r, 1] = float((d * d).sum())

    def gs_level(self, *, k, ld, items, S_ptr, S_idx, S_val, alpha, factor, rhs, colsum, sumr, indptr,
                 lam_b, lam_b_row, V, bias, sumr2=None, lambda_eff=None, stat_out=None):
        pos = layout.perm_of_col(k)
        sp, si, sv = _np(S_ptr), _np(S_idx), _np(S_val)
        ptr = _np(indptr)
        Fm = _np(factor).reshape(-1, ld, ld)
        Vn = _np(V)
        new = {}
        for i in _np(items):
            M = Fm[i].astype(np.float64)
            L = np.tril(M, -1)
            L[np.diag_indices(ld)] = 1.0 / np.diag(M)
            g = sv[sp[i]:sp[i + 1]].astype(np.float64) @ Vn[si[sp[i]:sp[i + 1]]][:, :k].astype(np.float64)
            b = _np(rhs)[i].astype(np.float64).copy()
            b[pos[:k]] += alpha * g
            xp = np.linalg.solve(L.T, np.linalg.solve(L, b))
            x = xp[pos[:k]]
            lb = float(lam_b_row[i]) if lam_b_row is not None else float(lam_b)
            nnz = ptr[i + 1] - ptr[i]
            csp = _np(colsum)[i].astype(np.float64)
            bnew = (float(sumr[i]) - csp @ xp) / (nnz + lb + EPS)
            st = None
            if stat_out is not None:          # the closed form the kernel uses (no ratings at hand here)
                y = np.linalg.solve(L, b)
                bold = float(bias[i])
                s1 = float(sumr[i]) - nnz * bnew
                s2 = float(sumr2[i]) - 2 * bnew * float(sumr[i]) + nnz * bnew * bnew
                cross = _np(rhs)[i].astype(np.float64) @ xp + (bold - bnew) * (csp @ xp)
                quad = y @ y - float(lambda_eff[i]) * (xp @ xp)
                st = (s1 - csp @ xp, s2 - 2 * cross + quad)
            new[int(i)] = (x, bnew, st)
        for i, (x, bv, st) in new.items():       # a level's items never neighbour each other
            V[i, :k] = torch.from_numpy(x.astype(np.float32))
            bias[i] = float(bv)
            if st is not None:
                stat_out[i, 0], stat_out[i, 1] = float(st[0]), float(st[1])

    # -- W-step (same contracts as als_w_normal_equations / als_spd_solve_f64) ------------------------------
    @staticmethod
    def _gram_storage(gram, i, k, ld):
        """Item Gram in storage column order from the perm-space lower-block image the V-step wrote."""
        pos = layout.perm_of_col(k)[:k]
        Gp = _np(gram)[i].astype(np.float64)
        blk = np.arange(ld) // 16
        low = np.where(blk[:, None] >= blk[None, :], Gp, 0.0)
        strict = np.where(blk[:, None] > blk[None, :], Gp, 0.0)
        return (low + strict.T)[np.ix_(pos, pos)]

    def w_item_vectors(self, *, k, ld, item_begin, item_end, gram, rhs, colsum, V, b_new, b_old, X, feat_off, W, H):
        pos = layout.perm_of_col(k)[:k]
        off = _np(feat_off)
        Xn, Wn, Vn = _np(X).astype(np.float64), _np(W)[:, :k].astype(np.float64), _np(V)
        for i in range(int(item_begin), int(item_end)):
            G = self._gram_storage(gram, i, k, ld)
            db = float(b_new[i]) - float(b_old[i])
            ut_rho = _np(rhs)[i, pos].astype(np.float64) - db * _np(colsum)[i, pos].astype(np.float64)
            xw = [Xn[i, off[f]:off[f + 1]] @ Wn[off[f]:off[f + 1]] for f in range(len(off) - 1)]
            z = Vn[i, :k].astype(np.float64) + sum(xw)
            g = ut_rho - G @ z
            for f in range(len(off) - 1):
                h = np.zeros(ld, dtype=np.float32)
                h[pos] = g + G @ xw[f]
                H[f, i] = torch.from_numpy(h)

    def w_accumulate(self, *, k, ld, item_begin, item_end, gram, X, H, feat_index, feat_col0, feat_d):
        pos = layout.perm_of_col(k)[:k]
        d = int(feat_d)
        A = np.zeros((d, k, d, k))
        B = np.zeros((d, k))
        Xn = _np(X).astype(np.float64)[:, feat_col0:feat_col0 + d]
        for i in range(int(item_begin), int(item_end)):
            x = Xn[i]
            if not x.any():
                continue
            G = self._gram_storage(gram, i, k, ld)
            A += np.einsum("a,b,cd->acbd", x, x, G)
            B += np.outer(x, _np(H)[feat_index, i, pos].astype(np.float64))
        return torch.from_numpy(A.reshape(d * k, d * k)), torch.from_numpy(B.reshape(d * k))

    def spd_solve(self, A, b, diag_add, status):
        An = _np(A) + diag_add * np.eye(A.shape[0])
        try:
            L = np.linalg.cholesky(An)
        except np.linalg.LinAlgError:
            status[0] = 1
            return torch.zeros_like(b)
        status[0] = 0
        return torch.from_numpy(np.linalg.solve(L.T, np.linalg.solve(L, _np(b))))

    def residual_stats(self, *, k, ld, side, U, Z, b_u, b_i, mu, tasks, out):
        ptr, idx, vals = _np(side.indptr), _np(side.indices), _np(side.vals)
        Un, Zn = _np(U)[:, :k].astype(np.float64), _np(Z)[:, :k].astype(np.float64)
        bu, bi = _np(b_u), _np(b_i)
        mu = float(mu.item())
        sd = sd2 = 0.0
        for r in self._rows(tasks):
            cols = idx[ptr[r]:ptr[r + 1]]
            d = vals[ptr[r]:ptr[r + 1]] - (Zn[cols] @ Un[r] + bu[r] + bi[cols] + mu)
            sd += d.sum(); sd2 += (d * d).sum()
        out[0], out[1] = sd, sd2

    def sum_pairs(self, x, out):
        out.copy_(x.view(-1, 2).double().sum(0))

    def sumsq(self, x, out):
        out[0] = float((x.double() ** 2).sum())

    def compose_z(self, V, X, W, Z):
        if X is None:
            Z.copy_(V)
        else:
            Z.copy_(V + X @ W)

    def predict_at(self, *, k, ld, us, is_, U, Z, b_u, b_i, mu, out):
        u, i = us.long(), is_.long()
        out.copy_(((U[u] * Z[i]).sum(1).double() + mu.item() + b_u[u] + b_i[i]).float())

    def predict_dense(self, *, k, ld, m, n, U, Z, b_u, b_i, mu, out):
        out.copy_((U[:m].double() @ Z[:n].double().T + mu.item() + b_u[:m, None] + b_i[None, :n]).float())
