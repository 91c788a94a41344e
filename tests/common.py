"""Helpers shared by the test modules: golden-fixture loading and oracle runs."""
from __future__ import annotations

import glob
import json
import os

import numpy as np

from oracle.als_oracle import OracleALS, OracleConfig, ratings_from_coo
from tests.synth import make_features, make_ratings

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """ALS fit fixtures (g1 ... g9); other .npz files in the directory are fold-file fixtures."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "g[0-9]*.npz")))


class Golden:
    """One fixture written by tests/golden/make_golden.py."""

    def __init__(self, name: str):
        self.name = name
        self.d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.cfg = json.loads(str(self.d["cfg"]))
        c = self.cfg
        self.m, self.n = int(self.d["shape"][0]), int(self.d["shape"][1])
        if "rows" in self.d.files:
            rows, cols, vals = (self.d["rows"].astype(np.int64), self.d["cols"].astype(np.int64),
                                self.d["vals"].astype(np.float64))
        else:   # large fixture: inputs are regenerated from the recorded seed
            rows, cols, vals = make_ratings(c["m"], c["n"], c["nnz"], c["seed"],
                                            empty_users=c["empty_users"],
                                            empty_items=c["empty_items"])
        self.rows, self.cols, self.vals = rows, cols, vals
        self.val_pos = self.d["val_pos"]
        tr = np.ones(rows.size, dtype=bool)
        tr[self.val_pos] = False
        self.train = (rows[tr], cols[tr], vals[tr])
        G, Y = make_features(self.n, c["seed"] + 11)
        if c.get("dup_cols"):
            G = np.concatenate([G, G[:, : c["dup_cols"]]], axis=1)
        allf = {"genres": G, "years": Y}
        self.features = {f: allf[f] for f in c["feats"]}

    def oracle_config(self) -> OracleConfig:
        c = self.cfg
        return OracleConfig(
            n_factors=c["n_factors"], n_iters=c["n_iters"], lambda_u=c["lambda_u"],
            lambda_v=c["lambda_v"], pop_reg_mode=c["pop_reg_mode"], random_state=42,
            update_w_every=c["update_w_every"], lambda_bu=c["lambda_bu"],
            lambda_bi=c["lambda_bi"], alpha=c["alpha"], sim=c["sim"],
            lambda_w=dict(c["lambda_w"] or {}))

    def train_ratings(self):
        r, c, v = self.train
        return ratings_from_coo(r, c, v, (self.m, self.n))

    def val_flat(self):
        return self.rows[self.val_pos] * self.n + self.cols[self.val_pos]

    def val_truth(self):
        return self.vals[self.val_pos]

    def S_csr(self):
        if "S_ptr" not in self.d.files:
            return None
        return (self.d["S_ptr"], self.d["S_idx"].astype(np.int64), self.d["S_val"])

    def run_oracle(self, use_pinned_S: bool = False) -> OracleALS:
        o = OracleALS(self.oracle_config())
        o.fit(self.train_ratings(), self.features, tol=self.cfg["tol"],
              min_iters=self.cfg["min_iters"],
              S_csr=self.S_csr() if use_pinned_S else None)
        return o
