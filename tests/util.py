"""Shared helpers for the test-suite: golden-fixture loading and seeded
synthetic graphs.  Reads only files under tests/golden - never /root/reference."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAM_KEYS = lambda R: (["weight", "inter1.weight", "inter1.label_clf.weight", "inter1.label_clf.bias"]
                        + [f"inter1.intra_agg{r + 1}.weight" for r in range(R)])


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        z = self.z
        self.n, self.f, self.R, self.emb = int(z["n"]), int(z["f"]), int(z["R"]), int(z["emb"])
        self.alpha, self.lr, self.wd = float(z["alpha"]), float(z["lr"]), float(z["wd"])
        self.rhos = [float(r) for r in z["rhos"]]
        self.X = z["X"]
        self.labels = z["labels"]
        self.nodes = z["nodes"].tolist()
        self.batch_labels = z["batch_labels"]
        self.train_pos = z["train_pos"].tolist()
        self.csr = [(z[f"indptr{r}"], z[f"indices{r}"]) for r in range(self.R)]
        self.homo_csr = (z["homo_indptr"], z["homo_indices"])

    def adj(self, r=None):
        indptr, idx = self.homo_csr if r is None else self.csr[r]
        return {v: set(idx[indptr[v]:indptr[v + 1]].tolist()) for v in range(self.n)}

    def adj_lists(self):
        return [self.adj(r) for r in range(self.R)]

    def params(self, prefix="w_"):
        import torch
        return {k: torch.from_numpy(self.z[prefix + k].copy()) for k in PARAM_KEYS(self.R)}

    def sel(self, key, r):
        off, idx = self.z[f"{key}_sel_off{r}"], self.z[f"{key}_sel_idx{r}"]
        return [set(idx[off[b]:off[b + 1]].tolist()) for b in range(len(off) - 1)]

    def sel_csr(self, key, r):
        return self.z[f"{key}_sel_off{r}"], self.z[f"{key}_sel_idx{r}"]


def synth_graph(seed, n, feat_dim, rel_avg_deg, pos_rate, skew=1.5, hub=True):
    """Seeded synthetic multi-relation graph in the reference's input shape
    (symmetric, self-loops).  Returns X, labels, [ (indptr,indices) ], adj dicts are
    not built (use csr_to_adj for small cases)."""
    rs = np.random.RandomState(seed)
    X = rs.randn(n, feat_dim).astype(np.float32)
    labels = (rs.rand(n) < pos_rate).astype(np.int64)
    pop = rs.pareto(skew, n) + 0.05
    if hub:
        pop[min(7, n - 1)] = pop.max() * 4
    pop /= pop.sum()
    csrs = []
    for avg in rel_avg_deg:
        m = int(n * avg / 2)
        src = rs.choice(n, size=m, p=pop)
        dst = rs.randint(0, n, size=m)
        a = np.concatenate([src, dst, np.arange(n)])
        b = np.concatenate([dst, src, np.arange(n)])
        key = np.unique(a.astype(np.int64) * n + b)
        rows, cols = key // n, (key % n).astype(np.int32)
        indptr = np.zeros(n + 1, dtype=np.int64)
        np.add.at(indptr, rows + 1, 1)
        indptr = np.cumsum(indptr)
        csrs.append((indptr, cols))
    return X, labels, csrs


def csr_to_adj(csr, n):
    indptr, idx = csr
    return {v: set(idx[indptr[v]:indptr[v + 1]].tolist()) for v in range(n)}
