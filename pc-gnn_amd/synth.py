"""Seeded synthetic workloads with the statistics of the reference's datasets.

No real dataset exists offline, so bench / tests use graphs
of the same node / edge / feature-dim / positive-rate shape, built the way
``sparse_to_adjlist`` builds the real ones (src/utils.py:226-239): undirected,
symmetrised, de-duplicated, one self-loop per node.

    yelp_like    N=45,954 F=32  edges 49,315 / 573,616 / 3,402,743   pos 14.53 %
    amazon_like  N=11,944 F=25  edges 175,608 / 3,566,479 / 1,036,737 pos 6.87 %, 3,305 unlabeled
    power_law    N, E free (BASELINE config 4: 10 M nodes / 200 M edges / 3 relations, pos 1 %)
"""
import sys
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np


@dataclass
class Workload:
    name: str
    X: np.ndarray                       # [N, F] f32
    labels: np.ndarray                  # [N] int64
    csr: List[Tuple[np.ndarray, np.ndarray]]   # per relation (indptr int64, indices int32 ascending)
    homo_deg: np.ndarray                # [N] degree in the union graph (pick weights, utils.py:275)
    idx_train: np.ndarray               # sorted ids
    train_pos: List[int]
    meta: dict = field(default_factory=dict)

    @property
    def n(self):
        return self.X.shape[0]


def _csr_from_pairs(n, src, dst):
    a = np.concatenate([src, dst, np.arange(n, dtype=np.int64)])
    b = np.concatenate([dst, src, np.arange(n, dtype=np.int64)])
    key = np.unique(a * n + b)
    rows = key // n
    cols = (key - rows * n).astype(np.int32)
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=indptr[1:])
    return indptr, cols


def _endpoints(rs, n, m, skew, max_share=None):
    """m endpoints: uniform (skew None) or Pareto-popularity weighted."""
    if skew is None:
        return rs.randint(0, n, size=m).astype(np.int64)
    pop = rs.pareto(skew, n) + 0.02
    if max_share is not None:
        pop = np.minimum(pop, max_share * pop.sum())
    cdf = np.cumsum(pop)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rs.rand(m)).astype(np.int64).clip(0, n - 1)


def _homo_degree(n, csr):
    """degree in the union graph.  Exact (set union) for graphs of dataset size; for very large synthetic
    graphs the per-relation degrees are summed minus the shared self-loops (duplicate edges across relations
    are negligible there and the pick weights only need the degree profile)."""
    if sum(idx.shape[0] for _, idx in csr) > 100_000_000:
        deg = sum(np.diff(ip) for ip, _ in csr) - (len(csr) - 1)
        return deg.astype(np.int64)
    keys = []
    for indptr, idx in csr:
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
        keys.append(rows * n + idx)
    key = np.unique(np.concatenate(keys))
    return np.bincount(key // n, minlength=n).astype(np.int64)


def _split(rs, labels, first_labeled, train_ratio):
    """40 % stratified training split (model_handler.py:36-48, sklearn there)."""
    ids = np.arange(first_labeled, len(labels))
    train = []
    for cls in (0, 1):
        c = ids[labels[ids] == cls]
        train.append(rs.choice(c, size=int(round(len(c) * train_ratio)), replace=False))
    idx_train = np.sort(np.concatenate(train))
    return idx_train, [int(v) for v in idx_train if labels[v] == 1]


def make_workload(name, n, feat, rel_edges, pos_rate, seed=0, skew=2.0, first_labeled=0, train_ratio=0.4,
                  nonneg=False, max_share=0.002) -> Workload:
    rs = np.random.RandomState(seed)
    X = rs.randn(n, feat).astype(np.float32)
    if nonneg:   # Amazon features are non-negative and row-normalised (utils.py:213-223)
        X = np.abs(X)
        X = (X / (X.sum(1, keepdims=True) + 0.01)).astype(np.float32)
    labels = (rs.rand(n) < pos_rate).astype(np.int64)
    csr = []
    big = sum(rel_edges) >= 20_000_000
    for m in rel_edges:
        src = _endpoints(rs, n, m, skew, max_share)
        dst = rs.randint(0, n, size=m).astype(np.int64)
        csr.append(_csr_from_pairs(n, src, dst))
        if big:     # long silent phases look like a hang to a watchdog: say what is going on
            print(f"[synth] relation with {m} edges built ({csr[-1][1].shape[0]} CSR entries)", file=sys.stderr, flush=True)
    idx_train, train_pos = _split(rs, labels, first_labeled, train_ratio)
    return Workload(name, X, labels, csr, _homo_degree(n, csr), idx_train, train_pos,
                    {"n": n, "feat": feat, "rel_edges": list(rel_edges), "pos_rate": pos_rate, "seed": seed,
                     "endpoints": "uniform" if skew is None else f"pareto({skew}) x uniform"})


def yelp_like(seed=0, skew=2.0) -> Workload:
    return make_workload("yelpchi-like", 45954, 32, (49315, 573616, 3402743), 0.1453, seed, skew)


def amazon_like(seed=0, skew=2.0) -> Workload:
    return make_workload("amazon-like", 11944, 25, (175608, 3566479, 1036737), 0.0687, seed, skew,
                         first_labeled=3305, nonneg=True)


def power_law(n, n_edges, seed=0, feat=32, pos_rate=0.01, split=(0.05, 0.25, 0.70), skew=1.1, max_share=1e-4) -> Workload:
    """Heavy-tailed graph (BASELINE config 4 shape).  Pareto(1.1) popularity ~ degree
    exponent 2.1; the most popular node is capped at max_share of all endpoints."""
    return make_workload(f"powerlaw-{n}n-{n_edges}e", n, feat, [int(n_edges * s) for s in split], pos_rate, seed,
                         skew, max_share=max_share)


def relabel_by_degree(w: Workload) -> Workload:
    """The same graph with its nodes renumbered by DESCENDING total degree (new id 0 = the largest hub): popular nodes' scores and
    feature rows become neighbours in memory.  A diagnostic (bench.py --relabel-by-degree: does the select kernel's line
    over-fetch of 4-byte score gathers shrink?) - rows are ascending in the NEW ids, so ties at a cut break differently than in `w`."""
    n = w.n
    deg = sum(np.diff(ip) for ip, _ in w.csr)
    perm = np.argsort(-deg, kind="stable")             # perm[new] = old
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n, dtype=np.int64)
    csr = []
    for indptr, idx in w.csr:
        rows_old = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
        key = np.sort(inv[rows_old] * n + inv[idx])    # (row, column) in new ids, row-major, columns ascending
        rows = key // n
        ip = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=ip[1:])
        csr.append((ip, (key - rows * n).astype(np.int32)))
    idx_train = np.sort(inv[w.idx_train])
    return Workload(w.name + "-by-degree", w.X[perm], w.labels[perm], csr, w.homo_deg[perm], idx_train,
                    [int(inv[v]) for v in w.train_pos], dict(w.meta, relabelled="descending total degree"))


def pick_cum_weights(w: Workload) -> np.ndarray:
    """deg / LF, then the sequential fp64 running sum random.choices uses (utils.py:275-278)."""
    y = w.labels[w.idx_train]
    lf = (y.sum() - len(y)) * y + len(y)
    return np.cumsum(w.homo_deg[w.idx_train] / lf)


# ---------------------------------------------------------------------------------------------------------------------
# Sharded generation (multi-GPU runs at BASELINE config 4 scale): every rank builds ITS rows of one global power-law graph
# without any rank ever holding the whole graph.  Everything is a pure function of (seed, node id) or of (seed, relation,
# edge chunk), so the ranks agree on the graph without communicating:
#   features / labels / train split : counter-based (a 64-bit mix of the node id), any subset of nodes on demand
#   edges : chunks of `chunk` (src, dst) pairs from Philox streams keyed by (relation, chunk); a rank walks all chunks and
#           keeps the pairs that touch its id range (transient memory: one chunk; kept: ~2 m / world pairs)
# (A different generator than power_law() above - same shape parameters, not the same graph.)
# ---------------------------------------------------------------------------------------------------------------------
def _mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays."""
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _unit(seed: int, stream: int, idx: np.ndarray) -> np.ndarray:
    """uniform (0, 1) doubles, a pure function of (seed, stream, idx)."""
    with np.errstate(over="ignore"):
        h = _mix64(idx.astype(np.uint64) * np.uint64(0xD1342543DE82EF95) + np.uint64((seed * 1000003 + stream) & 0xFFFFFFFFFFFFFFFF))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def node_features(seed: int, ids: np.ndarray, feat: int) -> np.ndarray:
    """N(0, 1) feature rows of the given nodes (Box-Muller on counter-based uniforms)."""
    ids = np.asarray(ids, dtype=np.int64)
    flat = (ids[:, None] * feat + np.arange(feat, dtype=np.int64)[None, :]).reshape(-1)
    u1, u2 = _unit(seed, 11, flat), _unit(seed, 12, flat)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32).reshape(len(ids), feat)


@dataclass
class ShardedWorkload:
    name: str
    n: int                              # nodes of the whole graph
    bounds: np.ndarray                  # [world + 1] partition
    rank: int
    X_local: np.ndarray                 # [n_local, F]
    labels_local: np.ndarray            # [n_local]
    csr: List[Tuple[np.ndarray, np.ndarray]]   # this rank's rows, neighbour ids GLOBAL
    idx_train_local: np.ndarray         # global ids of this rank's training nodes (sorted)
    labels_train_local: np.ndarray
    homo_deg_train: np.ndarray
    train_pos: List[int]                # global ids of ALL training positives (replicated)
    X_pos: np.ndarray                   # their feature rows (replicated)
    n_train: int                        # global counts (the pick weights' LF, utils.py:276)
    n_train_pos: int
    meta: dict = field(default_factory=dict)


def power_law_shard(n: int, n_edges: int, seed: int, world: int, rank: int, feat: int = 32, pos_rate: float = 0.01,
                    split=(0.05, 0.25, 0.70), skew: float = 1.1, max_share: float = 1e-4, train_ratio: float = 0.4,
                    chunk: int = 1 << 22, balanced: bool = True) -> ShardedWorkload:
    """Rank `rank`'s shard of a heavy-tailed graph with BASELINE config 4's shape parameters."""
    ids_all = np.arange(n, dtype=np.int64)
    pop = np.random.Generator(np.random.Philox(key=seed)).pareto(skew, n) + 0.02
    pop = np.minimum(pop, max_share * pop.sum())
    cdf = np.cumsum(pop)
    cdf /= cdf[-1]
    rel_edges = [int(n_edges * s) for s in split]
    # partition: equal expected CSR entries (popularity endpoints + the uniform ones + the self-loop), no pass over the edges
    if balanced and world > 1:
        m = float(sum(rel_edges))
        exp_deg = m * pop / pop.sum() + m / n + len(rel_edges)
        cum = np.cumsum(exp_deg)
        cuts = np.searchsorted(cum, cum[-1] * np.arange(1, world) / world) + 1
        bounds = np.maximum.accumulate(np.concatenate([[0], np.minimum(cuts, n), [n]])).astype(np.int64)
    else:
        n_per = (n + world - 1) // world
        bounds = np.array([min(r * n_per, n) for r in range(world)] + [n], dtype=np.int64)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    n_local = hi - lo
    labels = (_unit(seed, 1, ids_all) < pos_rate).astype(np.int64)
    in_train = _unit(seed, 2, ids_all) < train_ratio            # (independent 40 % per node: both classes ~40 %)
    idx_train = ids_all[in_train]
    train_pos = idx_train[labels[idx_train] == 1]
    csr = []
    for r, m_r in enumerate(rel_edges):
        keys = [np.arange(lo, hi, dtype=np.int64) * n + np.arange(lo, hi, dtype=np.int64)]       # self-loops
        for c in range((m_r + chunk - 1) // chunk):
            cnt = min(chunk, m_r - c * chunk)
            gen = np.random.Generator(np.random.Philox(key=seed + 1, counter=[r, c, 0, 0]))
            src = np.searchsorted(cdf, gen.random(cnt)).clip(0, n - 1).astype(np.int64)
            dst = gen.integers(0, n, cnt, dtype=np.int64)
            a = (src >= lo) & (src < hi)
            b = (dst >= lo) & (dst < hi)
            keys.append(src[a] * n + dst[a])
            keys.append(dst[b] * n + src[b])
            if m_r >= 20_000_000 and c % 8 == 7:
                print(f"[synth] rank {rank}: relation {r} chunk {c + 1}/{(m_r + chunk - 1) // chunk}", file=sys.stderr, flush=True)
        key = np.unique(np.concatenate(keys))
        rows = key // n - lo
        indptr = np.zeros(n_local + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n_local), out=indptr[1:])
        csr.append((indptr, (key % n).astype(np.int32)))
    tr_local = idx_train[(idx_train >= lo) & (idx_train < hi)]
    homo = (sum(np.diff(ip) for ip, _ in csr) - (len(csr) - 1)).astype(np.int64)      # (duplicate edges across relations are negligible here)
    return ShardedWorkload(
        f"powerlaw-sharded-{n}n-{n_edges}e", n, bounds, rank, node_features(seed, np.arange(lo, hi), feat), labels[lo:hi], csr,
        tr_local, labels[tr_local], homo[tr_local - lo], [int(v) for v in train_pos], node_features(seed, train_pos, feat),
        int(len(idx_train)), int(len(train_pos)),
        {"n": n, "feat": feat, "rel_edges": rel_edges, "pos_rate": pos_rate, "seed": seed,
         "endpoints": f"pareto({skew}) x uniform, generated per rank"})
