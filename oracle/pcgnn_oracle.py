"""CPU restatement of the PC-GNN pick -> choose -> aggregate hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``): the checker for the HIP
path and the timed ``cpu_baseline`` ("port") in ``bench.py``.  Parity status:
PINNED by ``tests/golden/*.npz`` (generated from the imported reference by
``tests/golden/make_golden.py``).

It keeps the reference's *algorithmic shape* on purpose - Python sets / dicts,
one ``torch.sort`` per centre node per relation, a dense ``[B x U]`` mask matmul
for the mean - so that timing it on the GPU box's host cores is a fair stand-in
for the reference's own CPU path (which cannot travel to that box).

Every function cites the reference lines it follows (paths are relative to
``/root/reference``).

The single deliberate difference: where the reference leaves an order
unspecified, the oracle fixes one, and the HIP path implements the same one:

* neighbour lists are taken in ascending node-id order (the reference uses
  CPython ``list(set)`` iteration order, ``src/layers.py:246-248``);
* the distance sort is *stable* - ties on ``|c - s_j|`` are broken by list
  position (the reference calls ``torch.sort`` without ``stable=True``,
  ``src/layers.py:658,687,722``, whose tie order is unspecified).

Both only matter when two candidates straddling the cut have bit-identical
f32 distances; any outcome of such a tie is a valid outcome of the reference.
"""
from __future__ import annotations

import bisect
import math
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch
import torch.nn.functional as F

AdjList = Dict[int, Set[int]]


# --------------------------------------------------------------------------
# P1  pick  (src/utils.py:274-278)
# --------------------------------------------------------------------------
def pick_weights(idx_train: Sequence[int], y_train: np.ndarray, homo: AdjList) -> np.ndarray:
    """Sampling weight deg(v) / LF(label(v)).  src/utils.py:275-277.

    ``LF`` is #train-positives for a positive node and ``len(y_train)`` (not
    #negatives) for a negative one - that is what :276 evaluates to.
    """
    deg = np.array([len(homo[v]) for v in idx_train])
    y = np.asarray(y_train)
    lf = (y.sum() - len(y)) * y + len(y)
    return deg / lf


def pick_cum_weights(weights: np.ndarray) -> np.ndarray:
    """Sequential fp64 running sum == ``itertools.accumulate`` inside
    ``random.choices`` (CPython 3.10 ``Lib/random.py``; call site utils.py:278)."""
    return np.cumsum(np.asarray(weights, dtype=np.float64))


def pick_from_uniforms(idx_train: Sequence[int], cum: np.ndarray, uniforms: Sequence[float]) -> List[int]:
    """``random.choices`` body given its ``random()`` draws: bisect_right on
    ``u * total`` with ``hi = n - 1``.  utils.py:278."""
    total = float(cum[-1]) + 0.0
    hi = len(cum) - 1
    cl = cum.tolist()
    return [idx_train[bisect.bisect(cl, u * total, 0, hi)] for u in uniforms]


def pick_step(idx_train, y_train, homo, size, rng) -> List[int]:
    """utils.py:274-278 with an explicit ``random.Random`` instance."""
    w = pick_weights(idx_train, y_train, homo)
    return rng.choices(idx_train, weights=w, k=size)


# --------------------------------------------------------------------------
# C1 / C2  choose  (src/layers.py:633-738)
# --------------------------------------------------------------------------
def sample_count(deg: int, threshold: float) -> int:
    """layers.py:260-262."""
    return math.ceil(deg * threshold)


def choose_row(c0: torch.Tensor, ids: Sequence[int], s0: torch.Tensor, k: int,
               minority: Optional[Tuple[Sequence[int], torch.Tensor, int]] = None) -> Set[int]:
    """One centre node, one relation.  layers.py:648-694 (train) / 713-735 (test).

    c0   : 0-dim f32, the centre's class-0 logit (:649)
    ids  : neighbour ids (ascending), s0: their class-0 logits [deg] f32 (:650)
    k    : num_sample (:653)
    minority = (train_pos ids, their class-0 logits [P], m) for a positive
               centre in training (:675-691); ``m = int(k * rho)`` (:681).
    """
    diff = torch.abs(c0 - s0)                                   # :657
    if len(ids) > k + 1:                                        # :662
        order = torch.sort(diff, dim=0, stable=True).indices    # :658 (+ stable)
        kept = [ids[i] for i in order[:k].tolist()]             # :664
    else:
        kept = list(ids)                                        # :669 keep everything
    if minority is not None:
        pos_ids, pos_s0, m = minority
        dm = torch.abs(c0 - pos_s0)                             # :685
        om = torch.sort(dm, dim=0, stable=True).indices         # :687
        kept.extend(pos_ids[i] for i in om[:m].tolist())        # :690
    return set(kept)                                            # :694


def choose_sets(center_s0: torch.Tensor, labels: Optional[Sequence[int]],
                neigh_lists: Sequence[Sequence[int]], neigh_s0: Sequence[torch.Tensor],
                pos_ids: Sequence[int], pos_s0: torch.Tensor,
                threshold: float, rho: float, train_flag: bool) -> List[Set[int]]:
    """choose_step_neighs / choose_step_test for one relation (layers.py:633,700)."""
    out = []
    for b, ids in enumerate(neigh_lists):
        k = sample_count(len(ids), threshold)
        mino = None
        if train_flag and int(labels[b]) == 1:                  # :675
            mino = (pos_ids, pos_s0, int(k * rho))              # :681
        out.append(choose_row(center_s0[b], ids, neigh_s0[b], k, mino))
    return out


# --------------------------------------------------------------------------
# G1  mean aggregation  (src/layers.py:594-624; graphsage.py:78-95, 210-231)
# --------------------------------------------------------------------------
def dense_mask_aggregate(sets: Sequence[Set[int]], X: torch.Tensor, norm: str = "count") -> torch.Tensor:
    """The reference's dense formulation: mask[B,U] of ones, divide by the row
    count (or its sqrt for GCN), ``mask.mm(X[unique])``.  layers.py:594-624,
    graphsage.py:78-95 (count), 210-231 (sqrt)."""
    uniq = sorted(set.union(*sets))
    col = {n: i for i, n in enumerate(uniq)}
    mask = torch.zeros(len(sets), len(uniq))
    cols = [col[n] for s in sets for n in s]
    rows = [i for i, s in enumerate(sets) for _ in range(len(s))]
    mask[rows, cols] = 1
    cnt = mask.sum(1, keepdim=True)
    mask = mask.div(cnt.sqrt() if norm == "sqrt_count" else cnt)
    return mask.mm(X[torch.LongTensor(uniq)])


def sparse_aggregate(sets: Sequence[Set[int]], X: torch.Tensor, norm: str = "count") -> torch.Tensor:
    """Same quantity without the [B x U] mask (for large test cases)."""
    out = torch.empty(len(sets), X.shape[1])
    for b, s in enumerate(sets):
        ids = torch.LongTensor(sorted(s))
        n = float(len(ids))
        out[b] = X[ids].sum(0) / (math.sqrt(n) if norm == "sqrt_count" else n)
    return out


# --------------------------------------------------------------------------
# A1-A8, G1, M1, B1  the PC-GNN layer  (src/layers.py:161-291,539-630; src/model.py)
# --------------------------------------------------------------------------
class OraclePCGNN:
    """R-generic restatement of ``PCALayer(InterAgg{1,3,5}(IntraAgg x R))``.

    Parameters use the reference's state-dict names:
      weight                         [2, E]        model.py:29
      inter1.weight                  [F + R*E, E]  layers.py:196
      inter1.intra_agg{r}.weight     [2F, E]       layers.py:559
      inter1.label_clf.weight/.bias  [2, F] / [2]  layers.py:200
    """

    def __init__(self, X: torch.Tensor, adj_lists: Sequence[AdjList], train_pos: Sequence[int],
                 params: Dict[str, torch.Tensor], rho: float, lambda_1: float,
                 thresholds: Optional[Sequence[float]] = None, dense_mask: bool = True):
        self.X = X.float()
        self.adj = list(adj_lists)
        self.R = len(self.adj)
        self.train_pos = list(train_pos)
        self.rho = rho
        self.lambda_1 = lambda_1
        self.thresholds = list(thresholds) if thresholds is not None else [0.5] * self.R  # layers.py:193
        self.dense_mask = dense_mask
        self.p = {k: v.clone().float().requires_grad_(True) for k, v in params.items()}
        self.last_sets: List[List[Set[int]]] = []

    # -- label-aware scores (A3/A4, layers.py:230-243) ----------------------
    def _label_clf(self, rows: torch.Tensor) -> torch.Tensor:
        return F.linear(rows, self.p["inter1.label_clf.weight"], self.p["inter1.label_clf.bias"])

    def forward(self, nodes: Sequence[int], labels, train_flag: bool = True):
        """InterAgg.forward + PCALayer.forward.  layers.py:207-291, model.py:34-39.
        Returns (gnn_logits[B,2], center_scores[B,2])."""
        nodes = [int(n) for n in nodes]
        neigh = [[sorted(a[n]) for n in nodes] for a in self.adj]             # :217-219, :246-248
        uniq = sorted(set().union(*[set(l) for rel in neigh for l in rel], set(nodes)))  # :226-227
        pos_of = {n: i for i, n in enumerate(uniq)}                            # :240
        batch_scores = self._label_clf(self.X[torch.LongTensor(uniq)])         # :231-236
        pos_scores = self._label_clf(self.X[torch.LongTensor(self.train_pos)])  # :232-237
        center_scores = batch_scores[[pos_of[n] for n in nodes], :]            # :243
        self_feats = self.X[torch.LongTensor(nodes)]                           # :277
        lab = None if labels is None else [int(v) for v in labels]

        feats = [self_feats]
        self.last_sets = []
        for r in range(self.R):
            lists = neigh[r]
            nscore = [batch_scores[[pos_of[j] for j in l], 0].detach() for l in lists]   # :251-253, :650
            sets = choose_sets(center_scores[:, 0].detach(), lab, lists, nscore,
                               self.train_pos, pos_scores[:, 0].detach(),
                               self.thresholds[r], self.rho, train_flag)        # :587-591
            self.last_sets.append(sets)
            agg = (dense_mask_aggregate if self.dense_mask else sparse_aggregate)(sets, self.X)  # :594-624
            cat = torch.cat((self_feats, agg), dim=1)                           # :625
            feats.append(F.relu(cat.mm(self.p[f"inter1.intra_agg{r + 1}.weight"])))  # :629
        combined = F.relu(torch.cat(feats, dim=1).mm(self.p["inter1.weight"]).t())  # :284-289  [E,B]
        logits = self.p["weight"].mm(combined).t()                             # model.py:38-39
        self.last_feats = feats
        self.last_combined = combined
        return logits, center_scores

    def to_prob(self, nodes, labels, train_flag: bool = True):
        """model.py:41-45."""
        g, l = self.forward(nodes, labels, train_flag)
        return torch.sigmoid(g), torch.sigmoid(l)

    def loss(self, nodes, labels, train_flag: bool = True) -> torch.Tensor:
        """model.py:47-62."""
        y = torch.as_tensor(np.asarray(labels), dtype=torch.long)
        g, l = self.forward(nodes, y.tolist(), train_flag)
        return F.cross_entropy(g, y) + self.lambda_1 * F.cross_entropy(l, y)

    def parameters(self):
        return list(self.p.values())


def train_step(model: OraclePCGNN, opt: torch.optim.Optimizer, nodes, labels) -> float:
    """One iteration of the batch loop, model_handler.py:149-153."""
    opt.zero_grad()
    loss = model.loss(nodes, labels)
    loss.backward()
    opt.step()
    return loss.item()


def make_adam(model: OraclePCGNN, lr: float, weight_decay: float) -> torch.optim.Adam:
    """model_handler.py:124 (coupled L2 weight decay, torch defaults otherwise)."""
    return torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)


def run_epoch(model: OraclePCGNN, opt, sampled: Sequence[int], all_labels: np.ndarray, batch_size: int,
              timer=None) -> Tuple[int, float]:
    """Batch loop of one epoch incl. the reference's timing window
    (model_handler.py:134-156).  Unlike :134 it does not run the empty
    trailing batch the reference crashes on.  Returns (#nodes, seconds)."""
    import time
    timer = timer or time.perf_counter
    n = len(sampled)
    nb = (n + batch_size - 1) // batch_size
    spent = 0.0
    for b in range(nb):
        t0 = timer()
        batch = sampled[b * batch_size:min((b + 1) * batch_size, n)]          # :144-147
        lab = all_labels[np.array(batch)]                                      # :148
        train_step(model, opt, batch, lab)                                     # :149-153
        spent += timer() - t0
    return n, spent


# --------------------------------------------------------------------------
# S1  GraphSAGE / GCN aggregators  (src/graphsage.py)
# --------------------------------------------------------------------------
def sage_mean(nodes: Sequence[int], adj: AdjList, X: torch.Tensor, gcn: bool = False) -> torch.Tensor:
    """MeanAggregator.forward without fan-out sampling (graphsage.py:62-96);
    ``gcn=True`` unions the centre into its own set (:78-79)."""
    sets = [set(adj[int(n)]) | ({int(n)} if gcn else set()) for n in nodes]
    return dense_mask_aggregate(sets, X, "count")


def sage_mean_fanout(nodes: Sequence[int], to_neighs: Sequence[Set[int]], X: torch.Tensor, num_sample: int, rng,
                     gcn: bool = False) -> torch.Tensor:
    """MeanAggregator.forward with the random fan-out (graphsage.py:70-74): a row with at least ``num_sample``
    neighbours keeps ``set(random.sample(to_neigh, num_sample))``; ``random.sample`` on a set draws from
    ``tuple(to_neigh)`` (CPython 3.10 Lib/random.py), i.e. in the set's iteration order.  ``rng``: a ``random.Random``."""
    samp = [set(rng.sample(tuple(s), num_sample)) if len(s) >= num_sample else set(s) for s in to_neighs]
    if gcn:
        samp = [s | {int(nodes[i])} for i, s in enumerate(samp)]                # :78-79
    return dense_mask_aggregate(samp, X, "count")


def gcn_mean(nodes: Sequence[int], adj: AdjList, X: torch.Tensor) -> torch.Tensor:
    """GCNAggregator.forward: union self, divide by sqrt(row count) (graphsage.py:200-232)."""
    sets = [set(adj[int(n)]) | {int(n)} for n in nodes]
    return dense_mask_aggregate(sets, X, "sqrt_count")


def encoder_forward(agg: torch.Tensor, self_feats: Optional[torch.Tensor], W: torch.Tensor) -> torch.Tensor:
    """Encoder / GCNEncoder tail: relu(W . combined^T) -> [E,B] (graphsage.py:148-149, 274)."""
    comb = agg if self_feats is None else torch.cat((self_feats, agg), dim=1)
    return F.relu(W.mm(comb.t()))


# --------------------------------------------------------------------------
# helpers shared by tests / bench (not part of the reference)
# --------------------------------------------------------------------------
def adj_to_csr(adj: AdjList, n_nodes: int) -> Tuple[np.ndarray, np.ndarray]:
    """dict-of-sets -> (indptr int64 [N+1], indices int32 ascending per row)."""
    indptr = np.zeros(n_nodes + 1, dtype=np.int64)
    for v in range(n_nodes):
        indptr[v + 1] = indptr[v] + len(adj.get(v, ()))
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    for v in range(n_nodes):
        row = adj.get(v, ())
        if row:
            indices[indptr[v]:indptr[v + 1]] = sorted(row)
    return indptr, indices


def sets_to_csr(sets: Sequence[Set[int]]) -> Tuple[np.ndarray, np.ndarray]:
    off = np.zeros(len(sets) + 1, dtype=np.int64)
    for i, s in enumerate(sets):
        off[i + 1] = off[i] + len(s)
    flat = np.empty(int(off[-1]), dtype=np.int32)
    for i, s in enumerate(sets):
        flat[off[i]:off[i + 1]] = sorted(s)
    return off, flat
