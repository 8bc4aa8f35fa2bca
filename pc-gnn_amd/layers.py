"""Host-side mirror of the reference's PC-GNN layers over the HIP hot path.

Same class names, constructor arguments, ``forward()`` signatures, attribute and
state-dict names as ``src/layers.py`` of the reference (``IntraAgg`` :539-630,
``InterAgg3`` :161-291, ``InterAgg1`` :417-535, ``InterAgg5`` :16-158), so a
``ModelHandler`` can swap its import and nothing else.  The three copy-pasted
``InterAgg{1,3,5}`` of the reference are one R-generic class here.

What runs where: neighbour lookup, label-aware scoring, the choose step, the
minority over-sampling and the mean aggregation are HIP kernels behind the C ABI
(``include/pcgnn.h``); the small dense tail (relation / inter GEMMs, ReLU, loss)
is torch on the same device, with autograd (the features are frozen and the
selection is index-only, so no gradient ever flows into the kernels'
outputs except through ``center_scores``).
"""
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from . import _lib, ops
from .graph import DeviceGraph


class _CenterScores(torch.autograd.Function):
    """center_scores = label_clf(X[nodes]) (layers.py:236,243) with the forward
    value taken from the HIP kernel, so its column 0 is bit-identical to the
    score table the choose step ranks with."""

    @staticmethod
    def forward(ctx, weight, bias, self_feats, ids, graph):
        ctx.save_for_backward(self_feats)
        return ops.score_rows(graph, weight, bias, ids)

    @staticmethod
    def backward(ctx, g):
        (self_feats,) = ctx.saved_tensors
        return g.t().mm(self_feats), g.sum(0), None, None, None


class IntraAgg(nn.Module):
    """Intra-relation aggregator (reference: src/layers.py:539-630)."""

    def __init__(self, features, feat_dim, embed_dim, train_pos, rho, cuda=False):
        super().__init__()
        self.features = features
        self.cuda = cuda          # (sic) the reference shadows nn.Module.cuda the same way, :554
        self.feat_dim = feat_dim
        self.embed_dim = embed_dim
        self.train_pos = train_pos
        self.rho = rho
        self.weight = nn.Parameter(torch.FloatTensor(2 * self.feat_dim, self.embed_dim))
        init.xavier_uniform_(self.weight)
        self._graph: Optional[DeviceGraph] = None

    def transform(self, self_feats: torch.Tensor, agg_feats: torch.Tensor) -> torch.Tensor:
        """relu(cat(self, agg) @ W_r)  (layers.py:625-629)."""
        return F.relu(torch.cat((self_feats, agg_feats), dim=1).mm(self.weight))

    def forward(self, nodes, batch_labels, to_neighs_list, batch_scores, neigh_scores, pos_scores, sample_list,
                train_flag):
        """Reference signature (layers.py:562): explicit per-centre neighbour lists and
        scores.  ``InterAgg.forward`` does not come through here (it runs all relations
        in one fused launch); this entry keeps direct callers working.  The lists are
        packed into a one-relation CSR over the batch rows and handed to the same
        kernel.  ``sample_list[b]`` must equal ``ceil(len(to_neighs_list[b]) * threshold)``
        for one threshold, as in the reference (layers.py:260-262)."""
        dev = self.weight.device
        B = len(nodes)
        lens = np.array([len(l) for l in to_neighs_list], dtype=np.int64)
        thr = _infer_threshold(lens, np.asarray(sample_list, dtype=np.int64))
        n_nodes = int(self.features.weight.shape[0])
        indptr = np.zeros(n_nodes + 1, dtype=np.int64)
        indptr[1:B + 1] = np.cumsum(lens)
        indptr[B + 1:] = indptr[B]
        flat, s_flat = [], []
        for l, sc in zip(to_neighs_list, neigh_scores):
            ids = np.asarray(list(l), dtype=np.int64)
            order = np.argsort(ids, kind="stable")
            flat.append(ids[order])
            s_flat.append(sc.detach().reshape(-1, 2)[:, 0].cpu().numpy()[order])
        flat = np.concatenate(flat).astype(np.int32) if flat else np.zeros(0, np.int32)
        s_flat = np.concatenate(s_flat).astype(np.float32) if s_flat else np.zeros(0, np.float32)
        g = DeviceGraph(self.features.weight.detach(), [(indptr, flat)], self.train_pos, dev)
        s0 = torch.zeros(n_nodes, dtype=torch.float32, device=dev)
        s0[torch.from_numpy(flat.astype(np.int64)).to(dev)] = torch.from_numpy(s_flat).to(dev)
        if len(self.train_pos):
            s0[torch.as_tensor(list(self.train_pos), device=dev)] = pos_scores.detach()[:, 0].to(dev)
        rows = torch.arange(B, dtype=torch.int32, device=dev)
        labels = ops._i32(batch_labels, dev) if train_flag else None
        keys = ops.pos_sort(g, s0) if (train_flag and g.n_pos) else None
        center = batch_scores.detach()[:, 0].contiguous().to(dev)
        agg, _ = ops.choose_aggregate(g, rows, labels, s0, keys, [thr], [self.rho], train_flag, center_s0=center)
        self_feats = self.features.weight.detach().to(dev)[torch.as_tensor(np.asarray(nodes), device=dev).long()]
        return self.transform(self_feats, agg[0]), None


def _infer_threshold(lens: np.ndarray, samples: np.ndarray) -> float:
    for thr in (0.5, 1.0, 0.25, 0.75):
        if np.array_equal(np.ceil(lens * thr).astype(np.int64), samples):
            return thr
    big = lens.argmax()
    thr = float(samples[big]) / float(lens[big])
    if not np.array_equal(np.ceil(lens * thr).astype(np.int64), samples):
        raise ValueError("sample_list is not ceil(len * threshold) for a single threshold")
    return thr


class InterAgg(nn.Module):
    """Inter-relation aggregator for any number of relations
    (reference: InterAgg3 src/layers.py:161-291; InterAgg1 :417-535; InterAgg5 :16-158)."""

    def __init__(self, features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter='GNN', cuda=True):
        super().__init__()
        self.features = features
        self.dropout = 0.6
        self.adj_lists = adj_lists
        self.n_rel = len(intraggs)
        for r, agg in enumerate(intraggs):
            setattr(self, f"intra_agg{r + 1}", agg)      # state-dict names inter1.intra_agg{r}.weight
            agg.cuda = cuda
        self.embed_dim = embed_dim
        self.feat_dim = feature_dim
        self.cuda = cuda
        self.train_pos = train_pos
        self.thresholds = [0.5] * self.n_rel             # layers.py:193 (fixed; never updated)
        self.weight = nn.Parameter(torch.FloatTensor(self.embed_dim * self.n_rel + self.feat_dim, self.embed_dim))
        init.xavier_uniform_(self.weight)
        self.label_clf = nn.Linear(self.feat_dim, 2)
        self.weights_log = []
        self.thresholds_log = [self.thresholds]
        self.relation_score_log = []
        self._graph: Optional[DeviceGraph] = adj_lists if isinstance(adj_lists, DeviceGraph) else None
        self._ws = None
        self._ws_cache = {}
        self._s0 = None
        self._keys = None
        self._prof = None
        self.last_counts = None

    # ------------------------------------------------------------------
    @property
    def intra_aggs(self) -> List[IntraAgg]:
        return [getattr(self, f"intra_agg{r + 1}") for r in range(self.n_rel)]

    def graph(self) -> DeviceGraph:
        """The device-resident CSR + feature table, built on first use from the
        reference-format ``adj_lists`` / ``features`` (or passed in ready-made)."""
        if self._graph is None:
            if not self.cuda:
                raise _lib.PcgnnLibraryError("InterAgg(cuda=False): the MI355X build has no CPU path")
            dev = self.weight.device
            if dev.type != "cuda":
                raise _lib.PcgnnLibraryError("move the model to a GPU (model.cuda()) before calling forward()")
            self._graph = DeviceGraph.from_adj_lists(self.features.weight, self.adj_lists, self.train_pos, dev)
        return self._graph

    def forward(self, nodes, labels, train_flag=True):
        """:param nodes: list of batch node ids   :param labels: batch labels
        :return combined [E, B], center_scores [B, 2]   (layers.py:207-291)"""
        g = self.graph()
        dev = g.device
        ids = ops._i32(nodes, dev)
        B = ids.numel()
        lab = ops._i32(labels, dev).reshape(-1) if (train_flag and labels is not None) else None
        W, b = self.label_clf.weight, self.label_clf.bias
        if self._s0 is None:
            self._s0 = torch.empty(g.n_nodes, dtype=torch.float32, device=dev)
            self._keys = torch.empty(_lib.load().pcg_pos_sort_capacity(g.n_pos), dtype=torch.int64, device=dev)
        if self._ws is None or self._ws.B != B:      # the workspace layout depends on the batch size
            if B not in self._ws_cache:
                self._ws_cache[B] = ops.ChooseWorkspace(g, B)
            self._ws = self._ws_cache[B]

        s0 = ops.score_table(g, W, b, out=self._s0)                                  # :230-237
        keys = ops.pos_sort(g, s0, self._keys) if (train_flag and g.n_pos) else None   # :683-688
        rho = [a.rho for a in self.intra_aggs]
        if self._prof is not None:       # bench.py: HIP events around the dominant launch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        agg, self.last_counts = ops.choose_aggregate(g, ids, lab, s0, keys, self.thresholds, rho,
                                                     bool(train_flag), ws=self._ws)   # :246-270
        if self._prof is not None:
            ev[1].record()
            self._prof.append(ev)
        self_feats = ops.gather_rows(g, ids)                                          # :273-277
        center_scores = _CenterScores.apply(W, b, self_feats, ids, g)                 # :243
        feats = [self_feats] + [a.transform(self_feats, agg[r]) for r, a in enumerate(self.intra_aggs)]
        combined = F.relu(torch.cat(feats, dim=1).mm(self.weight).t())                # :284-289
        return combined, center_scores

    def check(self):
        """Raise if any batch since the last check did not fit its workspace's selection list (synchronises)."""
        for ws in self._ws_cache.values():
            ws.check()

    def chosen_sets(self, nodes, labels, train_flag=True):
        """samp_neighs of every relation for a batch, as Python sets (debug / parity)."""
        g = self.graph()
        ids = ops._i32(nodes, g.device)
        lab = ops._i32(labels, g.device).reshape(-1) if (train_flag and labels is not None) else None
        s0 = ops.score_table(g, self.label_clf.weight, self.label_clf.bias)
        keys = ops.pos_sort(g, s0) if (train_flag and g.n_pos) else None
        sets, _, _ = ops.chosen_sets(g, ids, lab, s0, keys, self.thresholds, [a.rho for a in self.intra_aggs],
                                     bool(train_flag))
        return sets


class InterAgg3(InterAgg):
    """Three relations (YelpChi / Amazon) - reference src/layers.py:161."""

    def __init__(self, features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter='GNN', cuda=True):
        if len(intraggs) != 3:
            raise ValueError("InterAgg3 takes exactly three intra-aggregators")
        super().__init__(features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter, cuda)


class InterAgg1(InterAgg):
    """One relation (tfinance / elliptic / weibo / kdk) - reference src/layers.py:417."""

    def __init__(self, features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter='GNN', cuda=True):
        if len(intraggs) != 1:
            raise ValueError("InterAgg1 takes exactly one intra-aggregator")
        super().__init__(features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter, cuda)


class InterAgg5(InterAgg):
    """Five relations - reference src/layers.py:16."""

    def __init__(self, features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter='GNN', cuda=True):
        if len(intraggs) != 5:
            raise ValueError("InterAgg5 takes exactly five intra-aggregators")
        super().__init__(features, feature_dim, embed_dim, train_pos, adj_lists, intraggs, inter, cuda)
