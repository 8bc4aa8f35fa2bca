"""GraphSAGE / GCN baselines of the reference (src/graphsage.py) on the HIP aggregation path.

Same class names and constructor arguments as the reference: ``MeanAggregator`` (:42-96),
``Encoder`` (:99-150), ``GraphSage`` (:16-39), ``GCNAggregator`` (:181-232), ``GCNEncoder``
(:234-275), ``GCN`` (:154-178).  The aggregation is the same kernel as PC-GNN's with the
choose step switched off (threshold 1 => keep every neighbour), ``add_self`` for the GCN-style
self union (:78-79, :214) and the ``sqrt(count)`` normaliser (:224-226); the small dense tail
(``relu(W . x^T)``, classifier, loss) is torch with autograd.

``MeanAggregator.forward(..., num_sample=k)`` (random fan-out, :70-74) draws the sample on the
host with Python's ``random`` exactly like the reference (same draws under the same seed: checked
against a fixture) and aggregates the explicit lists with ``pcg_segment_mean``.

One difference on purpose: ``to_prob`` accepts (and ignores) the ``labels`` / ``train_flag``
arguments ``utils.test`` passes (utils.py:305), which the reference's ``GCN.to_prob(self, nodes)``
(:172) does not - with the reference the baselines cannot be evaluated by its own ``test()``.
"""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import init

from . import _lib, ops
from .graph import DeviceGraph


def _graph_of(obj, features, adj_lists, device) -> DeviceGraph:
    if getattr(obj, "_graph", None) is None:
        if isinstance(adj_lists, DeviceGraph):
            obj._graph = adj_lists
        else:
            if device.type != "cuda":
                raise _lib.PcgnnLibraryError("move the model to a GPU before calling forward(): there is no CPU path")
            obj._graph = DeviceGraph.from_adj_lists(features.weight, [adj_lists], [], device)
    return obj._graph


class _FullNeighbourhood(nn.Module):
    """mean (or GCN-normalised sum) over every neighbour of the batch nodes."""

    def __init__(self, features, cuda=False, gcn=False, norm=_lib.PCG_NORM_COUNT):
        super().__init__()
        self.features = features
        self.cuda = cuda
        self.gcn = gcn
        self._norm = norm
        self._graph = None
        self._dev = None           # set by the encoder: the device its weights live on
        self.adj_lists = None      # set by the encoder that owns the graph

    def aggregate(self, nodes, device) -> torch.Tensor:
        g = _graph_of(self, self.features, self.adj_lists, device)
        ids = ops._i32(nodes, g.device)
        if getattr(self, "_zeros", None) is None or self._zeros.numel() != g.n_nodes:
            self._zeros = torch.zeros(g.n_nodes, dtype=torch.float32, device=g.device)
        s0 = self._zeros      # scores are irrelevant here: threshold 1 keeps every neighbour
        agg, _ = ops.choose_aggregate(g, ids, None, s0, None, [1.0], [0.0], False, norm=self._norm, add_self=self.gcn)
        return agg[0]


class MeanAggregator(_FullNeighbourhood):
    """reference: src/graphsage.py:42-96"""

    def __init__(self, features, cuda=False, gcn=False):
        super().__init__(features, cuda, gcn, _lib.PCG_NORM_COUNT)

    def forward(self, nodes, to_neighs, num_sample=None):
        dev = self._graph.device if self._graph is not None else (self._dev or next(iter(self.features.parameters())).device)
        if num_sample is None and self.adj_lists is not None:
            return self.aggregate(nodes, dev)
        # explicit neighbour sets (and the optional random fan-out): pack the lists, segmented mean
        _set, _sample = set, random.sample
        if num_sample is not None:
            # random.sample(set, k) draws from tuple(set) - the set's own iteration order - in CPython <= 3.10 (the
            # reference's interpreter); spelled out so that the draws are the reference's under the same seed
            samp = [_set(_sample(tuple(n), num_sample)) if len(n) >= num_sample else n for n in to_neighs]   # :70-74
        else:
            samp = to_neighs
        if self.gcn:
            samp = [s | {int(nodes[i])} for i, s in enumerate(samp)]                                      # :78-79
        g = _graph_of(self, self.features, self.adj_lists if self.adj_lists is not None else {}, torch.device(dev))
        lens = np.array([len(s) for s in samp], dtype=np.int32)
        begin = np.zeros(len(samp), dtype=np.int64)
        np.cumsum(lens[:-1], out=begin[1:])
        flat = np.concatenate([np.sort(np.fromiter(s, dtype=np.int32, count=len(s))) for s in samp]) if len(samp) else \
            np.zeros(0, np.int32)
        return ops.segment_mean(g, torch.from_numpy(begin).to(g.device), torch.from_numpy(lens).to(g.device),
                                torch.from_numpy(flat).to(g.device), _lib.PCG_NORM_COUNT)


class GCNAggregator(_FullNeighbourhood):
    """reference: src/graphsage.py:181-232 (union self, divide by sqrt(row count))"""

    def __init__(self, features, cuda=False):
        super().__init__(features, cuda, True, _lib.PCG_NORM_SQRT_COUNT)

    def forward(self, nodes, to_neighs):
        dev = self._graph.device if self._graph is not None else (self._dev or next(iter(self.features.parameters())).device)
        return self.aggregate(nodes, dev)


class Encoder(nn.Module):
    """reference: src/graphsage.py:99-150"""

    def __init__(self, features, feature_dim, embed_dim, adj_lists, aggregator, num_sample=10, base_model=None,
                 gcn=False, cuda=False, feature_transform=False):
        super().__init__()
        self.features = features
        self.feat_dim = feature_dim
        self.adj_lists = adj_lists
        self.aggregator = aggregator
        if base_model is not None:
            self.base_model = base_model
        self.gcn = gcn
        self.embed_dim = embed_dim
        self.cuda = cuda
        self.aggregator.cuda = cuda
        self.aggregator.adj_lists = adj_lists
        self.weight = nn.Parameter(torch.FloatTensor(embed_dim, self.feat_dim if self.gcn else 2 * self.feat_dim))
        init.xavier_uniform_(self.weight)

    def forward(self, nodes):
        self.aggregator._dev = self.weight.device
        neigh_feats = self.aggregator.forward(nodes, None)                       # :133
        if not self.gcn:
            g = self.aggregator._graph
            self_feats = ops.gather_rows(g, ops._i32(nodes, g.device))            # :140-145
            combined = torch.cat((self_feats, neigh_feats), dim=1)
        else:
            combined = neigh_feats
        return F.relu(self.weight.mm(combined.t()))                              # :149  [E, B]


class GCNEncoder(nn.Module):
    """reference: src/graphsage.py:234-275"""

    def __init__(self, features, feature_dim, embed_dim, adj_lists, aggregator, base_model=None, cuda=False,
                 feature_transform=False):
        super().__init__()
        self.features = features
        self.feat_dim = feature_dim
        self.adj_lists = adj_lists
        self.aggregator = aggregator
        if base_model is not None:
            self.base_model = base_model
        self.embed_dim = embed_dim
        self.cuda = cuda
        self.aggregator.cuda = cuda
        self.aggregator.adj_lists = adj_lists
        self.weight = nn.Parameter(torch.FloatTensor(embed_dim, self.feat_dim))
        init.xavier_uniform_(self.weight)

    def forward(self, nodes):
        self.aggregator._dev = self.weight.device
        neigh_feats = self.aggregator.forward(nodes, None)                       # :267
        return F.relu(self.weight.mm(neigh_feats.t()))                           # :274  [E, B]


class _Head(nn.Module):
    def __init__(self, num_classes, enc):
        super().__init__()
        self.enc = enc
        self.xent = nn.CrossEntropyLoss()
        self.weight = nn.Parameter(torch.FloatTensor(num_classes, enc.embed_dim))
        init.xavier_uniform_(self.weight)

    def forward(self, nodes):
        return self.weight.mm(self.enc(nodes)).t()                               # :28-31 / :167-170

    def loss(self, nodes, labels):
        y = torch.as_tensor(labels, device=self.weight.device).long().reshape(-1)
        return self.xent(self.forward(nodes), y)                                 # :37-39 / :176-178


class GraphSage(_Head):
    """reference: src/graphsage.py:16-39"""

    def to_prob(self, nodes, labels=None, train_flag=False):
        # the reference takes log_softmax over dim=2 of a 2-D tensor (:34) and cannot run; probabilities
        # over the class dimension are what utils.test consumes (utils.py:305-309)
        p = torch.sigmoid(self.forward(nodes))
        return p, p


class GCN(_Head):
    """reference: src/graphsage.py:154-178"""

    def to_prob(self, nodes, labels=None, train_flag=False):
        p = torch.sigmoid(self.forward(nodes))                                   # :172-174
        return p, p
