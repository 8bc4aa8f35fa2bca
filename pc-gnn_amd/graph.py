"""Device-resident graph container: multi-relation CSR + frozen feature table.

Input contract = what the reference hands its layers:
``adj_lists``: list of ``dict[int -> set[int]]`` per relation, symmetric, with
self-loops (src/utils.py:226-239); ``features``: frozen ``nn.Embedding`` weight
``[N, F]`` (src/model_handler.py:85-87); ``train_pos``: list of ids.

Layout in HBM (see include/pcgnn.h): X ``[N, Fs]`` f32 with ``Fs = ceil4(F)``
zero-padded so every row is a whole number of 16-byte chunks (128-B rows for
YelpChi F=32 and for Amazon F=25), per relation ``indptr`` int64 ``[N+1]`` and
``indices`` int32 ascending inside a row, ``train_pos`` int32.
"""
import ctypes as C
from typing import Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch

from . import _lib

AdjList = Dict[int, Set[int]]


def adj_to_csr(adj: AdjList, n_nodes: int) -> Tuple[np.ndarray, np.ndarray]:
    """dict-of-sets -> (indptr int64 [N+1], indices int32, ascending per row)."""
    deg = np.zeros(n_nodes, dtype=np.int64)
    for v, s in adj.items():
        deg[int(v)] = len(s)
    indptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=indptr[1:])
    indices = np.empty(int(indptr[-1]), dtype=np.int32)
    for v, s in adj.items():
        if s:
            v = int(v)
            row = np.fromiter(s, dtype=np.int32, count=len(s))
            row.sort()
            indices[indptr[v]:indptr[v + 1]] = row
    return indptr, indices


def csr_from_pairs_device(n_nodes: int, src: torch.Tensor, dst: torch.Tensor, symmetrise: bool = True,
                          self_loops: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """CSR of an edge list, built ON THE DEVICE: what ``sparse_to_adjlist_for_train`` builds on the host with Python sets
    (src/utils.py:243-254: both directions, one self-loop per node, duplicates dropped) as one sort + unique of 64-bit
    (row, column) keys.  src / dst: integer device tensors of equal length.  Returns (indptr int64 [N + 1], indices int32
    ascending inside a row), both on the device of `src`."""
    dev = src.device
    a, b = src.to(torch.int64), dst.to(torch.int64)
    if a.numel() and (int(torch.min(torch.minimum(a, b))) < 0 or int(torch.max(torch.maximum(a, b))) >= n_nodes):
        raise ValueError("edge endpoint out of range")
    rows, cols = [a], [b]
    if symmetrise:
        rows.append(b)
        cols.append(a)
    if self_loops:
        loop = torch.arange(n_nodes, dtype=torch.int64, device=dev)
        rows.append(loop)
        cols.append(loop)
    key = torch.unique(torch.cat(rows) * n_nodes + torch.cat(cols))          # sorted: by row, then by column
    r = torch.div(key, n_nodes, rounding_mode="floor")
    indptr = torch.zeros(n_nodes + 1, dtype=torch.int64, device=dev)
    indptr[1:] = torch.cumsum(torch.bincount(r, minlength=n_nodes), 0)
    return indptr, (key - r * n_nodes).to(torch.int32)


def adj_to_pairs(adj: AdjList) -> Tuple[np.ndarray, np.ndarray]:
    """dict-of-sets -> (src, dst) int64 arrays (one pass over the dict; the sort / de-duplication happens on the device)."""
    n_entries = sum(len(s) for s in adj.values())
    src = np.empty(n_entries, dtype=np.int64)
    dst = np.empty(n_entries, dtype=np.int64)
    at = 0
    for v, s in adj.items():
        k = len(s)
        if k:
            src[at:at + k] = int(v)
            dst[at:at + k] = np.fromiter(s, dtype=np.int64, count=k)
            at += k
    return src, dst


class DeviceGraph:
    def __init__(self, X, csr: Sequence[Tuple[np.ndarray, np.ndarray]], train_pos: Sequence[int],
                 device: Optional[torch.device] = None, id_space: Optional[int] = None):
        """id_space: size of the id space neighbour / train_pos ids live in when it is not this
        table's own row count (a shard of a partitioned graph keeps GLOBAL ids, see dist.py)."""
        _lib.load()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.PcgnnLibraryError("DeviceGraph needs a GPU device: the PC-GNN hot path has no CPU fallback")
        X = torch.as_tensor(X, dtype=torch.float32)
        self.n_nodes, self.feat_dim = int(X.shape[0]), int(X.shape[1])
        self.id_space = int(id_space or X.shape[0])
        self.feat_stride = (self.feat_dim + 3) // 4 * 4
        if self.feat_stride > 512:
            raise _lib.PcgnnLibraryError(f"feat_dim {self.feat_dim} > 512 is not supported by the gather kernels")
        Xp = torch.zeros(self.n_nodes, self.feat_stride, dtype=torch.float32, device=self.device)
        Xp[:, :self.feat_dim] = X.to(self.device)
        self.X = Xp
        self.R = len(csr)
        if not 1 <= self.R <= _lib.PCG_MAX_REL:
            raise ValueError(f"1..{_lib.PCG_MAX_REL} relations supported, got {self.R}")
        self.indptr, self.indices, self.deg_host = [], [], []
        self.max_degree = 0
        for indptr, indices in csr:
            indptr = np.ascontiguousarray(indptr, dtype=np.int64)
            indices = np.ascontiguousarray(indices, dtype=np.int32)
            if indptr.shape[0] != self.n_nodes + 1 or indptr[-1] != indices.shape[0]:
                raise ValueError("CSR shape does not match the feature table")
            if indices.size and (indices.min() < 0 or indices.max() >= (id_space or self.n_nodes)):
                raise ValueError("neighbour id out of range")
            deg = np.diff(indptr)
            self.deg_host.append(deg)
            self.max_degree = max(self.max_degree, int(deg.max()) if deg.size else 0)
            self.indptr.append(torch.from_numpy(indptr).to(self.device))
            self.indices.append(torch.from_numpy(indices).to(self.device) if indices.size
                                else torch.zeros(1, dtype=torch.int32, device=self.device))
        tp = np.asarray(list(train_pos), dtype=np.int64)
        if tp.size and (tp.min() < 0 or tp.max() >= (id_space or self.n_nodes)):
            raise ValueError("train_pos id out of range")
        if np.unique(tp).size != tp.size:
            raise ValueError("train_pos contains duplicate ids (the reference builds it with pos_neg_split, "
                             "src/utils.py:256-271, which never does)")
        self.n_pos = int(tp.size)
        self.train_pos_host = tp
        self.train_pos = (torch.from_numpy(tp.astype(np.int32)).to(self.device) if tp.size
                          else torch.zeros(1, dtype=torch.int32, device=self.device))
        self._desc = None

    # -- constructors ---------------------------------------------------------
    @classmethod
    def from_adj_lists(cls, features_weight, adj_lists: Sequence[AdjList], train_pos, device=None, build_on_device: bool = True):
        """From the reference's input format (list of dict[int -> set[int]] incl. self-loops, src/utils.py:226-239).  The
        dicts are flattened in one host pass; sorting / de-duplicating the rows happens on the device
        (``csr_from_pairs_device``) unless ``build_on_device=False`` (per-row host sort, ``adj_to_csr``)."""
        n = int(features_weight.shape[0])
        X = features_weight.detach().cpu() if torch.is_tensor(features_weight) else features_weight
        if not build_on_device:
            return cls(X, [adj_to_csr(a, n) for a in adj_lists], train_pos, device)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if dev.type != "cuda":
            raise _lib.PcgnnLibraryError("DeviceGraph needs a GPU device: the PC-GNN hot path has no CPU fallback")
        csr = []
        for a in adj_lists:
            src, dst = adj_to_pairs(a)
            ip, ix = csr_from_pairs_device(n, torch.from_numpy(src).to(dev), torch.from_numpy(dst).to(dev),
                                           symmetrise=False, self_loops=False)       # the reference's dicts are complete already
            csr.append((ip.cpu().numpy(), ix.cpu().numpy()))
        return cls(X, csr, train_pos, dev)

    @classmethod
    def from_scipy(cls, features_weight, matrices, train_pos, device=None):
        """From scipy sparse adjacency matrices (what data_process.py reads from the .mat files): symmetrised, self-loops
        added, duplicates dropped on the device (src/utils.py:243-254 semantics)."""
        import scipy.sparse as sp
        n = int(features_weight.shape[0])
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        csr = []
        for m in matrices:
            coo = sp.coo_matrix(m)
            ip, ix = csr_from_pairs_device(n, torch.from_numpy(coo.row.astype(np.int64)).to(dev),
                                           torch.from_numpy(coo.col.astype(np.int64)).to(dev))
            csr.append((ip.cpu().numpy(), ix.cpu().numpy()))
        X = features_weight.detach().cpu() if torch.is_tensor(features_weight) else features_weight
        return cls(X, csr, train_pos, dev)

    # -- C ABI view -------------------------------------------------------------
    @property
    def desc(self) -> _lib.GraphDesc:
        if self._desc is None:
            d = _lib.GraphDesc()
            d.n_nodes, d.feat_dim, d.feat_stride = self.n_nodes, self.feat_dim, self.feat_stride
            d.n_rel, d.n_pos, d.max_degree = self.R, self.n_pos, self.max_degree
            d.X = self.X.data_ptr()
            d.train_pos = self.train_pos.data_ptr()
            for r in range(self.R):
                d.indptr[r] = self.indptr[r].data_ptr()
                d.indices[r] = self.indices[r].data_ptr()
            self._desc = d
        return self._desc

    def desc_ref(self):
        return C.byref(self.desc)

    def nbytes(self) -> int:
        return (self.X.numel() * 4 + sum(t.numel() * 8 for t in self.indptr) + sum(t.numel() * 4 for t in self.indices)
                + self.train_pos.numel() * 4)
