"""Multi-GPU PC-GNN: destination-node partition, RCCL all-to-all for remote neighbour rows.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  The
reference has no distributed code at all; this is new design:

* **Partition.**  Contiguous id ranges balanced by in-edge count (``Partition.balanced``: the CSR
  entries of all relations a rank's rows hold; equal ranges when no degrees are given).  The owner
  holds ``X[v]``, the CSR rows of ``v`` for every relation (neighbour ids stay GLOBAL) and ``v``'s
  label.  Replicated on every rank: all parameters, the train-pos ids and their feature rows (so
  minority over-sampling never needs a fetch).
* **Per step** (each rank works on centres it owns):
    1. class-0 scores of the owned rows (``pcg_step_front_a``) -> ``all_gather`` -> every rank has
       ``s0[N]``  (4 N bytes; 0.18 MB for YelpChi, 40 MB at 10 M nodes);
    2. train-pos sort + choose (``pcg_choose_select_planned``) on local rows -> selection lists of
       global ids;
    3. halo exchange: the list's entries are classified, the remote ids de-duplicated in a hash table
       sized by the halo capacity (``pcg_halo_classify`` - work and memory proportional to the list,
       not to the node count), the per-owner request counts of every rank all-gathered (the step's
       one host synchronisation: it sizes both all-to-alls, and every rank derives the SAME overflow
       verdict from it), ``all_to_all`` #1 sends the ids, the owners gather those rows,
       ``all_to_all`` #2 returns them into the halo region of the extended feature table; the lists
       are re-indexed into that table (``pcg_halo_remap``);
    4. ``pcg_aggregate_lists`` over the extended table, ``pcg_dense_step`` on the local batch
       with the loss scaled by 1 / global batch, gradient ``all_reduce`` (~107 KB), identical
       Adam on every rank.
  xGMI is point-to-point, so the all-to-all uses all 7 links of a GPU at once; the gradient
  all-reduce is latency-bound at this size.
* **Memory per rank**: owned rows + train-pos rows + a halo region sized from the batch's expected
  demand (``halo_rows``: pick-weighted mean list length x batch x a margin, never more than the remote
  nodes there are) - not from the node count; no per-node flag / slot / position tables.

The exchange layer (`HaloExchange`) is plain ``torch`` + ``torch.distributed`` and therefore also
runs on CPU tensors over ``gloo`` - that is what the world-size-2 CPU tests drive.  The kernels
themselves have no CPU path.
"""
import math
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist


class Partition:
    """Contiguous id ranges: rank r owns [bounds[r], bounds[r + 1])."""

    def __init__(self, n_nodes: int, world: int, rank: int, bounds: Optional[Sequence[int]] = None):
        self.n_nodes, self.world, self.rank = n_nodes, world, rank
        if bounds is None:                      # equal ranges
            n_per = (n_nodes + world - 1) // world
            bounds = [min(r * n_per, n_nodes) for r in range(world)] + [n_nodes]
        self.bounds_host = np.asarray(bounds, dtype=np.int64)
        assert self.bounds_host.shape[0] == world + 1 and self.bounds_host[0] == 0 and self.bounds_host[-1] == n_nodes
        assert np.all(np.diff(self.bounds_host) >= 0)
        self.lo, self.hi = int(self.bounds_host[rank]), int(self.bounds_host[rank + 1])
        self.n_local = self.hi - self.lo
        self.n_max = int(np.diff(self.bounds_host).max())       # longest shard (all-gather buffers are padded to it)
        self.n_per = self.n_max

    @classmethod
    def balanced(cls, degree: np.ndarray, world: int, rank: int) -> "Partition":
        """Ranges with (nearly) equal sums of `degree` (the per-node CSR entry count over all relations: what a
        rank's select / gather work and CSR memory are proportional to)."""
        n = int(degree.shape[0])
        cum = np.cumsum(np.asarray(degree, dtype=np.int64) + 1)            # (+1: a node costs something even without edges)
        if n and world > 1:
            cuts = np.searchsorted(cum, cum[-1] * np.arange(1, world) / world, side="left") + 1
        else:
            cuts = np.zeros(world - 1, np.int64)
        bounds = np.concatenate([[0], np.minimum(cuts, n), [n]]).astype(np.int64)
        bounds = np.maximum.accumulate(bounds)
        return cls(n, world, rank, bounds)

    def owner(self, ids):
        """owning rank of every id (numpy array or torch tensor)."""
        if torch.is_tensor(ids):
            b = torch.as_tensor(self.bounds_host[1:-1], device=ids.device)
            return torch.searchsorted(b, ids, right=True)
        return np.searchsorted(self.bounds_host[1:-1], ids, side="right")

    def bounds(self, device=None) -> torch.Tensor:
        return torch.as_tensor(self.bounds_host, device=device)


def total_degree(csr) -> np.ndarray:
    return sum(np.diff(ip) for ip, _ in csr)


def shard_workload(w, part: Partition):
    """The pieces of a synth.Workload rank `part.rank` holds: owned feature rows, owned CSR rows
    (global neighbour ids), owned labels / training nodes; plus the replicated train-pos rows."""
    lo, hi = part.lo, part.hi
    csr = []
    for indptr, idx in w.csr:
        a, b = int(indptr[lo]), int(indptr[hi])
        csr.append((indptr[lo:hi + 1] - indptr[lo], idx[a:b]))
    tr = w.idx_train[(w.idx_train >= lo) & (w.idx_train < hi)]
    return dict(X_local=w.X[lo:hi], csr=csr, labels_local=w.labels[lo:hi], idx_train_local=tr,
                homo_deg_train=w.homo_deg[tr], train_pos=list(w.train_pos), X_pos=w.X[np.array(w.train_pos, dtype=np.int64)]
                if len(w.train_pos) else np.zeros((0, w.X.shape[1]), np.float32))


def shard_pick_weights(labels_train_local: np.ndarray, homo_deg_train_local: np.ndarray, n_train_global: int,
                       n_train_pos_global: int) -> np.ndarray:
    """Pick weights deg / LF of a rank's training nodes with the GLOBAL label frequencies (utils.py:276: LF = #train
    positives for a positive, n_train for a negative) - so that the ranks' draws together follow the single-GPU sampler's
    distribution (every rank draws its share of the epoch from its own nodes)."""
    lf = np.where(np.asarray(labels_train_local) == 1, n_train_pos_global, n_train_global)
    return np.asarray(homo_deg_train_local, dtype=np.int64) / lf


def expected_halo_rows(deg_rel_train_local: Sequence[np.ndarray], weights: np.ndarray, batch: int, world: int, n_remote: int,
                       margin: float = 1.5) -> int:
    """Halo capacity from the batch's expected demand: a picked centre has, per relation, ~ceil(deg / 2) chosen
    neighbours (its minority picks are local: the train-pos block); a fraction (world - 1) / world of them is remote;
    distinct ones are at most that many.  Pick-weighted mean over the rank's training nodes x batch x margin, never more
    than the remote nodes there are."""
    if world == 1 or n_remote <= 0:
        return 1
    if weights.size == 0 or weights.sum() <= 0:
        return min(n_remote, 1024)
    p = weights / weights.sum()
    per_centre = sum(float((np.ceil(d / 2.0) * p).sum()) for d in deg_rel_train_local)
    est = per_centre * batch * (world - 1) / world * margin + 1024
    return int(min(n_remote, math.ceil(est)))


class HaloExchange:
    """Fetch the feature rows of remote ids and re-index a selection list into the extended table

        X_ext = [ owned rows (n_local) | train-pos rows (P) | halo (per step) ]

    `lst` holds global ids (-1 = hole).  After `fetch_and_remap(lst)`, `lst` holds row numbers of
    X_ext and the halo region holds the rows fetched this step.  Works on any device / backend;
    `stage_host=True` stages the collectives through CPU tensors (gloo with device tensors).
    Work and memory are proportional to the list (sort-unique of its remote entries, binary search in the
    sorted train-pos ids), not to the node count.
    """

    def __init__(self, part: Partition, X_ext: torch.Tensor, train_pos: Sequence[int], group=None, stage_host: bool = False):
        self.part, self.X_ext, self.P = part, X_ext, len(train_pos)
        self.group, self.stage_host = group, stage_host
        tp = torch.as_tensor(np.asarray(list(train_pos), dtype=np.int64), device=X_ext.device)
        self.pos_ids, self.pos_idx = torch.sort(tp) if self.P else (tp, tp)
        self.halo_base = part.n_local + self.P
        self.halo_cap = X_ext.shape[0] - self.halo_base
        self._all_caps = None
        self.last_stats = {}

    # -- collectives (optionally staged through the host) -------------------------------------
    def _a2a(self, out: torch.Tensor, inp: torch.Tensor, out_splits: Optional[List[int]], in_splits: Optional[List[int]]):
        if self.stage_host and inp.device.type != "cpu":
            o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(o, i, out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)

    def _gather_counts(self, out, inp):
        if self.stage_host and inp.device.type != "cpu":
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp.contiguous(), group=self.group)

    def _share_caps(self, halo_cap: int, serve_cap: int):
        """every rank's capacities on every rank (one collective at construction), so that every rank can judge every rank"""
        w = self.part.world
        caps = torch.tensor([halo_cap, serve_cap], dtype=torch.int64, device=self.X_ext.device)
        allc = torch.empty(2 * w, dtype=torch.int64, device=self.X_ext.device)
        self._gather_counts(allc, caps)
        self._all_caps = allc.view(w, 2).cpu().tolist()

    def _check(self, mat: torch.Tensor, flags: Optional[torch.Tensor] = None):
        """Every rank holds the same count matrix (and the same capacities), so every rank reaches the same verdict:
        all raise, or none does - a rank raising on its own would leave the others blocked in the next collective."""
        need = mat.sum(1)                       # rows each rank fetches
        serve = mat.sum(0)                      # rows each rank serves
        bad = []
        for r in range(self.part.world):
            cap_r, srv_r = self._all_caps[r] if self._all_caps is not None else (self.halo_cap, None)
            if int(need[r]) > int(cap_r):
                bad.append(f"rank {r} needs {int(need[r])} halo rows, capacity {int(cap_r)}")
            if srv_r is not None and int(serve[r]) > int(srv_r):
                bad.append(f"rank {r} has to serve {int(serve[r])} rows, capacity {int(srv_r)}")
            if flags is not None and int(flags[r]):
                bad.append(f"rank {r}: request table full (flags {int(flags[r])})")
        if bad:
            raise RuntimeError("halo exchange over capacity - raise halo_rows / serve_rows: " + "; ".join(bad))

    def fetch_and_remap(self, lst: torch.Tensor) -> int:
        part, dev = self.part, lst.device
        w = part.world
        if self._all_caps is None:
            self._share_caps(self.halo_cap, (w - 1) * part.n_local)
        ids = lst.long()
        valid = ids >= 0
        is_local = valid & (ids >= part.lo) & (ids < part.hi)
        if self.P:
            at = torch.searchsorted(self.pos_ids, ids.clamp(min=0)).clamp(max=self.P - 1)
            is_pos = valid & ~is_local & (self.pos_ids[at] == ids)
            pm = self.pos_idx[at]
        else:
            is_pos, pm = torch.zeros_like(valid), torch.zeros_like(ids)
        is_rem = valid & ~is_local & ~is_pos
        uniq, inv = torch.unique(ids[is_rem], return_inverse=True)         # sorted ascending = grouped by owner
        cuts = torch.searchsorted(uniq, part.bounds(dev))
        send_counts = (cuts[1:] - cuts[:-1]).to(torch.int64)
        mat = torch.empty(w * w, dtype=torch.int64, device=dev)
        self._gather_counts(mat, send_counts)
        mat = mat.view(w, w).cpu()                                          # host needs the split sizes
        self._check(mat)
        sc, rc = mat[part.rank].tolist(), mat[:, part.rank].tolist()
        n_halo = int(uniq.numel())
        req = torch.empty(sum(rc), dtype=torch.int64, device=dev)
        self._a2a(req, uniq, rc, sc)                                        # all-to-all #1: requested ids
        rows = self.X_ext.index_select(0, req - part.lo)                    # the owner gathers its rows
        halo = self.X_ext[self.halo_base:self.halo_base + n_halo]
        self._a2a(halo, rows, sc, rc)                                       # all-to-all #2: feature rows
        new = torch.where(is_local, ids - part.lo, torch.where(is_pos, part.n_local + pm, ids))
        new[is_rem] = self.halo_base + inv
        lst.copy_(new.to(lst.dtype))
        self.last_stats = {"entries": int(valid.sum()), "remote_entries": int(is_rem.sum()), "halo_rows": n_halo,
                           "bytes_in": n_halo * self.X_ext.shape[1] * 4, "bytes_out": sum(rc) * self.X_ext.shape[1] * 4}
        return n_halo


class HaloExchangeHip(HaloExchange):
    """Same exchange with the list work done by HIP kernels (pcg_halo_classify / pcg_halo_remap: hash-table
    de-duplication of the list's remote entries) and ONE host synchronisation per step: the per-owner request counts
    (+ a table-overflow word) of every rank are all-gathered as a world x (world + 1) matrix, which gives a rank its send
    and receive split sizes and gives every rank the same overflow verdict."""

    def __init__(self, part, X_ext, train_pos, serve_cap: int, group=None, stage_host=False):
        super().__init__(part, X_ext, train_pos, group, stage_host)
        from . import _lib, ops
        self._lib, self._ops = _lib, ops
        dev = X_ext.device
        lib = _lib.load()
        w = part.world
        self.slots = int(lib.pcg_halo_table_slots(self.halo_cap))
        self.table = torch.empty(2 * self.slots, dtype=torch.int32, device=dev)
        self.counts = torch.zeros(129, dtype=torch.int32, device=dev)
        self.uniq = torch.empty(max(self.halo_cap, 1), dtype=torch.int32, device=dev)
        self.bounds_dev = part.bounds(dev).to(torch.int32)
        self.pos_ids32, self.pos_idx32 = self.pos_ids.to(torch.int32), self.pos_idx.to(torch.int32)
        self.serve_cap = max(1, int(serve_cap))
        self._req = torch.empty(self.serve_cap, dtype=torch.int32, device=dev)
        self._rows = torch.zeros(self.serve_cap, X_ext.shape[1], dtype=torch.float32, device=dev)
        self._vec = torch.zeros(w + 1, dtype=torch.int32, device=dev)
        self._mat = torch.zeros(w * (w + 1), dtype=torch.int32, device=dev)
        self._share_caps(self.halo_cap, self.serve_cap)
        self.max_seen = {"halo_rows": 0, "rows_served": 0}

    def fetch_and_remap_device(self, ws, B: int, graph) -> int:
        """ws: the step's ChooseWorkspace (its list holds global ids, its chunk table says which entries are in use)."""
        lib, ops, part = self._lib.load(), self._ops, self.part
        _p, st = ops._p, ops._stream(self.X_ext.device)
        w, rank = part.world, part.rank
        self.table[:self.slots].fill_(-1)
        self.counts.zero_()
        self._lib.check(lib.pcg_halo_classify(
            graph.desc_ref(), B, _p(ws.buf), ws.list_capacity, part.lo, part.hi, part.n_local, _p(self.pos_ids32),
            _p(self.pos_idx32), self.P, _p(self.bounds_dev), w, _p(self.table), self.slots, _p(self.counts), _p(self.uniq),
            self.halo_cap, self.halo_base, st), "pcg_halo_classify")
        self._vec[:w].copy_(self.counts[:w])
        self._vec[w:].copy_(self.counts[128:129])
        self._gather_counts(self._mat, self._vec)
        full = self._mat.view(w, w + 1).cpu()                              # the step's single host sync
        mat, flags = full[:, :w].long(), full[:, w]
        self._check(mat, flags=flags)
        sc, rc = mat[rank].tolist(), mat[:, rank].tolist()
        n_halo, n_req = sum(sc), sum(rc)
        req = self._req[:n_req]
        self._a2a(req, self.uniq[:n_halo], rc, sc)                         # all-to-all #1: requested ids
        rows = self._rows[:n_req]
        if n_req:
            ops.gather_rows(graph, req - part.lo, out=rows)                # the owner gathers its rows (pad columns stay 0)
        halo = self.X_ext[self.halo_base:self.halo_base + n_halo]
        self._a2a(halo, rows, sc, rc)                                      # all-to-all #2: feature rows
        self._lib.check(lib.pcg_halo_remap(graph.desc_ref(), B, _p(ws.buf), ws.list_capacity, _p(self.table), self.slots,
                                           self.halo_cap, self.halo_base, st), "pcg_halo_remap")
        self.last_stats = {"halo_rows": n_halo, "rows_served": n_req, "bytes_in": n_halo * self.X_ext.shape[1] * 4,
                           "bytes_out": n_req * self.X_ext.shape[1] * 4}
        self.max_seen["halo_rows"] = max(self.max_seen["halo_rows"], n_halo)
        self.max_seen["rows_served"] = max(self.max_seen["rows_served"], n_req)
        return n_halo


class DistributedPCGNN:
    """The step driver of one rank of a node-partitioned run (HIP kernels + RCCL)."""

    def __init__(self, w, model_cfg: dict, device, group=None, stage_host: bool = False, halo_rows: Optional[int] = None,
                 serve_rows: Optional[int] = None, balanced: bool = True):
        from . import _lib, ops
        from .graph import DeviceGraph
        from .sampler import PickSampler
        self.ops, self.lib = ops, _lib.load()
        self._libmod = _lib
        self.group, self.stage_host = group, stage_host
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dev = torch.device(device)
        cfg = dict(emb_size=64, rho=0.5, alpha=2.0, lr=0.01, weight_decay=0.001, batch_size=1024, seed=0)
        cfg.update(model_cfg or {})
        self.cfg = cfg
        if hasattr(w, "X_local"):                   # a synth.ShardedWorkload: this rank's shard, generated in place (no rank ever held the whole graph)
            assert w.rank == self.rank and len(w.bounds) == self.world + 1
            part = self.part = Partition(w.n, self.world, self.rank, w.bounds)
            sh = dict(X_local=w.X_local, csr=w.csr, labels_local=w.labels_local, idx_train_local=w.idx_train_local,
                      homo_deg_train=w.homo_deg_train, train_pos=list(w.train_pos), X_pos=w.X_pos)
            n_train, n_train_pos, y_loc = w.n_train, w.n_train_pos, w.labels_train_local
            F = w.X_local.shape[1]
        else:
            if balanced:
                part = self.part = Partition.balanced(total_degree(w.csr), self.world, self.rank)
            else:
                part = self.part = Partition(w.n, self.world, self.rank)
            sh = shard_workload(w, part)
            y_all = w.labels[w.idx_train]
            n_train, n_train_pos, y_loc = len(y_all), int(y_all.sum()), w.labels[sh["idx_train_local"]]
            F = w.X.shape[1]
        P = len(sh["train_pos"])
        B = cfg["batch_size"]
        n_local = part.n_local
        # pick weights with the GLOBAL label frequencies (utils.py:276), not this shard's
        weights = shard_pick_weights(y_loc, sh["homo_deg_train"], n_train, n_train_pos)
        n_remote = max(w.n - n_local, 0)
        if halo_rows is None:       # from the batch's expected demand, not from the node count
            deg_rel = [np.diff(ip)[sh["idx_train_local"] - part.lo] for ip, _ in sh["csr"]]
            halo_rows = expected_halo_rows(deg_rel, weights, B, self.world, n_remote)
        halo_rows = max(int(halo_rows), 1)
        if serve_rows is None:      # what the others may ask of this rank: about what it asks of them; at most every owned row once per rank
            serve_rows = min((self.world - 1) * n_local, 2 * halo_rows + 1024) if self.world > 1 else 1
        serve_rows = max(int(serve_rows), 1)
        n_ext = n_local + P + halo_rows
        X_ext = np.zeros((n_ext, F), np.float32)
        X_ext[:n_local] = sh["X_local"]
        X_ext[n_local:n_local + P] = sh["X_pos"]
        csr_ext = []
        for indptr, idx in sh["csr"]:
            ip = np.concatenate([indptr, np.full(n_ext - n_local, indptr[-1], dtype=np.int64)])
            csr_ext.append((ip, idx))
        self.g = DeviceGraph(X_ext, csr_ext, sh["train_pos"], self.dev, id_space=w.n)
        g = self.g
        self.halo = HaloExchangeHip(part, g.X, sh["train_pos"], serve_rows, group, stage_host)
        self.labels_local = torch.from_numpy(sh["labels_local"].astype(np.int32)).to(self.dev)
        self.feature_rows = {"owned": n_local, "train_pos": P, "halo": halo_rows, "serve_buffer": serve_rows,
                             "unpartitioned_table": int(w.n)}

        # parameters: identical on every rank (same seed), flat buffer as in fused.py
        self.E, self.R, self.F = cfg["emb_size"], g.R, F
        n = int(self.lib.pcg_dense_n_params(F, self.E, self.R))
        self.n_params = n
        gen = torch.Generator().manual_seed(cfg["seed"])
        theta = torch.zeros(n)
        for which, rel, shape in ([(0, 0, (2, self.E)), (1, 0, (F + self.R * self.E, self.E))]
                                  + [(2, r, (2 * F, self.E)) for r in range(self.R)]):
            off = self.lib.pcg_dense_param_offset(F, self.E, self.R, which, rel)
            bound = math.sqrt(6.0 / (shape[0] + shape[1]))                  # xavier_uniform_ (model.py:30, layers.py:197,560)
            theta[off:off + shape[0] * shape[1]] = (torch.rand(shape[0] * shape[1], generator=gen) * 2 - 1) * bound
        # label_clf is an nn.Linear (layers.py:200): torch's default init, U(+-1/sqrt(fan_in)) for weight and bias
        for which, count in ((3, 2 * F), (4, 2)):
            off = self.lib.pcg_dense_param_offset(F, self.E, self.R, which, 0)
            theta[off:off + count] = (torch.rand(count, generator=gen) * 2 - 1) / math.sqrt(F)
        self.theta = theta.to(self.dev)
        self.m, self.v = torch.zeros_like(self.theta), torch.zeros_like(self.theta)
        self.grad = torch.zeros_like(self.theta)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        o3, o4 = (self.lib.pcg_dense_param_offset(F, self.E, self.R, wch, 0) for wch in (3, 4))
        self.w_clf = self.theta[o3:o3 + 2 * F].view(2, F)
        self.b_clf = self.theta[o4:o4 + 2]

        # score all-gather: shards padded to the longest one, then copied to their places in s0_full
        self.s0_send = torch.zeros(part.n_max, dtype=torch.float32, device=self.dev)
        self.s0_pad = torch.zeros(part.n_max * self.world, dtype=torch.float32, device=self.dev)
        self.s0_full = torch.zeros(w.n, dtype=torch.float32, device=self.dev)
        self.keys = torch.empty(self.lib.pcg_pos_sort_capacity(P), dtype=torch.int64, device=self.dev)
        self.ws = ops.ChooseWorkspace(g, B)
        self.cnt = torch.empty(g.R * B, dtype=torch.int32, device=self.dev)
        self.agg = torch.empty(g.R, B, F, dtype=torch.float32, device=self.dev)
        self.logits = torch.empty(B, 2, dtype=torch.float32, device=self.dev)
        self.center = torch.empty(B, 2, dtype=torch.float32, device=self.dev)
        self.row_loss = torch.zeros(B, dtype=torch.float32, device=self.dev)
        self.slabs = torch.empty(self.lib.pcg_dense_n_tiles(B), n, dtype=torch.float32, device=self.dev)
        self.thresholds, self.rho = [0.5] * g.R, [cfg["rho"]] * g.R
        self.sampler = PickSampler(sh["idx_train_local"] - part.lo, y_loc, sh["homo_deg_train"],
                                   self.dev, seed=cfg["seed"] + 7919 * self.rank)
        self.sampler.weights = weights
        self.sampler.cum_host = np.cumsum(weights)
        self.sampler.cum = torch.from_numpy(self.sampler.cum_host).to(self.dev)
        self.B = B
        self.ids_buf = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.lab_buf = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.center_buf = torch.zeros(B, dtype=torch.float32, device=self.dev)
        self._graphs, self._ws_extra = {}, {}

    # -- collectives ----------------------------------------------------------------------------
    def _all_gather(self, out, inp):
        if self.stage_host:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def _all_reduce(self, t):
        if self.stage_host:
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, group=self.group)

    def _gather_scores(self):
        """s0 of every node from every rank's shard (step 1's collective)."""
        part = self.part
        if self.world == 1:
            self.s0_full.copy_(self.s0_send[:part.n_local])
            return
        self._all_gather(self.s0_pad, self.s0_send)
        for r in range(self.world):                                        # (shards of unequal length: `world` small copies)
            lo, hi = int(part.bounds_host[r]), int(part.bounds_host[r + 1])
            self.s0_full[lo:hi].copy_(self.s0_pad[r * part.n_max:r * part.n_max + hi - lo])

    # -- one step ---------------------------------------------------------------------------------
    def _seg_scores(self, ids_local, labels, B, train_flag):
        """collective-free segment 0: this rank's rows of the score table || plan pass 1 (pcg_step_front_a)."""
        self.ops.step_front_a(self.g, self.w_clf, self.b_clf, self.s0_send, 0, self.part.n_local, ids_local,
                              labels if train_flag else None, self.thresholds, self.rho, train_flag, self._ws_of(B))

    def _seg_select(self, ids_local, labels, B, train_flag):
        """collective-free segment 1 (after the score all-gather): train-pos sort || plan pass 2, centre scores,
        select (lists of global ids)."""
        ops, g, part = self.ops, self.g, self.part
        ws = self._ws_of(B)
        keys = ops.step_front_b(g, self.s0_full, self.keys, ids_local, labels if train_flag else None, self.thresholds,
                                self.rho, train_flag, ws)
        self.center_buf[:B].copy_(self.s0_full[(ids_local.long() + part.lo)])
        ops.choose_select(g, ids_local, labels if train_flag else None, self.s0_full, keys, self.thresholds, self.rho,
                          train_flag, ws, self.cnt[:g.R * B], center_s0=self.center_buf[:B], planned=True)

    def _seg_dense(self, ids_local, labels, B):
        """collective-free segment 2: gather + mean over the extended table, dense step, gradient reduction."""
        ops, g, lib, _p = self.ops, self.g, self.lib, self.ops._p
        agg = self.agg.view(-1)[:g.R * B * self.F].view(g.R, B, self.F)
        ops.aggregate_lists(g, g.X, B, self._ws_of(B), self.cnt[:g.R * B], agg)
        st = ops._stream(self.dev)
        check, c = self._libmod.check, self.cfg
        check(lib.pcg_dense_step(g.desc_ref(), _p(self.theta), self.E, _p(ids_local), _p(labels), B, _p(agg), agg.stride(1),
                                 float(c["alpha"]), 1.0 / (B * self.world), _p(self.logits), _p(self.center), None,
                                 _p(self.row_loss), _p(self.slabs), _p(self.step_counter), st), "pcg_dense_step")
        check(lib.pcg_adam_step(_p(self.theta), _p(self.m), _p(self.v), _p(self.slabs), lib.pcg_dense_n_tiles(B),
                                self.n_params, _p(self.step_counter), c["lr"], 0.9, 0.999, 1e-8, c["weight_decay"],
                                _p(self.grad), 0, st), "pcg_adam_step")

    def _ws_of(self, B):
        if B == self.ws.B:
            return self.ws
        if B not in self._ws_extra:
            self._ws_extra[B] = self.ops.ChooseWorkspace(self.g, B)
        return self._ws_extra[B]

    def _exchange(self, B):
        self.halo.fetch_and_remap_device(self._ws_of(B), B, self.g)

    def forward_sample(self, ids_local: torch.Tensor, labels: Optional[torch.Tensor], train_flag: bool = True):
        """steps 1-3 + aggregate: returns agg [R, B, F] for this rank's centres (local row numbers)."""
        ops, g = self.ops, self.g
        B = ids_local.numel()
        self._seg_scores(ids_local, labels, B, train_flag)
        self._gather_scores()
        self._seg_select(ids_local, labels, B, train_flag)
        self._exchange(B)
        agg = self.agg.view(-1)[:g.R * B * self.F].view(g.R, B, self.F)
        cnt = self.cnt[:g.R * B]
        ops.aggregate_lists(g, g.X, B, self._ws_of(B), cnt, agg)
        return agg, cnt

    def _apply_adam(self):
        c, lib, _p = self.cfg, self.lib, self.ops._p
        self._libmod.check(lib.pcg_adam_step(_p(self.theta), _p(self.m), _p(self.v), _p(self.grad), 1, self.n_params,
                                             _p(self.step_counter), c["lr"], 0.9, 0.999, 1e-8, c["weight_decay"], None, 1,
                                             self.ops._stream(self.dev)), "pcg_adam_step")

    def _eager_step(self, ids_local, labels, B):
        self._seg_scores(ids_local, labels, B, True)
        self._gather_scores()
        self._seg_select(ids_local, labels, B, True)
        self._exchange(B)
        self._seg_dense(ids_local, labels, B)
        self._all_reduce(self.grad)
        self._apply_adam()

    def _graphs_for(self, B):
        """hipGraphs of the two collective-free segments for batch size B (static id / label buffers).  The warm-up that
        precedes the capture is one whole eager step - exchange included, so the dense segment never sees a list of
        global ids (every rank captures at the same step: the first one of a batch size) - whose effect on the
        parameters is undone."""
        gr = self._graphs.get(B)
        if gr is not None:
            return gr
        ids, lab = self.ids_buf[:B], self.lab_buf[:B]
        state = (self.theta.clone(), self.m.clone(), self.v.clone(), self.step_counter.clone())
        self._eager_step(ids, lab, B)              # warm-up: kernel attributes, workspaces, RCCL channels
        torch.cuda.synchronize(self.dev)           # no collective of ours is in flight while capturing
        gr = {}
        for name, fn in (("select", lambda: self._seg_select(ids, lab, B, True)), ("dense", lambda: self._seg_dense(ids, lab, B))):
            g_ = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread may query events while this thread captures
            with torch.cuda.graph(g_, capture_error_mode="thread_local"):
                fn()
            gr[name] = g_
        for dst, src in zip((self.theta, self.m, self.v, self.step_counter), state):
            dst.copy_(src)
        self._graphs[B] = gr
        return gr

    def train_step(self, ids_local: torch.Tensor, labels: torch.Tensor, use_graphs: bool = True):
        """One training step of this rank: only the score all-gather, the halo exchange and the gradient
        all-reduce are launched eagerly; the rest replays two captured graphs."""
        B = ids_local.numel()
        if not use_graphs:
            return self._eager_step(ids_local, labels, B)
        self.ids_buf[:B].copy_(ids_local)
        self.lab_buf[:B].copy_(labels)
        ids_local, labels = self.ids_buf[:B], self.lab_buf[:B]
        gr = self._graphs_for(B)
        self._seg_scores(ids_local, labels, B, True)
        self._gather_scores()
        prof = getattr(self, "_prof", None)
        timed = prof is not None and self._prof_step % self._prof_every == 0
        if prof is not None:
            self._prof_step += 1
        if timed:      # HIP events around the select segment (bench.py's roofline at N > 1), on the launching stream
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        gr["select"].replay()
        if timed:
            ev[1].record()
            prof.append((ev[0], ev[1], ids_local.clone(), self.cnt[:self.g.R * B].clone()))
        self._exchange(B)
        gr["dense"].replay()
        self._all_reduce(self.grad)
        self._apply_adam()

    def profile_select(self, every: int = 10):
        """Start collecting (start event, end event, ids, |set| counts) of every `every`-th step's select segment."""
        self._prof, self._prof_every, self._prof_step = [], every, 0
        return self._prof

    def pick_epoch(self, size: int, epoch: int) -> torch.Tensor:
        """this rank's share of the epoch's picks: local row numbers of owned training nodes."""
        return self.sampler.pick(size, epoch)

    def labels_of(self, ids_local: torch.Tensor) -> torch.Tensor:
        return self.labels_local[ids_local.long()]
