"""Multi-GPU PC-GNN: destination-node partition, RCCL all-to-all for remote neighbour rows.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  The
reference has no distributed code at all; this is new design:

* **Partition.**  Contiguous id ranges balanced by in-edge count (``Partition.balanced``: the CSR
  entries of all relations a rank's rows hold; equal ranges when no degrees are given).  The owner
  holds ``X[v]``, the CSR rows of ``v`` for every relation (neighbour ids stay GLOBAL) and ``v``'s
  label.  Replicated on every rank: all parameters, the train-pos ids and their feature rows (so
  minority over-sampling never needs a fetch).
* **Per window of steps** (``begin_window``; feature rows never change, so a fetched row stays valid):
  every rank walks the CSR rows of the centres it will train on in the next steps, de-duplicates their
  remote neighbours in a hash table sized by the halo capacity (``pcg_halo_collect``), ``all_to_all`` #1
  sends the ids to their owners, the owners gather those rows (``pcg_halo_serve``), ``all_to_all`` #2
  returns them into the halo region of the extended table ``[owned | train-pos | halo]``.  Owner o's
  requests sit in a fixed range of the request list, so both all-to-alls have equal splits and nothing
  waits for the host.  xGMI is point-to-point: the all-to-alls use all 7 links of a GPU at once.
* **Per step** (each rank works on centres it owns): ONE hipGraph replay - five launches and the step's only
  collective, captured with them (RCCL all-reduce inside the graph; probed at construction, eager otherwise):
    1. front (``pcg_step_scores_dist``): Adam on every parameter from the previous step's all-reduced gradient ||
       the train positives' unsorted keys || class-0 scores of every row the rank holds (owned, train-pos, halo -
       every node a list of this window can name), stored by global node id: no score exchange - a row's score
       is the same arithmetic on whichever rank computes it.  The score workgroups recompute the label
       classifier's update themselves (from a snapshot of its state + the gradient), so nothing waits inside;
    2. select (``pcg_choose_select_planned``): sorts the keys itself, centres' scores by global id -> lists of ids;
    3. gather over the extended table (``pcg_gather_lists_dist``): translates the lists' ids as it reads them;
    4. ``pcg_train_dense`` on the local batch, loss scaled by 1 / global batch -> transposed activations (no slabs);
    5. ``pcg_wgrad``: the weight gradients as GEMMs over the local batch -> the flat gradient; then its
       ``all_reduce`` (~107 KB, latency-bound); identical Adam on every rank (the next step's front launch).
* **Memory per rank**: owned rows + train-pos rows + a halo region sized from the window's expected
  demand (``halo_rows``: distinct remote neighbours of window x batch picked centres, x a margin,
  never more than the remote nodes there are) - not from the node count; no per-node flag / slot /
  position tables (the one per-node array is the score vector, 4 bytes a node).
* **Capacity errors** are device flags (an owner asked for more than its range, a list over capacity,
  a step outside its window); ``check()`` all-reduces them so that every rank raises, or none.

The exchange layer (`HaloExchange`) is plain ``torch`` + ``torch.distributed`` and therefore also
runs on CPU tensors over ``gloo`` - that is what the world-size-2 CPU tests drive; `HaloExchangeHip`
replaces its list work by kernels.  The kernels themselves have no CPU path.
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
                       margin: float = 1.5, kept: float = 0.5) -> int:
    """Halo capacity from the expected demand of `batch` picked centres: a centre contributes, per relation,
    ~ceil(kept * deg) neighbours (kept = 0.5: the chosen ones of one step; 1.0: all of them - what a window prefetch
    fetches; minority picks are local: the train-pos block); a fraction (world - 1) / world of them is remote:
    e entries (pick-weighted mean over the rank's training nodes x batch).  e draws from n_remote nodes hit at most
    n_remote * (1 - exp(-e / n_remote)) distinct ones (uniform draws; skewed ones hit fewer).  x margin + 1024, never
    more than the remote nodes there are."""
    if world == 1 or n_remote <= 0:
        return 1
    if weights.size == 0 or weights.sum() <= 0:
        return min(n_remote, 1024)
    p = weights / weights.sum()
    per_centre = sum(float((np.ceil(d * kept) * p).sum()) for d in deg_rel_train_local)
    entries = per_centre * batch * (world - 1) / world
    distinct = n_remote * -math.expm1(-entries / n_remote)
    return int(min(n_remote, math.ceil(distinct * margin + 1024)))


class HaloExchange:
    """The halo exchange of a WINDOW of steps, on any device / backend (plain ``torch`` + ``torch.distributed``).

        X_ext = [ owned rows (n_local) | train-pos rows (P) | halo ((world - 1) x pitch rows) ]

    Feature rows never change, so a fetched row stays valid: `prefetch(csr, centres)` requests every remote, non-train-pos
    neighbour the centres have (de-duplicated; the j-th other rank's requests sit in slots [j * pitch, (j + 1) * pitch) of
    the request list, unused slots = -1, and the halo region has the same layout - the split sizes of both all-to-alls are
    known in advance and nothing has to reach the host before they are issued); `lookup(lst)` then turns a list of global ids (-1 = hole) into row numbers
    of X_ext.  An owner asked for more than `pitch` rows, or a lookup of an id outside the window, sets bits in the sticky
    device word `overflow_word` (2 / 4; those ids become holes).  Work and memory are proportional to the window's CSR
    rows and the lists, not to the node count.  `stage_host=True` stages the collectives through CPU tensors (gloo with
    device tensors)."""

    def __init__(self, part: Partition, X_ext: torch.Tensor, train_pos: Sequence[int], pitch: Optional[int] = None, group=None,
                 stage_host: bool = False, req_out: Optional[torch.Tensor] = None):
        self.part, self.X_ext, self.P = part, X_ext, len(train_pos)
        self.group, self.stage_host = group, stage_host
        dev = X_ext.device
        tp = torch.as_tensor(np.asarray(list(train_pos), dtype=np.int64), device=dev)
        self.pos_ids, self.pos_idx = torch.sort(tp) if self.P else (tp, tp)
        self.halo_base = part.n_local + self.P
        self.halo_cap = X_ext.shape[0] - self.halo_base
        self.peers = max(part.world - 1, 1)
        self.pitch = int(pitch) if pitch is not None else self.halo_cap // self.peers
        assert self.pitch >= 1 and self.halo_cap == self.peers * self.pitch, "the halo region is (world - 1) x pitch rows"
        # split sizes of both all-to-alls: `pitch` with every other rank, nothing with oneself (world 1: one dummy range)
        self.splits = [self.pitch if (r != part.rank or part.world == 1) else 0 for r in range(part.world)]
        self.counts = torch.zeros(131, dtype=torch.int32, device=dev)
        self.overflow_word = self.counts[128:129]
        # the request list doubles as the halo rows' id column (the caller may pass that slice of its row -> id array)
        self.req_out = req_out if req_out is not None else torch.empty(self.halo_cap, dtype=torch.int32, device=dev)
        assert self.req_out.numel() == self.halo_cap and self.req_out.dtype == torch.int32 and self.req_out.is_contiguous()
        self.req_out.fill_(-1)
        self.req_in = torch.full((self.halo_cap,), -1, dtype=torch.int32, device=dev)
        self.rows_out = torch.zeros(self.halo_cap, X_ext.shape[1], dtype=torch.float32, device=dev)
        self.halo_rows = X_ext[self.halo_base:self.halo_base + self.halo_cap]
        self._uniq = torch.zeros(0, dtype=torch.int64, device=dev)
        self._slot = torch.zeros(0, dtype=torch.int64, device=dev)

    # -- collectives (optionally staged through the host) -------------------------------------
    def _a2a(self, out: torch.Tensor, inp: torch.Tensor):
        """fixed splits: the j-th range of `inp` goes to the j-th other rank, the j-th range of `out` comes from it"""
        if self.stage_host and inp.device.type != "cpu":
            o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(o, i, self.splits, self.splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp.contiguous(), self.splits, self.splits, group=self.group)

    def _classify(self, ids: torch.Tensor):
        """(valid & owned, valid & train-pos, train-pos row, valid & remote) of a tensor of global ids (int64)"""
        part = self.part
        valid = ids >= 0
        is_local = valid & (ids >= part.lo) & (ids < part.hi)
        if self.P:
            at = torch.searchsorted(self.pos_ids, ids.clamp(min=0)).clamp(max=self.P - 1)
            is_pos = valid & ~is_local & (self.pos_ids[at] == ids)
            pm = self.pos_idx[at]
        else:
            is_pos, pm = torch.zeros_like(valid), torch.zeros_like(ids)
        return is_local, is_pos, pm, valid & ~is_local & ~is_pos

    # -- the window's exchange ---------------------------------------------------------------------
    def collect(self, csr, centres: torch.Tensor):
        """(1) the request list: the remote, non-train-pos neighbours of `centres` (local rows; duplicates allowed) over all
        relations of `csr` = [(indptr, indices)] (tensors on this device, global neighbour ids), once each."""
        part, dev = self.part, self.X_ext.device
        c = centres.long()
        c = c[(c >= 0) & (c < part.n_local)]
        parts = []
        for indptr, indices in csr:
            beg, deg = indptr[c], indptr[c + 1] - indptr[c]
            if int(deg.sum()) == 0:
                continue
            off = torch.arange(int(deg.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(deg, 0) - deg, deg)
            parts.append(indices[torch.repeat_interleave(beg, deg) + off].long())
        ids = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.int64, device=dev)
        uniq = torch.unique(ids[self._classify(ids)[3]])                     # sorted ascending = grouped by owner
        owner = part.owner(uniq)
        cuts = torch.searchsorted(uniq, part.bounds(dev))
        nth = torch.arange(uniq.numel(), device=dev) - cuts[owner]
        ok = nth < self.pitch
        self._uniq = uniq
        self._slot = torch.where(ok, (owner - (owner > part.rank).long()) * self.pitch + nth, torch.full_like(nth, -1))
        self.req_out.fill_(-1)
        self.req_out[self._slot[ok]] = uniq[ok].to(torch.int32)
        per_owner = (cuts[1:] - cuts[:-1]).to(torch.int32)
        self.counts[:part.world] = per_owner
        self.counts[128] |= (~ok).any().to(torch.int32) * 2
        self.counts[129] = torch.maximum(self.counts[129], per_owner.sum())
        self.counts[130] = torch.maximum(self.counts[130], per_owner.max() if part.world else per_owner.sum())

    def exchange_ids(self):
        """(2) all-to-all #1: slice o of req_out goes to owner o; slice r of req_in = what rank r asks of this rank"""
        self._a2a(self.req_in, self.req_out)

    def serve(self, graph=None):
        """(3) the owner gathers the rows it was asked for (unused slots and ids it does not own: left alone)"""
        row = self.req_in.long() - self.part.lo
        ok = (self.req_in >= 0) & (row >= 0) & (row < self.part.n_local)
        self.rows_out[ok] = self.X_ext[row[ok]]

    def exchange_rows(self):
        """(4) all-to-all #2: the rows come back into the halo region, in the request list's layout"""
        self._a2a(self.halo_rows, self.rows_out)

    def prefetch(self, csr, centres: torch.Tensor) -> None:
        """COLLECTIVE: afterwards the halo region holds every remote row the centres' lists can name"""
        self.collect(csr, centres)
        self.exchange_ids()
        self.serve(csr)
        self.exchange_rows()

    def lookup(self, lst: torch.Tensor) -> None:
        """per step: `lst` (global ids, -1 = hole) -> rows of X_ext, in place"""
        ids = lst.long()
        is_local, is_pos, pm, is_rem = self._classify(ids)
        new = torch.where(is_local, ids - self.part.lo, torch.where(is_pos, self.part.n_local + pm, ids))
        if self._uniq.numel():
            at = torch.searchsorted(self._uniq, ids.clamp(min=0)).clamp(max=self._uniq.numel() - 1)
            hit = is_rem & (self._uniq[at] == ids) & (self._slot[at] >= 0)
            row = self.halo_base + self._slot[at]
        else:
            hit, row = torch.zeros_like(is_rem), ids
        new = torch.where(is_rem, torch.where(hit, row, torch.full_like(ids, -1)), new)
        self.counts[128] |= (is_rem & ~hit).any().to(torch.int32) * 4
        lst.copy_(new.to(lst.dtype))

    @property
    def max_seen(self):
        """largest demand of a window so far (one device read: not for the step loop)"""
        c = self.counts[129:131].cpu().tolist()
        return {"halo_rows": int(c[0]), "rows_from_one_owner": int(c[1]), "pitch": self.pitch}


class HaloExchangeHip(HaloExchange):
    """The same exchange with the list / CSR work done by HIP kernels (pcg_halo_collect: hash-table de-duplication sized by
    the halo capacity; pcg_halo_serve; pcg_halo_lookup).  No host synchronisation anywhere."""

    def __init__(self, part, X_ext, train_pos, pitch: int, group=None, stage_host=False, req_out: Optional[torch.Tensor] = None):
        super().__init__(part, X_ext, train_pos, pitch, group, stage_host, req_out)
        from . import _lib, ops
        self._lib, self._ops = _lib, ops
        dev = X_ext.device
        self.slots = int(_lib.load().pcg_halo_table_slots(self.halo_cap))
        self.table = torch.empty(2 * self.slots, dtype=torch.int32, device=dev)
        self.bounds_dev = part.bounds(dev).to(torch.int32)
        self.pos_ids32, self.pos_idx32 = self.pos_ids.to(torch.int32), self.pos_idx.to(torch.int32)

    def collect(self, graph, centres: torch.Tensor):
        lib, ops, part = self._lib.load(), self._ops, self.part
        _p = ops._p
        centres = centres.to(torch.int32).contiguous()
        self._lib.check(lib.pcg_halo_collect(
            graph.desc_ref(), _p(centres), centres.numel(), part.lo, part.hi, part.n_local, _p(self.pos_ids32), self.P,
            _p(self.bounds_dev), part.world, _p(self.table), self.slots, _p(self.counts), _p(self.req_out), self.halo_cap,
            self.halo_base, self.pitch, part.rank, ops._stream(self.X_ext.device)), "pcg_halo_collect")

    def serve(self, graph):
        ops, part = self._ops, self.part
        self._lib.check(self._lib.load().pcg_halo_serve(
            graph.desc_ref(), ops._p(self.req_in), self.halo_cap, part.lo, part.n_local, ops._p(self.rows_out),
            self.rows_out.stride(0), ops._stream(self.X_ext.device)), "pcg_halo_serve")

    def lookup(self, data, plan, list_capacity: int, B: int, graph):
        """per step: the selection list in `data` (global ids; the chunk table of `plan` - an address, or None: the plan lies
        inside `data` - says which entries are in use) -> rows of X_ext"""
        ops, part = self._ops, self.part
        _p = ops._p
        import ctypes as C
        self._lib.check(self._lib.load().pcg_halo_lookup(
            graph.desc_ref(), B, _p(data), None if plan is None else C.c_void_p(plan), list_capacity, part.lo, part.hi, part.n_local, _p(self.pos_ids32),
            _p(self.pos_idx32), self.P, _p(self.table), self.slots, _p(self.counts), self.halo_cap, self.halo_base,
            ops._stream(self.X_ext.device)), "pcg_halo_lookup")


class DistributedPCGNN:
    """The step driver of one rank of a node-partitioned run (HIP kernels + RCCL)."""

    def __init__(self, w, model_cfg: dict, device, group=None, stage_host: bool = False, halo_rows: Optional[int] = None,
                 halo_pitch: Optional[int] = None, balanced: bool = True, window: int = 8):
        """window: steps per halo prefetch (begin_window) the default capacities are sized for;
        halo_rows: distinct remote rows a window is expected to need (default: from window x batch centres' neighbourhoods);
        halo_pitch: rows reserved per owner (default: from halo_rows); the halo region is (world - 1) x halo_pitch rows."""
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
        # every rank trains centres it owns: a rank without training nodes (e.g. the unlabeled head of the Amazon graph,
        # utils.py:98-99, landing on one rank) cannot take part - say so on every rank instead of failing in the sampler
        if self._agree_max(0 if len(sh["idx_train_local"]) else 1):
            raise ValueError(f"node partition x{self.world}: some rank owns no training node - use fewer ranks or pass `bounds` "
                             "that give every rank labeled training nodes")
        P = len(sh["train_pos"])
        B = cfg["batch_size"]
        n_local = part.n_local
        # pick weights with the GLOBAL label frequencies (utils.py:276), not this shard's
        weights = shard_pick_weights(y_loc, sh["homo_deg_train"], n_train, n_train_pos)
        n_remote = max(w.n - n_local, 0)
        if halo_rows is None:       # from the batch's expected demand, not from the node count
            deg_rel = [np.diff(ip)[sh["idx_train_local"] - part.lo] for ip, _ in sh["csr"]]
            halo_rows = expected_halo_rows(deg_rel, weights, B * max(int(window), 1), self.world, n_remote, kept=1.0)
        halo_rows = max(int(halo_rows), 1)
        # one owner's share of it (+ 25 % for the owners' imbalance), never more than the longest shard has rows; every rank
        # must arrive at the same pitch (equal-split all-to-alls): the largest any rank computed
        if halo_pitch is None:
            halo_pitch = min(part.n_max, -(-halo_rows * 5 // (4 * max(self.world - 1, 1))) + 64) if self.world > 1 else 64
        halo_pitch = self._agree_max(max(int(halo_pitch), 1))
        halo_rows = max(self.world - 1, 1) * halo_pitch
        n_ext = n_local + P + halo_rows
        X_ext = np.zeros((n_ext, F), np.float32)
        X_ext[:n_local] = sh["X_local"]
        X_ext[n_local:n_local + P] = sh["X_pos"]
        csr_ext = []
        for indptr, idx in sh["csr"]:
            ip = np.concatenate([indptr, np.full(n_ext - n_local, indptr[-1], dtype=np.int64)])
            csr_ext.append((ip, idx))
        self.csr_host = sh["csr"]                 # this rank's rows, neighbour ids global (bench.py counts a batch's unique nodes from it)
        self.g = DeviceGraph(X_ext, csr_ext, sh["train_pos"], self.dev, id_space=w.n)
        g = self.g
        # node id of every table row: owned | train-pos | halo (= the exchange's request list, -1 = unused slot)
        self.row_gid = torch.cat([torch.arange(part.lo, part.hi, dtype=torch.int32),
                                  torch.as_tensor(np.asarray(sh["train_pos"], dtype=np.int32).reshape(-1)),
                                  torch.full((halo_rows,), -1, dtype=torch.int32)]).to(self.dev)
        self.halo = HaloExchangeHip(part, g.X, sh["train_pos"], halo_pitch, group, stage_host, req_out=self.row_gid[n_local + P:])
        self.labels_local = torch.from_numpy(sh["labels_local"].astype(np.int32)).to(self.dev)
        self.window = max(int(window), 1)
        self.feature_rows = {"owned": n_local, "train_pos": P, "halo": halo_rows, "halo_pitch": halo_pitch,
                             "serve_buffer": halo_rows, "window_steps": self.window, "unpartitioned_table": int(w.n)}

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

        # scores by GLOBAL node id (4 bytes per node of the whole graph; only the entries of rows this rank holds are ever
        # written or read)
        self.s0_full = torch.zeros(w.n, dtype=torch.float32, device=self.dev)
        self.keys = torch.empty(self.lib.pcg_pos_sort_capacity(P), dtype=torch.int64, device=self.dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        # the plan of a batch depends on its ids / labels and the CSR degrees only: a window's batches are planned together,
        # one plan slot each (pcg_plan_batches); the selection list and the partial sums (the data part) are shared
        from .fused import default_list_capacity
        self.list_capacity, self.clipped = default_list_capacity(g, B)
        self.data = torch.zeros(int(self.lib.pcg_choose_data_bytes(g.desc_ref(), B, self.list_capacity)), dtype=torch.uint8, device=self.dev)
        self.plan_stride = int(self.lib.pcg_choose_plan_bytes(g.desc_ref(), B, self.list_capacity))
        self.win_plans = torch.zeros(max(int(window), 1) * self.plan_stride, dtype=torch.uint8, device=self.dev)
        self._plan_one = {}                       # batch size -> plan slot of a step outside a window
        self.sync = torch.zeros(int(self.lib.pcg_sync_words_count()), dtype=torch.int32, device=self.dev)
        self.opt_flag = self.sync[1:2]             # "a gradient is waiting for its Adam update" (set by the slab sum, cleared by the next select launch)
        self._thr, self._rhos = ops._host_arrays(g, [0.5] * g.R, [cfg["rho"]] * g.R)
        self.cnt = torch.empty(g.R * B, dtype=torch.int32, device=self.dev)
        self.agg = torch.empty(g.R, B, F, dtype=torch.float32, device=self.dev)
        self.logits = torch.empty(B, 2, dtype=torch.float32, device=self.dev)
        self.center = torch.empty(B, 2, dtype=torch.float32, device=self.dev)
        self.row_loss = torch.zeros(B, dtype=torch.float32, device=self.dev)
        # the dense kernel leaves the step's transposed activations (no gradient slabs); the weight gradients are GEMMs over the
        # local batch (pcg_wgrad -> self.grad, which the all-reduce sums over the ranks)
        self.act_ld = 16 * int(self.lib.pcg_dense_n_tiles(B))
        self.acts = torch.zeros(int(self.lib.pcg_wgrad_act_rows(F, self.E, self.R)), self.act_ld, dtype=torch.float32, device=self.dev)
        self.wg_scratch = torch.zeros(int(self.lib.pcg_wgrad_scratch_bytes(F, self.E, self.R, self.act_ld)) // 4, dtype=torch.float32,
                                      device=self.dev)
        # the label classifier's parameters, m, v as of the last applied update: the step's front launch recomputes the
        # classifier's Adam update from them while other workgroups of the same launch store it (pcg_step_scores_dist)
        self.n_clf = 2 * F + 2
        self.off_clf = int(o3)
        self.clf_snap = torch.zeros(3 * self.n_clf, dtype=torch.float32, device=self.dev)
        self._snap_refresh()
        self.thresholds, self.rho = [0.5] * g.R, [cfg["rho"]] * g.R
        self.sampler = PickSampler(sh["idx_train_local"] - part.lo, y_loc, sh["homo_deg_train"],
                                   self.dev, seed=cfg["seed"] + 7919 * self.rank)
        self.sampler.weights = weights
        self.sampler.cum_host = np.cumsum(weights)
        self.sampler.cum = torch.from_numpy(self.sampler.cum_host).to(self.dev)
        self.B = B
        self.ids_buf = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.lab_buf = torch.zeros(B, dtype=torch.int32, device=self.dev)
        self.win_ids = torch.zeros(B * self.window, dtype=torch.int32, device=self.dev)
        self.win_lab = torch.zeros(B * self.window, dtype=torch.int32, device=self.dev)
        self._graphs = {}
        # the gradient all-reduce INSIDE the step's hipGraph (RCCL collectives can be stream-captured): a step is then one graph
        # launch and nothing else - no eager collective, no second host call per step.  Probed once (a captured all-reduce of
        # eight floats, replayed and checked) and only used if every rank's probe passed; PCG_DIST_GRAPH_COLLECTIVES=0 turns it off.
        self.collectives_in_graph = self._probe_collective_capture()
        import os
        # a whole window as ONE graph (train_window): on by default at world size 1 - where its all-to-alls are copies -, opt-in
        # (PCG_DIST_WINDOW_GRAPH=1) beyond: there they are grouped send / receive pairs, a capture path that no one-GPU box can
        # rehearse; the per-step graphs with the captured all-reduce (the probe above covers exactly that) are the default at N > 1
        wg_env = os.environ.get("PCG_DIST_WINDOW_GRAPH")
        self.window_graphs = self.collectives_in_graph and (wg_env == "1" or (wg_env is None and self.world == 1))
        if self.collectives_in_graph:
            # graphs holding captured collectives must be gone before the communicator is torn down (close()): also when the
            # process ends without the caller having said so
            import atexit
            import weakref
            ref = weakref.ref(self)
            atexit.register(lambda: ref() is not None and ref().close())

    def _probe_collective_capture(self) -> bool:
        import os
        if self.stage_host or os.environ.get("PCG_DIST_GRAPH_COLLECTIVES", "1") == "0":
            return False
        ok = False
        try:
            if dist.get_backend(self.group) == "nccl":
                t = torch.ones(8, dtype=torch.float32, device=self.dev)
                dist.all_reduce(t, group=self.group)               # (the communicator is set up outside the capture)
                torch.cuda.synchronize(self.dev)
                a_in = torch.full((self.world * 4,), float(self.rank), dtype=torch.float32, device=self.dev)
                a_out = torch.empty_like(a_in)
                dist.all_to_all_single(a_out, a_in, group=self.group)
                torch.cuda.synchronize(self.dev)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                    dist.all_reduce(t, group=self.group)
                    dist.all_to_all_single(a_out, a_in, group=self.group)
                t.fill_(1.0)
                a_out.fill_(-1.0)
                gr.replay()
                torch.cuda.synchronize(self.dev)
                want = torch.arange(self.world, dtype=torch.float32, device=self.dev).repeat_interleave(4)
                ok = bool((t == float(self.world)).all().item()) and bool(torch.equal(a_out, want))
        except Exception:                                           # (any refusal to capture: the eager collective stays)
            ok = False
        return self._agree_max(0 if ok else 1) == 0

    def _agree_max(self, value: int) -> int:
        """the largest `value` over the ranks (construction-time collective)"""
        if self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.int64, device="cpu" if self.stage_host else self.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    # -- collectives ----------------------------------------------------------------------------
    def _all_reduce(self, t):
        if self.stage_host:
            c = t.cpu()
            dist.all_reduce(c, group=self.group)
            t.copy_(c)
        else:
            dist.all_reduce(t, group=self.group)

    # -- the window's exchange -------------------------------------------------------------------
    def begin_window(self, ids_window_local: torch.Tensor) -> None:
        """COLLECTIVE.  Fetch the feature rows of every remote neighbour the given centres (local rows of owned nodes, any
        number, duplicates allowed) have; until the next call, steps on (subsets of) these centres need no exchange.  Two
        equal-split all-to-alls, no host synchronisation."""
        self.halo.prefetch(self.g, ids_window_local.to(torch.int32))

    # -- one step ---------------------------------------------------------------------------------
    def _plan(self, ids, labels, n_total, B, plans: torch.Tensor, train_flag: bool):
        """plans of the batches ids[s * B : (s + 1) * B] into consecutive plan slots (one launch for all of them)"""
        _p = self.ops._p
        self._libmod.check(self.lib.pcg_plan_batches(
            self.g.desc_ref(), _p(ids), _p(labels if train_flag else None), n_total, B, self._thr, self._rhos, 1 if train_flag else 0, 0,
            _p(plans), self.plan_stride, self.list_capacity, _p(self.status), None, self.ops._stream(self.dev)), "pcg_plan_batches")

    def _plan_single(self, ids, labels, B, train_flag) -> int:
        """a step outside a window: its own plan slot (one per batch size), planned now; returns the slot's address"""
        buf = self._plan_one.get(B)
        if buf is None:
            buf = self._plan_one[B] = torch.zeros(self.plan_stride, dtype=torch.uint8, device=self.dev)
        self._plan(ids, labels, B, B, buf, train_flag)
        return buf.data_ptr()

    def _seg_front(self, ids_local, labels, B, train_flag, plan: int):
        """scores of every row this rank holds (owned, train-pos, halo), by node id || the train positives' unsorted keys from
        their replicated rows (one launch); select - it sorts the keys itself and reads the centres' scores by global id -
        (lists of global ids); lists -> rows of the extended table."""
        import ctypes as C
        ops, g, part, lib, _p = self.ops, self.g, self.part, self.lib, self.ops._p
        check = self._libmod.check
        st = ops._stream(self.dev)
        lab = labels if train_flag else None
        P = g.n_pos
        sort = train_flag and P > 0
        in_select = sort and bool(lib.pcg_pos_sort_in_select(P))
        check(lib.pcg_step_scores(g.desc_ref(), _p(self.w_clf), _p(self.b_clf), 0, g.n_nodes, _p(self.s0_full), _p(self.row_gid),
                                  _p(self.keys) if in_select else None, part.n_local, _p(self.sync), None, st), "pcg_step_scores")
        if sort and not in_select:           # too many positives for the in-kernel sort: the bucket sort's launches
            ops.pos_sort(g, self.s0_full, self.keys)
        check(lib.pcg_choose_select_planned(
            g.desc_ref(), _p(ids_local), _p(lab), B, _p(self.s0_full), None, _p(self.keys) if sort else None, self._thr, self._rhos,
            1 if train_flag else 0, 0, _p(self.cnt[:g.R * B]), _p(self.data), C.c_void_p(plan), self.list_capacity, _p(self.status),
            _p(self.sync) if (in_select or train_flag) else None, part.lo, st), "pcg_choose_select_planned")
        self.halo.lookup(self.data, plan, self.list_capacity, B, g)

    def _snap_refresh(self):
        """the classifier snapshot from theta / m / v (construction, and after an update applied outside a step: flush())"""
        n, o = self.n_clf, self.off_clf
        for j, src in enumerate((self.theta, self.m, self.v)):
            self.clf_snap[j * n:(j + 1) * n].copy_(src[o:o + n])

    def _seg_step(self, ids_local, labels, B, plan: int):
        """the collective-free part of a training step, five launches:
          front   [Adam on every parameter from the all-reduced gradient, if one is waiting || the train positives' unsorted keys ||
                   the score of every row this rank holds (owned, train-pos, halo), by node id - with the classifier AFTER that
                   update, recomputed from its snapshot]                                     (pcg_step_scores_dist)
          select  [sorts the keys itself, reads the centres' scores by node id; lists of node ids; clears the "waiting" flag]
          gather  over the extended table, translating the lists' node ids as it reads them (no look-up launch) [+ the snapshot]
          dense   forward, loss (scaled by 1 / global batch), activation gradients -> transposed activations (no slabs)
          wgrad   the weight gradients as GEMMs over the local batch -> self.grad; marks it as waiting
        (round 3: seven - apply_pending, front, select, halo_lookup, gather, dense, grad_reduce)."""
        import ctypes as C
        ops, g, part, lib, _p = self.ops, self.g, self.part, self.lib, self.ops._p
        check, c = self._libmod.check, self.cfg
        st = ops._stream(self.dev)
        P = g.n_pos
        in_select = P > 0 and bool(lib.pcg_pos_sort_in_select(P))
        check(lib.pcg_step_scores_dist(g.desc_ref(), _p(self.theta), _p(self.m), _p(self.v), self.E, _p(self.grad), _p(self.clf_snap), 0,
                                       g.n_nodes, _p(self.s0_full), _p(self.row_gid), _p(self.keys) if in_select else None,
                                       part.n_local, _p(self.step_counter), _p(self.sync), c["lr"], 0.9, 0.999, 1e-8,
                                       c["weight_decay"], st), "pcg_step_scores_dist")
        if P > 0 and not in_select:          # too many positives for the in-kernel sort: the bucket sort's launches
            ops.pos_sort(g, self.s0_full, self.keys)
        check(lib.pcg_choose_select_planned(
            g.desc_ref(), _p(ids_local), _p(labels), B, _p(self.s0_full), None, _p(self.keys) if P > 0 else None, self._thr, self._rhos,
            1, 0, _p(self.cnt[:g.R * B]), _p(self.data), C.c_void_p(plan), self.list_capacity, _p(self.status), _p(self.sync),
            part.lo, st), "pcg_choose_select_planned")
        agg = self.agg.view(-1)[:g.R * B * self.F].view(g.R, B, self.F)
        h = self.halo
        check(lib.pcg_gather_lists_dist(
            _p(g.X), g.feat_dim, g.X.stride(0), g.X.shape[0], g.R * B, _p(self.cnt), g.desc_ref(), B, _p(self.data), C.c_void_p(plan),
            self.list_capacity, _p(agg), agg.stride(1), _p(self.status), part.lo, part.hi, part.n_local, _p(h.pos_ids32),
            _p(h.pos_idx32), h.P, _p(h.table), h.slots, _p(h.counts), h.halo_cap, h.halo_base, _p(self.theta), _p(self.m), _p(self.v),
            self.off_clf, self.n_clf, _p(self.clf_snap), st), "pcg_gather_lists_dist")
        check(lib.pcg_train_dense(g.desc_ref(), _p(self.theta), None, None, self.E, _p(ids_local), _p(labels), B, _p(agg),
                                  agg.stride(1), _p(self.cnt), _p(self.data), C.c_void_p(plan), self.list_capacity, float(c["alpha"]),
                                  1.0 / (B * self.world), _p(self.logits), _p(self.center), None, _p(self.row_loss),
                                  None, _p(self.step_counter), None, c["lr"], 0.9, 0.999, 1e-8, c["weight_decay"], 4,
                                  _p(self.acts), self.act_ld, None, st), "pcg_train_dense")
        check(lib.pcg_wgrad(_p(self.acts), self.act_ld, B, self.F, self.E, self.R, None, None, None, None, c["lr"], 0.9, 0.999, 1e-8,
                            c["weight_decay"], _p(self.grad), 0, 1, _p(self.wg_scratch), _p(self.opt_flag), st), "pcg_wgrad")

    def _enqueue_apply(self, clear: bool = False):
        """Adam from self.grad on every parameter if a gradient is waiting (device flag; a no-op launch otherwise).  Inside a
        step the flag is cleared by the step's select launch; clear=True adds a one-thread launch that does it."""
        c, lib, _p = self.cfg, self.lib, self.ops._p
        self._libmod.check(lib.pcg_adam_apply_pending(_p(self.theta), _p(self.m), _p(self.v), _p(self.grad), self.n_params,
                                                      _p(self.step_counter), _p(self.opt_flag), 1 if clear else 0, c["lr"], 0.9, 0.999,
                                                      1e-8, c["weight_decay"], self.ops._stream(self.dev)), "pcg_adam_apply_pending")

    def flush(self):
        """Apply the last step's Adam update now (it otherwise rides at the head of the next step's launches): call before
        the parameters are read.  Enqueued, not synchronised; every rank applies the same all-reduced gradient."""
        self._enqueue_apply(clear=True)
        self._snap_refresh()

    def forward_sample(self, ids_local: torch.Tensor, labels: Optional[torch.Tensor], train_flag: bool = True,
                       prefetch: bool = True):
        """COLLECTIVE if prefetch.  Scores, choose and aggregate for this rank's centres (local row numbers): returns agg
        [R, B, F] and the set sizes.  prefetch=False: the centres are covered by the current window."""
        ops, g = self.ops, self.g
        B = ids_local.numel()
        import ctypes as C
        self.flush()                               # (the select launch below clears the "gradient waiting" word)
        if prefetch:
            self.begin_window(ids_local)
        plan = self._plan_single(ids_local, labels, B, train_flag)
        self._seg_front(ids_local, labels, B, train_flag, plan)
        agg = self.agg.view(-1)[:g.R * B * self.F].view(g.R, B, self.F)
        cnt = self.cnt[:g.R * B]
        _p, lib = ops._p, self.lib
        self._libmod.check(lib.pcg_aggregate_lists_planned(_p(g.X), g.feat_dim, g.X.stride(0), g.X.shape[0], g.R * B, _p(cnt), g.desc_ref(),
                                                           B, _p(self.data), C.c_void_p(plan), self.list_capacity, self._libmod.PCG_NORM_COUNT,
                                                           _p(agg), agg.stride(1), _p(self.status), ops._stream(self.dev)),
                           "pcg_aggregate_lists_planned")
        return agg, cnt

    def _graph_for(self, B, slot: Optional[int] = None):
        """hipGraph of the collective-free part of a step for batch size B, reading its centres from the static id / label
        buffers (slot None) or from batch `slot` of the static window buffers.  The warm-up that precedes the capture runs the
        same kernels once; its effect on the step counter is undone, and the gradient it produced is discarded."""
        gr = self._graphs.get((B, slot))
        if gr is not None:
            return gr
        if slot is None:
            ids, lab = self.ids_buf[:B], self.lab_buf[:B]
        else:
            ids, lab = self.win_ids[slot * self.B:slot * self.B + B], self.win_lab[slot * self.B:slot * self.B + B]
        counter = self.step_counter.clone()
        # (a step outside a window plans itself, inside its graph; a window's batches were planned together by train_window)
        if slot is None:
            step = lambda: self._seg_step(ids, lab, B, self._plan_single(ids, lab, B, True))
        else:
            plan = self.win_plans.data_ptr() + slot * self.plan_stride
            step = lambda: self._seg_step(ids, lab, B, plan)
        step()                                     # warm-up: kernel attributes, plan slots (its head applies a waiting gradient: due anyway)
        self.opt_flag.zero_()                      # ... and the gradient the warm-up itself left behind is not one to apply
        if self.collectives_in_graph and not getattr(self, "_ar_warm", False):
            # an eager all-reduce of the gradient's size first: whatever the communicator sets up lazily for a message of this
            # size (channels, staging buffers) is set up outside a capture, where allocating is allowed
            dist.all_reduce(torch.zeros_like(self.grad), group=self.group)
            self._ar_warm = True
        torch.cuda.synchronize(self.dev)           # no collective of ours is in flight while capturing
        gr = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog thread may query events while this thread captures
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            step()
            if self.collectives_in_graph:          # the step's only collective, captured behind its last kernel
                dist.all_reduce(self.grad, group=self.group)
        self.step_counter.copy_(counter)
        self._graphs[(B, slot)] = (gr, ids)
        return gr, ids

    def _replay_step(self, gr, ids, B, timed: Optional[bool] = None):
        """graph replay (bracketed by HIP events on every `profile_select`-th step, or as the caller says), gradient all-reduce"""
        prof = getattr(self, "_prof", None)
        if timed is None:
            timed = prof is not None and self._prof_step % self._prof_every == 0
            if prof is not None:
                self._prof_step += 1
        timed = timed and prof is not None
        if timed:      # on the launching stream: bench.py's roofline at N > 1
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        gr.replay()
        if timed:
            ev[1].record()
            prof.append((ev[0], ev[1], ids.clone(), self.cnt[:self.g.R * B].clone()))
        if not self.collectives_in_graph:
            self._all_reduce(self.grad)            # (its Adam update rides in the next step's front launch; flush() applies it now)

    def train_step(self, ids_local: torch.Tensor, labels: torch.Tensor, use_graphs: bool = True):
        """COLLECTIVE.  One training step on centres the current window covers (begin_window): one graph replay (or the same
        kernels launched one by one), the gradient all-reduce - the step's only collective -, Adam.  Nothing waits for the
        host; a list or an exchange over capacity raises a device flag (check())."""
        B = ids_local.numel()
        if not use_graphs:
            self._seg_step(ids_local, labels, B, self._plan_single(ids_local, labels, B, True))
            self._all_reduce(self.grad)
            return
        self.ids_buf[:B].copy_(ids_local)
        self.lab_buf[:B].copy_(labels)
        gr, ids = self._graph_for(B)
        self._replay_step(gr, ids, B)

    def train_window(self, ids_window_local: torch.Tensor, labels_window: torch.Tensor, use_graphs: bool = True) -> None:
        """COLLECTIVE.  begin_window on all the centres, then one training step per consecutive batch of cfg['batch_size'] (the
        tail batch may be shorter; every rank must pass the same number of centres).  With graphs, the window's ids and
        labels are copied once into static buffers that the steps' graphs (one per batch position) read in place."""
        n = ids_window_local.numel()
        if use_graphs and n <= self.win_ids.numel():
            self.win_ids[:n].copy_(ids_window_local)
            self.win_lab[:n].copy_(labels_window)
            # With the collectives capturable the WHOLE window is one hipGraph: the halo exchange (collect, both all-to-alls, serve),
            # the plans, every step's five launches and its all-reduce - one launch per window, no gap between its kernels.
            # (A window of a size not seen yet runs step by step once - which also warms every kernel up - and is captured
            #  afterwards; so does every `profile_select`-th window, all of its steps bracketed by events.)
            prof = getattr(self, "_prof", None)
            bracket = prof is not None and self._prof_win % self._prof_every == 0
            if prof is not None:
                self._prof_win += 1
            gr_w = self._graphs.get(("window", n)) if self.window_graphs else None
            if gr_w is not None and not bracket:
                gr_w.replay()
                return
            self.begin_window(self.win_ids[:n])
            self._plan(self.win_ids, self.win_lab, n, self.B, self.win_plans, True)     # every batch of the window: one launch
            for slot, b0 in enumerate(range(0, n, self.B)):
                B = min(self.B, n - b0)
                gr, ids = self._graph_for(B, slot)
                self._replay_step(gr, ids, B, timed=bracket)
            if self.window_graphs and gr_w is None:
                self._capture_window(n)
        else:
            self.begin_window(ids_window_local)
            for b0 in range(0, n, self.B):
                self.train_step(ids_window_local[b0:b0 + self.B], labels_window[b0:b0 + self.B], use_graphs)

    def _capture_window(self, n: int) -> None:
        """one hipGraph of a whole window of n centres in the static buffers (every kernel in it has run at least once)"""
        torch.cuda.synchronize(self.dev)           # no collective of ours is in flight while capturing
        gr = torch.cuda.CUDAGraph()
        import gc
        gc.collect()
        was_on = gc.isenabled()
        gc.disable()                               # (a collection tears tensors / events down with calls a capture does not allow)
        try:
            with torch.cuda.graph(gr, capture_error_mode="thread_local"):
                self.begin_window(self.win_ids[:n])
                self._plan(self.win_ids, self.win_lab, n, self.B, self.win_plans, True)
                for slot, b0 in enumerate(range(0, n, self.B)):
                    B = min(self.B, n - b0)
                    self._seg_step(self.win_ids[b0:b0 + B], self.win_lab[b0:b0 + B], B, self.win_plans.data_ptr() + slot * self.plan_stride)
                    dist.all_reduce(self.grad, group=self.group)
        finally:
            if was_on:
                gc.enable()
        self._graphs[("window", n)] = gr

    def close(self) -> None:
        """Drop every captured graph (call before ``destroy_process_group``: graphs that hold captured collectives must be gone
        before their communicator is torn down - with a whole window's all-to-alls captured the teardown otherwise never returned)."""
        import gc
        torch.cuda.synchronize(self.dev)
        self._graphs.clear()
        gc.collect()
        torch.cuda.synchronize(self.dev)

    def check(self) -> None:
        """COLLECTIVE.  Raise - on every rank, or on none - if any rank's exchange or selection list went over capacity, or a
        step met an id outside its window, since the last check (those steps worked on lists with holes).  One small
        all-reduce and one host read: call it per epoch, not per step."""
        self.flush()
        # the two words are bit fields: every bit travels as a 0/1 entry of its own, so that MAX over the ranks is a bitwise OR
        # (rank A's pitch overflow and rank B's id-outside-window would otherwise collapse into the larger number; NCCL has no BOR)
        words = torch.stack([self.halo.overflow_word[0], self.status[0]]).to(torch.int64)
        bits = torch.arange(16, device=words.device, dtype=torch.int64)
        flags = (words[:, None] >> bits[None, :]) & 1
        if self.world > 1:
            if self.stage_host:
                c = flags.cpu()
                dist.all_reduce(c, op=dist.ReduceOp.MAX, group=self.group)
                flags = c
            else:
                dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        flags = (flags.to(bits.device) << bits[None, :]).sum(1)
        halo, lists = (int(x) for x in flags.cpu().tolist())
        self.halo.overflow_word.zero_()
        self.status.zero_()
        if halo or lists:
            seen = self.halo.max_seen
            what = []
            if halo & 3:
                what.append(f"halo exchange (flags {halo}; this rank needed up to {seen['rows_from_one_owner']} rows from one "
                            f"owner, pitch {seen['pitch']}) - raise halo_pitch / halo_rows or shorten the window")
            if halo & 4:
                what.append("a step's list named a remote node its window did not fetch (begin_window must cover the step's centres)")
            if lists:
                what.append(f"selection list (status {lists}) - raise the list capacity")
            raise RuntimeError("partitioned step over capacity on some rank: " + "; ".join(what))

    def profile_select(self, every: int = 10):
        """Start collecting (start event, end event, ids, |set| counts) of the steps of every `every`-th window (train_window;
        every `every`-th step of train_step calls)."""
        self._prof, self._prof_every, self._prof_step, self._prof_win = [], every, 0, 0
        return self._prof

    def pick_epoch(self, size: int, epoch: int) -> torch.Tensor:
        """this rank's share of the epoch's picks: local row numbers of owned training nodes."""
        return self.sampler.pick(size, epoch)

    def labels_of(self, ids_local: torch.Tensor) -> torch.Tensor:
        return self.labels_local[ids_local.long()]
