"""Multi-GPU PC-GNN: destination-node partition, RCCL all-to-all for remote neighbour rows.

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  The
reference has no distributed code at all; this is new design:

* **Partition.**  Node ``v`` is owned by rank ``v // n_per`` (equal contiguous id ranges).
  The owner holds ``X[v]``, the CSR rows of ``v`` for every relation (neighbour ids stay
  GLOBAL) and ``v``'s label.  Replicated on every rank: all parameters, the train-pos ids
  and their feature rows (so minority over-sampling never needs a fetch).
* **Per step** (each rank works on centres it owns):
    1. class-0 scores of the owned rows (``pcg_score_table``) -> ``all_gather`` -> every
       rank has ``s0[N]``  (4 N bytes; 0.18 MB for YelpChi, 40 MB at 10 M nodes);
    2. train-pos sort + choose (``pcg_choose_select``) on local rows -> selection lists of
       global ids;
    3. halo exchange: unique remote ids are bucketed by owner, ``all_to_all`` #1 sends the
       ids, the owners gather those rows, ``all_to_all`` #2 returns them into the halo region
       of the extended feature table; the lists are re-indexed into that table;
    4. ``pcg_aggregate_lists`` over the extended table, ``pcg_dense_step`` on the local batch
       with the loss scaled by 1 / global batch, gradient ``all_reduce`` (~107 KB), identical
       Adam on every rank.
  xGMI is point-to-point, so the all-to-all uses all 7 links of a GPU at once; the gradient
  all-reduce is latency-bound at this size.

The exchange layer (`HaloExchange`) is plain ``torch`` + ``torch.distributed`` and therefore also
runs on CPU tensors over ``gloo`` - that is what the world-size-2 CPU tests drive.  The kernels
themselves have no CPU path.
"""
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


class Partition:
    """Equal contiguous id ranges: rank r owns [r * n_per, min((r + 1) * n_per, N))."""

    def __init__(self, n_nodes: int, world: int, rank: int):
        self.n_nodes, self.world, self.rank = n_nodes, world, rank
        self.n_per = (n_nodes + world - 1) // world
        self.lo = min(rank * self.n_per, n_nodes)
        self.hi = min(self.lo + self.n_per, n_nodes)
        self.n_local = self.hi - self.lo

    def owner(self, ids):
        return ids // self.n_per

    def bounds(self, device=None) -> torch.Tensor:
        return torch.arange(self.world + 1, device=device, dtype=torch.int64) * self.n_per


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


class HaloExchange:
    """Fetch the feature rows of remote ids and re-index a selection list into the extended table

        X_ext = [ owned rows (n_local) | train-pos rows (P) | halo (per step) ]

    `lst` holds global ids (-1 = hole).  After `fetch_and_remap(lst)`, `lst` holds row numbers of
    X_ext and the halo region holds the rows fetched this step.  Works on any device / backend;
    `stage_host=True` stages the collectives through CPU tensors (gloo with device tensors).
    """

    def __init__(self, part: Partition, X_ext: torch.Tensor, n_pos: int, posmap: torch.Tensor,
                 group=None, stage_host: bool = False):
        self.part, self.X_ext, self.P, self.posmap = part, X_ext, n_pos, posmap
        self.group, self.stage_host = group, stage_host
        self.halo_base = part.n_local + n_pos
        self.halo_cap = X_ext.shape[0] - self.halo_base
        self.last_stats = {}

    # -- collectives (optionally staged through the host) -------------------------------------
    def _a2a(self, out: torch.Tensor, inp: torch.Tensor, out_splits: Optional[List[int]], in_splits: Optional[List[int]]):
        if self.stage_host and inp.device.type != "cpu":
            o, i = torch.empty(out.shape, dtype=out.dtype), inp.cpu()
            dist.all_to_all_single(o, i, out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp.contiguous(), out_splits, in_splits, group=self.group)

    def fetch_and_remap(self, lst: torch.Tensor) -> int:
        part, dev = self.part, lst.device
        ids = lst.long()
        valid = ids >= 0
        is_local = valid & (ids >= part.lo) & (ids < part.hi)
        pm = self.posmap[ids.clamp(min=0)].long()
        is_pos = valid & ~is_local & (pm >= 0)
        is_rem = valid & ~is_local & ~is_pos
        rem_ids = ids[is_rem]
        uniq, inv = torch.unique(rem_ids, return_inverse=True)             # sorted ascending = grouped by owner
        cuts = torch.searchsorted(uniq, part.bounds(dev))
        send_counts = (cuts[1:] - cuts[:-1]).to(torch.int64)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts, None, None)
        sc, rc = send_counts.tolist(), recv_counts.tolist()                 # host needs the split sizes
        n_halo = int(uniq.numel())
        if n_halo > self.halo_cap:
            raise RuntimeError(f"halo needs {n_halo} rows but only {self.halo_cap} were reserved")
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
    """Same exchange with the list work done by three HIP kernels (pcg_halo_classify / _compact / _remap)
    and ONE host synchronisation per step: the per-owner request counts of every rank are all-gathered as
    a world x world matrix, which gives a rank both its send and its receive split sizes."""

    def __init__(self, part, X_ext, n_pos, posmap, n_nodes, group=None, stage_host=False):
        super().__init__(part, X_ext, n_pos, posmap, group, stage_host)
        from . import _lib, ops
        self._lib, self._ops = _lib, ops
        dev = X_ext.device
        self.flag = torch.zeros(n_nodes, dtype=torch.int32, device=dev)
        self.uniq = torch.empty(self.halo_cap, dtype=torch.int32, device=dev)
        self.n_nodes = n_nodes
        w = part.world
        self._last = torch.tensor([min((r + 1) * part.n_per, n_nodes) - 1 for r in range(w)], dtype=torch.long, device=dev)
        self._counts_all = torch.zeros(w * w, dtype=torch.int32, device=dev)
        # what THIS rank may be asked for: every other rank can request each of its owned rows once per step
        self.serve_cap = max(1, (w - 1) * part.n_local)
        self._req = torch.empty(self.serve_cap, dtype=torch.int32, device=dev)
        self._rows = torch.zeros(self.serve_cap, X_ext.shape[1], dtype=torch.float32, device=dev)

    def _gather_counts(self, out, inp):
        if self.stage_host:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu(), group=self.group)
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp, group=self.group)

    def fetch_and_remap_device(self, lst_full: torch.Tensor, total_dev: torch.Tensor, graph) -> int:
        """lst_full: the workspace's whole list buffer; total_dev: device int64 scalar (entries in use)."""
        lib, ops, part = self._lib.load(), self._ops, self.part
        _p, st = ops._p, ops._stream(lst_full.device)
        w, rank = part.world, part.rank
        self._lib.check(lib.pcg_halo_classify(_p(lst_full), _p(total_dev), lst_full.numel(), part.lo, part.hi, part.n_local,
                                              _p(self.posmap), _p(self.flag), st), "pcg_halo_classify")
        slot = torch.cumsum(self.flag, 0, dtype=torch.int32)
        ends = slot[self._last]                                            # inclusive count up to each owner's last id
        counts = torch.diff(ends, prepend=ends.new_zeros(1))              # ids requested from every owner
        self._gather_counts(self._counts_all, counts.contiguous())
        mat = self._counts_all.view(w, w).cpu()                            # the step's single host sync
        sc, rc = mat[rank].tolist(), mat[:, rank].tolist()
        n_halo, n_req = sum(sc), sum(rc)
        if n_halo > self.halo_cap or n_req > self.serve_cap:
            raise RuntimeError(f"halo exchange: {n_halo} rows to fetch (capacity {self.halo_cap}), {n_req} to serve "
                               f"(capacity {self.serve_cap})")
        self._lib.check(lib.pcg_halo_compact(_p(self.flag), _p(slot), self.n_nodes, _p(self.uniq), st), "pcg_halo_compact")
        req = self._req[:n_req]
        self._a2a(req, self.uniq[:n_halo], rc, sc)                         # all-to-all #1: requested ids
        rows = self._rows[:n_req]
        if n_req:
            ops.gather_rows(graph, req - part.lo, out=rows)                # the owner gathers its rows (pad columns stay 0)
        halo = self.X_ext[self.halo_base:self.halo_base + n_halo]
        self._a2a(halo, rows, sc, rc)                                      # all-to-all #2: feature rows
        self._lib.check(lib.pcg_halo_remap(_p(lst_full), _p(total_dev), lst_full.numel(), _p(slot), self.halo_base,
                                           _p(self.flag), st), "pcg_halo_remap")
        self.last_stats = {"halo_rows": n_halo, "rows_served": n_req, "bytes_in": n_halo * self.X_ext.shape[1] * 4,
                           "bytes_out": n_req * self.X_ext.shape[1] * 4}
        return n_halo


class DistributedPCGNN:
    """The step driver of one rank of a node-partitioned run (HIP kernels + RCCL)."""

    def __init__(self, w, model_cfg: dict, device, group=None, stage_host: bool = False, halo_rows: Optional[int] = None):
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
        part = self.part = Partition(w.n, self.world, self.rank)
        sh = shard_workload(w, part)
        F = w.X.shape[1]
        P = len(sh["train_pos"])
        B = cfg["batch_size"]
        n_local = part.n_local
        if halo_rows is None:     # worst case: every chosen neighbour of every centre is a distinct remote node
            md = max(int(np.diff(ip).max()) if len(ip) > 1 else 0 for ip, _ in sh["csr"])
            halo_rows = min(w.n, len(sh["csr"]) * B * max(md, 1))
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
        posmap = torch.full((w.n,), -1, dtype=torch.int32)
        if P:
            posmap[torch.as_tensor(sh["train_pos"], dtype=torch.long)] = torch.arange(P, dtype=torch.int32)
        self.halo = HaloExchangeHip(part, g.X, P, posmap.to(self.dev), w.n, group, stage_host)
        self.labels_local = torch.from_numpy(sh["labels_local"].astype(np.int32)).to(self.dev)

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

        self.s0_send = torch.zeros(part.n_per, dtype=torch.float32, device=self.dev)
        self.s0_full = torch.zeros(part.n_per * self.world, dtype=torch.float32, device=self.dev)
        self.keys = torch.empty(self.lib.pcg_pos_sort_capacity(P), dtype=torch.int64, device=self.dev)
        self.ws = ops.ChooseWorkspace(g, B)
        self.cnt = torch.empty(g.R * B, dtype=torch.int32, device=self.dev)
        self.agg = torch.empty(g.R, B, F, dtype=torch.float32, device=self.dev)
        self.logits = torch.empty(B, 2, dtype=torch.float32, device=self.dev)
        self.center = torch.empty(B, 2, dtype=torch.float32, device=self.dev)
        self.row_loss = torch.zeros(B, dtype=torch.float32, device=self.dev)
        self.slabs = torch.empty(self.lib.pcg_dense_n_tiles(B), n, dtype=torch.float32, device=self.dev)
        self.thresholds, self.rho = [0.5] * g.R, [cfg["rho"]] * g.R
        self.sampler = PickSampler(sh["idx_train_local"] - part.lo, w.labels[sh["idx_train_local"]], sh["homo_deg_train"],
                                   self.dev, seed=cfg["seed"] + 7919 * self.rank)
        # LF uses the GLOBAL class counts (utils.py:276), not this shard's
        y_all = w.labels[w.idx_train]
        lf = np.where(w.labels[sh["idx_train_local"]] == 1, y_all.sum(), len(y_all))
        self.sampler.cum_host = np.cumsum(sh["homo_deg_train"] / lf)
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
        ws = self._ws_of(B)
        total_dev = ws.view(0, torch.int64, self.g.R * B + 1)[-1:]          # list entries in use: stays on the device
        self.halo.fetch_and_remap_device(ws.view(2, torch.int32, ws.list_capacity), total_dev, self.g)

    def forward_sample(self, ids_local: torch.Tensor, labels: Optional[torch.Tensor], train_flag: bool = True):
        """steps 1-3 + aggregate: returns agg [R, B, F] for this rank's centres (local row numbers)."""
        ops, g, part = self.ops, self.g, self.part
        B = ids_local.numel()
        self._seg_scores(ids_local, labels, B, train_flag)
        self._all_gather(self.s0_full, self.s0_send)
        self._seg_select(ids_local, labels, B, train_flag)
        self._exchange(B)
        agg = self.agg.view(-1)[:g.R * B * self.F].view(g.R, B, self.F)
        cnt = self.cnt[:g.R * B]
        ops.aggregate_lists(g, g.X, B, self._ws_of(B), cnt, agg)
        return agg, cnt

    def _graphs_for(self, B):
        """hipGraphs of the two collective-free segments for batch size B (static id / label buffers)."""
        gr = self._graphs.get(B)
        if gr is not None:
            return gr
        ids, lab = self.ids_buf[:B], self.lab_buf[:B]
        state = (self.theta.clone(), self.m.clone(), self.v.clone(), self.step_counter.clone())
        s = torch.cuda.Stream(self.dev)            # warm-up: kernel attributes, auxiliary streams, workspaces
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            self._seg_scores(ids, lab, B, True)
            self._seg_select(ids, lab, B, True)
            self._seg_dense(ids, lab, B)
        torch.cuda.current_stream(self.dev).wait_stream(s)
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
        ops, g, lib, _p, part = self.ops, self.g, self.lib, self.ops._p, self.part
        B = ids_local.numel()
        if use_graphs:
            gr = self._graphs_for(B)
            self.ids_buf[:B].copy_(ids_local)
            self.lab_buf[:B].copy_(labels)
            ids_local, labels = self.ids_buf[:B], self.lab_buf[:B]
        self._seg_scores(ids_local, labels, B, True)
        self._all_gather(self.s0_full, self.s0_send)
        prof = getattr(self, "_prof", None)
        timed = prof is not None and self._prof_step % self._prof_every == 0
        if prof is not None:
            self._prof_step += 1
        if timed:      # HIP events around the select segment (bench.py's roofline at N > 1), on the launching stream
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if use_graphs:
            gr["select"].replay()
        else:
            self._seg_select(ids_local, labels, B, True)
        if timed:
            ev[1].record()
            prof.append((ev[0], ev[1], ids_local.clone(), self.cnt[:self.g.R * B].clone()))
        self._exchange(B)
        if use_graphs:
            gr["dense"].replay()
        else:
            self._seg_dense(ids_local, labels, B)
        self._all_reduce(self.grad)
        c = self.cfg
        self._libmod.check(lib.pcg_adam_step(_p(self.theta), _p(self.m), _p(self.v), _p(self.grad), 1, self.n_params,
                                             _p(self.step_counter), c["lr"], 0.9, 0.999, 1e-8, c["weight_decay"], None, 1,
                                             ops._stream(self.dev)), "pcg_adam_step")

    def profile_select(self, every: int = 10):
        """Start collecting (start event, end event, ids, |set| counts) of every `every`-th step's select segment."""
        self._prof, self._prof_every, self._prof_step = [], every, 0
        return self._prof

    def pick_epoch(self, size: int, epoch: int) -> torch.Tensor:
        """this rank's share of the epoch's picks: local row numbers of owned training nodes."""
        return self.sampler.pick(size, epoch)

    def labels_of(self, ids_local: torch.Tensor) -> torch.Tensor:
        return self.labels_local[ids_local.long()]
