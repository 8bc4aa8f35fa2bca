"""Whole-step driver: every launch of one PC-GNN train step, back to back on one
stream, no torch ops and no host synchronisation in between - five launches:

    front_a [scores || plan 1 || train-pos keys || Adam of the previous step's gradient (all but the label classifier)]
    front_b [train-pos sort || plan 2]
    select_rows -> gather_chunks
    dense_step [sums of multi-chunk rows, forward, loss, backward partials, Adam of the label classifier]

(``pcg_step_front_train`` / ``pcg_choose_gather_planned`` / ``pcg_train_dense``); a deferred update is flushed
(``pcg_adam_flush``) before any public call returns unless the caller asks otherwise, so the parameters a caller
sees are always complete.  The same sequence is captured into hipGraphs (``torch.cuda.CUDAGraph``) - per batch, or a
whole epoch in one - so that a step costs no host work at all.

The model's parameters are re-pointed into ONE flat f32 buffer (order: see
``pcg_dense_step`` in include/pcgnn.h); the ``nn.Parameter`` objects, their names
and ``state_dict()`` stay exactly the reference's.

Reference loop replaced: src/model_handler.py:147-153 (and :305 of utils.py for
``predict``).
"""
import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib, ops
from .graph import DeviceGraph
from .model import PCALayer

_p = ops._p


class FusedPCGNN:
    def __init__(self, model: PCALayer, lr: float, weight_decay: float, betas=(0.9, 0.999), eps: float = 1e-8,
                 max_batch: int = 1024, global_batch_scale: int = 1, list_capacity: Optional[int] = None):
        lib = _lib.load()
        self.lib = lib
        self.model = model
        inter = model.inter1
        self.g: DeviceGraph = inter.graph()
        g = self.g
        self.dev = g.device
        self.F, self.E, self.R = g.feat_dim, inter.embed_dim, g.R
        self.lr, self.wd, self.betas, self.eps = lr, weight_decay, betas, eps
        self.lambda_1 = float(model.lambda_1)
        self.scale = global_batch_scale           # ranks sharing one global batch (loss is a mean over all of it)
        self.thresholds = list(inter.thresholds)
        self.rho = [a.rho for a in inter.intra_aggs]

        n = lib.pcg_dense_n_params(self.F, self.E, self.R)
        self.n_params = int(n)
        self.theta = torch.zeros(n, dtype=torch.float32, device=self.dev)
        named = dict(model.named_parameters())
        spec = [("weight", 0, 0), ("inter1.weight", 1, 0)] + \
               [(f"inter1.intra_agg{r + 1}.weight", 2, r) for r in range(self.R)] + \
               [("inter1.label_clf.weight", 3, 0), ("inter1.label_clf.bias", 4, 0)]
        self.views: Dict[str, torch.Tensor] = {}
        for name, which, rel in spec:
            p = named[name]
            off = lib.pcg_dense_param_offset(self.F, self.E, self.R, which, rel)
            view = self.theta[off:off + p.numel()].view(p.shape)
            view.copy_(p.data.to(self.dev))
            p.data = view                      # the Parameter now lives inside the flat buffer
            self.views[name] = view
        self.w_clf = self.views["inter1.label_clf.weight"]
        self.b_clf = self.views["inter1.label_clf.bias"]
        self.m = torch.zeros_like(self.theta)
        self.v = torch.zeros_like(self.theta)
        self.step_counter = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.grad = torch.zeros_like(self.theta)
        self.sync = torch.zeros(4, dtype=torch.int32, device=self.dev)    # [ticket, update pending, its slab count, -]
        self.n_rest = int(lib.pcg_dense_param_offset(self.F, self.E, self.R, 3, 0))   # parameters before the label classifier

        self.list_capacity = list_capacity      # entries of every workspace's selection list (None: worst case of the graph)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)   # ONE device status word for all workspaces
        self._graphs = {}
        self._ep_graphs = {}
        self._alloc(max_batch)
        self._prof = None          # bench.py: list of (start, end) events around the choose+aggregate launch
        self.last_counts = None

    # ------------------------------------------------------------------
    def _alloc(self, B: int):
        g, dev = self.g, self.dev
        if torch.cuda.is_current_stream_capturing():
            raise _lib.PcgnnLibraryError("batch larger than max_batch inside a graph capture: allocate before capturing")
        # every captured graph holds raw pointers into the buffers replaced below
        self._graphs.clear()
        self._ep_graphs.clear()
        self.maxB = B
        self.s0 = torch.empty(g.n_nodes, dtype=torch.float32, device=dev)
        self.keys = torch.empty(self.lib.pcg_pos_sort_capacity(g.n_pos), dtype=torch.int64, device=dev)
        self._ws_by_b = {B: ops.ChooseWorkspace(g, B, self.list_capacity, status=self.status)}
        self.agg = torch.empty(g.R, B, g.feat_dim, dtype=torch.float32, device=dev)
        self.cnt = torch.empty(g.R, B, dtype=torch.int32, device=dev)
        self.logits = torch.empty(B, 2, dtype=torch.float32, device=dev)
        self.center = torch.empty(B, 2, dtype=torch.float32, device=dev)
        self.row_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.slabs = torch.empty(self.lib.pcg_dense_n_tiles(B), self.n_params, dtype=torch.float32, device=dev)
        self.ids_buf = torch.zeros(B, dtype=torch.int32, device=dev)
        self.lab_buf = torch.zeros(B, dtype=torch.int32, device=dev)

    def _stream(self):
        return ops._stream(self.dev)

    # ------------------------------------------------------------------
    def _enqueue_scores(self, train_flag):
        """label-aware score table + per-step sort of the train positives."""
        g = self.g
        ops.score_table(g, self.w_clf, self.b_clf, out=self.s0)
        return ops.pos_sort(g, self.s0, self.keys) if (train_flag and g.n_pos) else None

    def _ws(self, B):
        ws = self._ws_by_b.get(B)
        if ws is None:      # the workspace layout depends on the batch size: one per size
            ws = self._ws_by_b[B] = ops.ChooseWorkspace(self.g, B, self.list_capacity, status=self.status)
        return ws

    def _enqueue_front(self, ids, labels, B, train_flag):
        """scores + train-pos sort + this batch's plan in two launches (pcg_step_front)."""
        g = self.g
        return ops.step_front(g, self.w_clf, self.b_clf, self.s0, self.keys if (train_flag and g.n_pos) else None,
                              ids, labels if train_flag else None, self.thresholds, self.rho, train_flag, self._ws(B))

    def _enqueue_front_train(self, ids, labels, B):
        """the front of a training step with the previous step's deferred Adam update beside the score pass."""
        g, ws = self.g, self._ws(B)
        thr, rhos = ops._host_arrays(g, self.thresholds, self.rho)
        b1, b2 = self.betas
        _lib.check(self.lib.pcg_step_front_train(
            g.desc_ref(), _p(self.theta), _p(self.m), _p(self.v), self.E, _p(self.s0), _p(self.keys) if g.n_pos else None,
            _p(ids), _p(labels), B, thr, rhos, 0, _p(ws.buf), ws.list_capacity, _p(ws.status), _p(self.slabs),
            _p(self.step_counter), _p(self.sync), self.lr, b1, b2, self.eps, self.wd, self._stream()), "pcg_step_front_train")
        return self.keys if g.n_pos else None

    def _enqueue_choose(self, ids, labels, B, keys, train_flag, planned=False, combine=True):
        """select + gather (+ combine unless the dense kernel will add up the partial sums itself)."""
        g = self.g
        agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
        cnt = self.cnt.view(-1)[:g.R * B].view(g.R, B)
        timed = self._prof is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if combine:
            ops.choose_aggregate(g, ids, labels if train_flag else None, self.s0, keys, self.thresholds, self.rho,
                                 train_flag, ws=self._ws(B), agg=agg, cnt=cnt, planned=planned)
        else:
            assert planned
            ws = self._ws(B)
            thr, rhos = ops._host_arrays(g, self.thresholds, self.rho)
            _lib.check(self.lib.pcg_choose_gather_planned(
                g.desc_ref(), _p(ids), _p(labels if train_flag else None), B, _p(self.s0), None, _p(keys), thr, rhos,
                1 if train_flag else 0, 0, _p(agg), agg.stride(-2), _p(cnt), _p(ws.buf), ws.list_capacity, _p(ws.status),
                self._stream()), "pcg_choose_gather_planned")
        if timed:
            ev[1].record()
            self._prof.append(ev)
        self.last_counts = cnt
        return agg, cnt

    def _enqueue_sample(self, ids, labels, B, train_flag):
        """score table + train-pos sort + plan (two launches), then select + aggregate, for one batch (views sized to B)."""
        keys = self._enqueue_front(ids, labels, B, train_flag)
        return self._enqueue_choose(ids, labels, B, keys, train_flag, planned=True)

    def _enqueue_dense(self, ids, labels, B, agg, train: bool, combined=None):
        g = self.g
        _lib.check(self.lib.pcg_dense_step(
            g.desc_ref(), _p(self.theta), self.E, _p(ids), _p(labels), B, _p(agg), agg.stride(1), self.lambda_1,
            1.0 / (B * self.scale), _p(self.logits), _p(self.center), _p(combined),
            _p(self.row_loss) if labels is not None else None, _p(self.slabs) if train else None,
            _p(self.step_counter) if train else None, self._stream()), "pcg_dense_step")

    def _enqueue_tail(self, ids, labels, B, agg, train: bool, combined=None):
        """dense tail reading the gather's partial sums (no combine launch); training: + the label classifier's Adam
        in the same launch, the update of the other parameters left pending (flush() or the next front applies it)."""
        g, ws = self.g, self._ws(B)
        cnt = self.cnt.view(-1)[:g.R * B]
        b1, b2 = self.betas
        _lib.check(self.lib.pcg_train_dense(
            g.desc_ref(), _p(self.theta), _p(self.m), _p(self.v), self.E, _p(ids), _p(labels), B, _p(agg), agg.stride(1),
            _p(cnt), _p(ws.buf), ws.list_capacity, self.lambda_1, 1.0 / (B * self.scale), _p(self.logits), _p(self.center),
            _p(combined), _p(self.row_loss) if labels is not None else None, _p(self.slabs) if train else None,
            _p(self.step_counter) if train else None, _p(self.sync), self.lr, b1, b2, self.eps, self.wd, 1 if train else 0,
            self._stream()), "pcg_train_dense")

    def flush(self):
        """Apply a deferred Adam update now (no-op on the device if none is pending).  Enqueued, not synchronised."""
        b1, b2 = self.betas
        _lib.check(self.lib.pcg_adam_flush(
            _p(self.theta), _p(self.m), _p(self.v), _p(self.slabs), 0, self.n_params, self.n_rest, _p(self.step_counter),
            _p(self.sync), self.lr, b1, b2, self.eps, self.wd, self._stream()), "pcg_adam_flush")

    def _enqueue_step(self, ids, labels, B, defer: bool):
        """the five launches of one training step (+ the flush unless deferred)."""
        keys = self._enqueue_front_train(ids, labels, B)
        agg, _ = self._enqueue_choose(ids, labels, B, keys, True, planned=True, combine=False)
        self._enqueue_tail(ids, labels, B, agg, True)
        if not defer:
            self.flush()

    def _enqueue_adam(self, B, apply=True, want_grad=False, from_grad=False):
        """from_grad: the (all-reduced) flat gradient in self.grad is the single slab."""
        b1, b2 = self.betas
        slabs, n_slabs = (self.grad, 1) if from_grad else (self.slabs, self.lib.pcg_dense_n_tiles(B))
        _lib.check(self.lib.pcg_adam_step(
            _p(self.theta), _p(self.m), _p(self.v), _p(slabs), n_slabs, self.n_params,
            _p(self.step_counter), self.lr, b1, b2, self.eps, self.wd, _p(self.grad) if want_grad else None,
            1 if apply else 0, self._stream()), "pcg_adam_step")

    # ------------------------------------------------------------------
    def train_step(self, ids: torch.Tensor, labels: torch.Tensor, allreduce=None, defer: bool = False):
        """zero_grad + loss + backward + Adam step for one batch (model_handler.py:149-153).
        ids / labels: int32 device tensors.  Nothing is returned and nothing syncs;
        ``last_loss()`` reads the batch loss afterwards.  ``allreduce(flat_grad)`` (data-parallel
        ranks) is called between the gradient reduction and the Adam update.  ``defer``: leave the Adam update of
        everything but the label classifier to the next step's front (or ``flush()``) - inside an epoch."""
        B = ids.numel()
        if B == 0:
            return
        if B > self.maxB:
            self.flush()
            self._alloc(B)
        self._lastB = B
        if allreduce is None:
            self._enqueue_step(ids, labels, B, defer)
            return
        # data-parallel ranks: gradient of the local batch -> all-reduce -> the same Adam on every rank
        self.flush()
        agg, _ = self._enqueue_sample(ids, labels, B, True)
        self._enqueue_dense(ids, labels, B, agg, True)
        self._enqueue_adam(B, apply=False, want_grad=True)
        allreduce(self.grad)
        self._enqueue_adam(B, apply=True, from_grad=True)

    def train_step_graph(self, ids: torch.Tensor, labels: torch.Tensor, timed: bool = False):
        """Same as train_step through captured hipGraphs (one set per batch size): one graph launch
        per step.  timed=True (bench.py, every Nth step) replays the step as [scores graph] ->
        eager choose+aggregate bracketed by HIP events -> [dense+Adam graph]; same kernels, same order."""
        B = ids.numel()
        if B == 0:
            return
        if B > self.maxB:
            self.flush()
            self._alloc(B)
        self._lastB = B
        gr = self._graphs.get(B)
        if gr is None:
            gr = self._capture(B)
        self.ids_buf[:B].copy_(ids)
        self.lab_buf[:B].copy_(labels)
        if timed and self._prof is not None:
            gr["pre"].replay()
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            gr["choose"].replay()          # the same launches as inside "full", as a graph of their own
            ev[1].record()
            self._prof.append(ev)
            self.last_counts = self.cnt.view(-1)[:self.g.R * B].view(self.g.R, B)
            gr["post"].replay()
        else:
            gr["full"].replay()

    def _capture_graphs(self, fns, warm=None):
        """Warm up (kernel attributes, per-batch-size workspaces: nothing may allocate inside a capture) on a side
        stream, capture every fn into a hipGraph of its own, and put the optimizer state back.  A deferred Adam update
        of real steps is applied first (the saved state then includes it); the warm-up's own is flushed and undone."""
        self.flush()
        state = (self.theta.clone(), self.m.clone(), self.v.clone(), self.step_counter.clone())
        prof, self._prof = self._prof, None
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for fn in (warm if warm is not None else fns):
                fn()
            self.flush()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        graphs = []
        for fn in fns:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                fn()
            graphs.append(gr)
        for dst, src in zip((self.theta, self.m, self.v, self.step_counter), state):
            dst.copy_(src)
        self._prof = prof
        return graphs

    def _capture(self, B):
        ids, lab = self.ids_buf[:B], self.lab_buf[:B]
        keys = self.keys if self.g.n_pos else None
        g = self.g
        agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)

        def pre():
            self._enqueue_front_train(ids, lab, B)

        def post():
            self._enqueue_tail(ids, lab, B, agg, True)
            self.flush()

        def choose():
            self._enqueue_choose(ids, lab, B, keys, True, planned=True, combine=False)

        def full():
            pre()
            choose()
            post()

        names = ("full", "pre", "choose", "post")
        graphs = dict(zip(names, self._capture_graphs([full, pre, choose, post], warm=[full])))
        self._graphs[B] = graphs
        return graphs

    # -- whole-epoch path: ids / labels of an epoch live in static buffers, one graph per batch slot ------
    def begin_epoch(self, ids: torch.Tensor, labels: torch.Tensor, batch_size: int):
        """Stage an epoch's (already shuffled) ids and labels; afterwards ``epoch_step(b)`` is exactly one
        graph launch - no copies, no indexing kernels (model_handler.py:142-148 slices a Python list here)."""
        n = ids.numel()
        ep_ids, ep_lab = self.stage_epoch(n, batch_size)
        ep_ids.copy_(ids)
        ep_lab.copy_(labels)

    def epoch_step(self, b: int, defer: bool = False):
        """Batch b of the staged epoch as one graph replay.  defer: as inside epoch_run - the Adam update of everything but
        the label classifier is left to the next batch's front launch (the epoch's last batch flushes it), so the
        parameters are complete only after the last batch or a flush()."""
        lo = b * self._ep_bs
        B = min(self._ep_bs, self._ep_n - lo)
        if B <= 0:
            return
        self._lastB = B
        defer = defer and lo + B < self._ep_n
        key = (lo, B, "deferred") if defer else (lo, B)
        gr = self._ep_graphs.get(key)
        if gr is None:
            ids, lab = self._ep_ids[lo:lo + B], self._ep_lab[lo:lo + B]
            gr = self._capture_graphs([lambda: self.train_step(ids, lab, defer=defer)])[0]
            self._ep_graphs[key] = gr
        gr.replay()

    def epoch_step_timed(self, b: int, eager: bool = True):
        """Batch b of the staged epoch with HIP events around the select + gather call (appended to ``_prof``): the same
        kernels in the same order as one step of ``epoch_run``, reading the staged ids in place (no copies, no label
        gather).  eager: the five kernels launched one by one with the two event records between them (the host stays
        ahead of the GPU, so the events bracket the two kernels and nothing else); otherwise three graphs - front | select +
        gather | dense - whose launch latency lands inside the bracket."""
        lo = b * self._ep_bs
        B = min(self._ep_bs, self._ep_n - lo)
        if B <= 0:
            return
        self._lastB = B
        key = (lo, B, "timed")
        grs = self._ep_graphs.get(key)
        ids, lab = self._ep_ids[lo:lo + B], self._ep_lab[lo:lo + B]
        g = self.g
        if eager:
            agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
            keys = self.keys if g.n_pos else None
            self._enqueue_front_train(ids, lab, B)
            self._enqueue_choose(ids, lab, B, keys, True, planned=True, combine=False)     # (records the two events itself)
            self._enqueue_tail(ids, lab, B, agg, True)
            if lo + B >= self._ep_n:              # the epoch's last batch: nothing follows that would apply the deferred update
                self.flush()
            self.last_counts = self.cnt.view(-1)[:g.R * B].view(g.R, B)
            return
        if grs is None:
            agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
            keys = self.keys if g.n_pos else None
            last = lo + B >= self._ep_n           # the epoch's last batch: nothing follows that would apply the deferred update
            parts = (lambda: self._enqueue_front_train(ids, lab, B),
                     lambda: self._enqueue_choose(ids, lab, B, keys, True, planned=True, combine=False),
                     lambda: (self._enqueue_tail(ids, lab, B, agg, True), self.flush() if last else None))
            grs = self._capture_graphs(list(parts))
            self._ep_graphs[key] = grs
        grs[0].replay()
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        grs[1].replay()
        ev[1].record()
        if self._prof is not None:
            self._prof.append(ev)
        self.last_counts = self.cnt.view(-1)[:g.R * B].view(g.R, B)
        grs[2].replay()

    def stage_epoch(self, n: int, batch_size: int):
        """Static id / label buffers of an epoch of n picks (filled by begin_epoch or by a sampler)."""
        if getattr(self, "_ep_ids", None) is None or self._ep_ids.numel() < n:
            self._ep_ids = torch.zeros(n, dtype=torch.int32, device=self.dev)
            self._ep_lab = torch.zeros(n, dtype=torch.int32, device=self.dev)
            self._ep_graphs.clear()
        self._ep_n, self._ep_bs = n, batch_size
        return self._ep_ids[:n], self._ep_lab[:n]

    def epoch_run(self, n_steps: Optional[int] = None, sample=None):
        """All batches of the staged epoch as ONE graph launch: the host latency between two graph launches (~8 us)
        is paid once per epoch instead of once per batch.  ``sample()``, if given, is enqueued (and captured) first: it
        fills the staged id / label buffers on the device (pick + shuffle + labels), so a replay is a whole new epoch."""
        nb = -(-self._ep_n // self._ep_bs)
        n_steps = nb if n_steps is None else min(n_steps, nb)
        key = ("epoch", self._ep_n, self._ep_bs, n_steps, sample is not None)
        gr = self._ep_graphs.get(key)
        if gr is None:
            def run():
                for b in range(n_steps):
                    lo = b * self._ep_bs
                    B = min(self._ep_bs, self._ep_n - lo)
                    self.train_step(self._ep_ids[lo:lo + B], self._ep_lab[lo:lo + B], defer=True)
                self.flush()
            def sampled_run():
                if sample is not None:
                    sample()
                run()
            gr = self._capture_graphs([sampled_run], warm=[run])[0]   # (the warm-up leaves the staged ids as they are)
            self._ep_graphs[key] = gr
        self._lastB = min(self._ep_bs, self._ep_n - (n_steps - 1) * self._ep_bs)
        gr.replay()
        return n_steps

    def check(self):
        """Raise if any batch since the last check did not fit its selection list (the kernels then select nothing and
        only set the device status word).  Reads one word: synchronises - call it where the host waits anyway
        (``last_loss``, the end of an evaluation pass or of an epoch, after a timed region)."""
        st = int(self.status.item())
        if st & _lib.PCG_ST_SEL_OVERFLOW:
            self.status.zero_()
            raise _lib.PcgnnLibraryError("selection list overflow: a batch needed more list entries than the workspace "
                                         "holds - raise FusedPCGNN(list_capacity=...)")

    def last_loss(self) -> torch.Tensor:
        self.check()
        B = self._lastB
        return self.row_loss[:B].sum() / (B * self.scale)

    def gradients(self, ids: torch.Tensor, labels: torch.Tensor) -> Dict[str, torch.Tensor]:
        """loss.backward() without the optimizer step: per-parameter gradients (parity tests)."""
        B = ids.numel()
        self.flush()
        agg, _ = self._enqueue_sample(ids, labels, B, True)
        self._enqueue_dense(ids, labels, B, agg, True)
        self.step_counter -= 1                   # dense_step counted a step that is not taken
        self._enqueue_adam(B, apply=False, want_grad=True)
        self._lastB = B
        out = {}
        for name, view in self.views.items():
            off = view.storage_offset()
            out[name] = self.grad[off:off + view.numel()].view(view.shape).clone()
        return out

    def predict(self, ids: torch.Tensor, labels: Optional[torch.Tensor] = None, train_flag: bool = False,
                want_combined: bool = False):
        """forward only -> (gnn logits [B,2], label-aware logits [B,2][, combined [B,E]])
        (PCALayer.forward, model.py:34-39; utils.py:305 calls it with train_flag=False)."""
        B = ids.numel()
        self.flush()
        if B > self.maxB:
            self._alloc(B)
        keys = self._enqueue_front(ids, labels, B, train_flag)
        agg, _ = self._enqueue_choose(ids, labels, B, keys, train_flag, planned=True, combine=False)
        comb = torch.empty(B, self.E, dtype=torch.float32, device=self.dev) if want_combined else None
        self._enqueue_tail(ids, None, B, agg, False, combined=comb)
        res = (self.logits[:B].clone(), self.center[:B].clone())
        return res + (comb,) if want_combined else res
