"""Whole-step driver: every launch of one PC-GNN train step, back to back on one
stream, no torch ops and no host synchronisation in between - three launches:

    select_rows          [the train positives' sort by its first workgroups || the label classifier's own step for this batch
                          (one workgroup: forward, loss gradient, Adam) || every row's selection]
    gather_train_kernel  [gather || Adam of the previous step's gradient (all but the label classifier) || the NEXT step's score
                          pass and train-pos keys, with the classifier the select launch has just stepped]
    dense_step           [sums of multi-chunk rows, forward, loss, backward partials (per-tile slabs)]

(``pcg_choose_gather_train`` / ``pcg_train_dense(adam_clf = 2)``; the first step after anything else has run is preceded by one
score launch, ``pcg_step_scores``).  Why this order is legal: the label classifier's parameters get gradient from nothing but
the centres' feature rows and labels, so its step does not have to wait for the dense kernel - and the scores the next batch is
selected by can be formed beside this batch's gather instead of at the head of the next step.  ``clf_next`` is that classifier,
one step ahead of theta's copy (which the current step's loss term is computed with); ``flush()`` makes them equal;
``params_changed()`` after parameters were written from outside.

The PLAN of a batch (row records, list
offsets, tier queues, the gather's chunk table) depends only on the batch's ids, labels and the CSR degrees - not on any
parameter - so it is not part of a step: ``pcg_plan_batches`` plans every batch of an epoch in ONE launch right after the
sampler (one plan slot per batch; the selection list and the partial sums - the data part - are shared).  A deferred Adam
update is flushed (``pcg_adam_flush``) before any public call returns unless the caller asks otherwise, so the parameters a
caller sees are always complete.  The same sequence is captured into hipGraphs (``torch.cuda.CUDAGraph``) - per batch, or a
whole epoch in one - so that a step costs no host work at all.

The model's parameters are re-pointed into ONE flat f32 buffer (order: see
``pcg_dense_step`` in include/pcgnn.h); the ``nn.Parameter`` objects, their names
and ``state_dict()`` stay exactly the reference's.

Reference loop replaced: src/model_handler.py:147-153 (and :305 of utils.py for
``predict``).
"""
import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib, ops
from .graph import DeviceGraph
from .model import PCALayer

_p = ops._p


def default_list_capacity(g: DeviceGraph, B: int, max_list_bytes: int = 8 << 30):
    """Worst case of the graph for a batch of B (every centre being the largest hub), clipped to ``max_list_bytes``:
    kept <= deg, minority m = int(ceil(deg / 2) * rho) <= 2 * deg for rho <= 4 (and <= n_pos), +1 self.
    Returns (entries, clipped)."""
    per_row = g.max_degree + min(2 * max(g.max_degree, 1), g.n_pos) + 1
    worst = per_row * g.R * max(B, 1)
    cap = min(worst, max_list_bytes // 4, (1 << 31) - 1)
    return int(max(cap, 1)), cap < worst


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
        # [ticket of the dense kernel, update pending, its slab count, group counter of the select kernel's in-kernel sort | its
        #  rank accumulators and group tickets]
        self.sync = torch.zeros(int(lib.pcg_sync_words_count()), dtype=torch.int32, device=self.dev)
        self.n_rest = int(lib.pcg_dense_param_offset(self.F, self.E, self.R, 3, 0))   # parameters before the label classifier
        # The label classifier is stepped on its own, one step ahead of the rest (pcg_choose_gather_train): clf_next is the
        # classifier the NEXT select launch scores by; theta's copy is the one the current step's loss is computed with.  flush()
        # makes them equal.  _fresh: s0 / the unsorted train-pos keys hold the scores of clf_next (a training step with
        # score_next leaves them so); anything else that touches s0, the keys or the parameters clears it.
        self.clf_next = self.theta[self.n_rest:].clone()
        self._fresh = False
        # model.load_state_dict writes the parameters behind the engine's back (they are views into theta): a deferred update
        # is applied first, the stepped classifier and the score table are taken from theta again afterwards
        model.register_load_state_dict_pre_hook(lambda *a, **k: self.flush())
        model.register_load_state_dict_post_hook(lambda *a, **k: self.params_changed())

        # Score only the rows a batch's selection can read (its centres and their neighbours: a byte map per batch, built per epoch
        # beside the plans) instead of the whole table - worth it when the table is far larger than what a batch touches
        # (PCG_TOUCHED=0/1 overrides; default: tables of 512 MB and more.  Measured, power-law graphs, batch 4096: 10 M nodes
        #  (1.28 GB): score pass 232 -> 63 us per step for ~35 us per step of map building; 2 M nodes (256 MB): 48 -> 25 us for
        #  ~30 us - not worth it there)
        env = os.environ.get("PCG_TOUCHED")
        self.touched_on = (env == "1") if env in ("0", "1") else g.n_nodes * g.X.stride(0) * 4 >= (512 << 20)
        self._touch_stride = int(lib.pcg_touched_bytes(g.n_nodes)) if self.touched_on else 0
        self._list_capacity_arg = list_capacity  # entries of the selection list (None: worst case of the graph)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)   # ONE device status word
        self._graphs = {}
        self._ep_graphs = {}
        self._ep_sets = None       # two sets of epoch buffers: ids | labels | plan slots (stage_epoch)
        self._cur, self._cur_ready, self._ep_stride = 0, False, 0
        self._alloc(max_batch)
        self._prof = None          # bench.py: list of (start, end) events around the choose+aggregate launch
        self.last_counts = None
        thr, rhos = ops._host_arrays(g, self.thresholds, self.rho)
        self._thr, self._rhos = thr, rhos

    # ------------------------------------------------------------------
    def _alloc(self, B: int):
        g, dev, lib = self.g, self.dev, self.lib
        if torch.cuda.is_current_stream_capturing():
            raise _lib.PcgnnLibraryError("batch larger than max_batch inside a graph capture: allocate before capturing")
        # every captured graph holds raw pointers into the buffers replaced below
        self._graphs.clear()
        self._ep_graphs.clear()
        self._ep_sets = None
        self._fresh = False
        self.maxB = B
        if self._list_capacity_arg is None:
            self.list_capacity, self.clipped = default_list_capacity(g, B)
        else:
            self.list_capacity, self.clipped = int(max(self._list_capacity_arg, 1)), False
        self.s0 = torch.empty(g.n_nodes, dtype=torch.float32, device=dev)
        self.keys = torch.empty(lib.pcg_pos_sort_capacity(g.n_pos), dtype=torch.int64, device=dev)
        nbytes = lib.pcg_choose_data_bytes(g.desc_ref(), B, self.list_capacity)
        if nbytes < 0:
            raise _lib.PcgnnLibraryError("pcg_choose_data_bytes rejected the arguments")
        self.data = torch.zeros(int(nbytes), dtype=torch.uint8, device=dev)     # list | partial sums | key scratch: shared by every plan
        self._plans = {}                                                        # batch size -> a single plan slot (calls outside an epoch)
        self._touch_one = None                                                  # ... and their byte map of touched rows
        self.agg = torch.empty(g.R, B, g.feat_dim, dtype=torch.float32, device=dev)
        self.cnt = torch.empty(g.R, B, dtype=torch.int32, device=dev)
        self.logits = torch.empty(B, 2, dtype=torch.float32, device=dev)
        self.center = torch.empty(B, 2, dtype=torch.float32, device=dev)
        self.row_loss = torch.zeros(B, dtype=torch.float32, device=dev)
        self.slabs = torch.empty(max(lib.pcg_dense_n_tiles(B), 8), self.n_params, dtype=torch.float32, device=dev)
        # the training step's dense kernel leaves no gradient slabs: it leaves the step's activations / activation gradients,
        # transposed ([rows][act_ld], a batch row per column), and the weight gradients are GEMMs over the batch riding in the next
        # step's gather launch (pcg_train_dense(adam_clf = 3), wgrad.h).  (slabs: the gradient-only paths - gradients(via="slabs"),
        # data-parallel all-reduce - and the classifier step's scratch)
        self.act_ld = 16 * int(lib.pcg_dense_n_tiles(B))
        self.acts = torch.zeros(int(lib.pcg_wgrad_act_rows(self.F, self.E, self.R)), self.act_ld, dtype=torch.float32, device=dev)
        self.wg_scratch = torch.zeros(int(lib.pcg_wgrad_scratch_bytes(self.F, self.E, self.R, self.act_ld)) // 4, dtype=torch.float32, device=dev)
        # One dense workgroup per 16 rows leaves most CUs of a batch of <= ~3000 rows idle: the dense launch then also SORTS the
        # next step's train-pos keys (formed beside the gather before it), and the select launch that follows is told so - it
        # sorts nothing, no row waits for the sort, a positive hub row's window search runs beside its key pass.  The engine's
        # invariant in this mode: whenever s0 / the keys are fresh (_fresh) the keys are SORTED (a refresh sorts them itself).
        self.presort = bool(lib.pcg_dense_sorts_keys(B, g.n_pos)) and os.environ.get("PCG_PRESORT", "1") != "0"
        self.ids_buf = torch.zeros(B, dtype=torch.int32, device=dev)
        self.lab_buf = torch.zeros(B, dtype=torch.int32, device=dev)

    def _stream(self):
        return ops._stream(self.dev)

    def _plan_bytes(self, B):
        n = self.lib.pcg_choose_plan_bytes(self.g.desc_ref(), B, self.list_capacity)
        if n < 0:
            raise _lib.PcgnnLibraryError("pcg_choose_plan_bytes rejected the arguments")
        return int(n)

    def _plan_slot(self, B) -> torch.Tensor:
        """the plan slot of calls that are not part of a staged epoch (one per batch size: the layout depends on it)"""
        buf = self._plans.get(B)
        if buf is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.PcgnnLibraryError("a new batch size inside a graph capture: warm up before capturing")
            buf = self._plans[B] = torch.zeros(self._plan_bytes(B), dtype=torch.uint8, device=self.dev)
        return buf

    # ------------------------------------------------------------------
    def _enqueue_plan(self, ids, labels, n_total, B, plans: torch.Tensor, stride: int, train_flag, bump_counter=None, n_epochs: int = 1):
        """the plans of the batches ids[s * B : (s + 1) * B] (s < ceil(n_total / B)) - of every one of n_epochs epochs of n_total
        picks each -, one launch"""
        _lib.check(self.lib.pcg_plan_epochs(
            self.g.desc_ref(), _p(ids), _p(labels if train_flag else None), n_total, n_epochs, B, self._thr, self._rhos,
            1 if train_flag else 0, 0, _p(plans), stride, self.list_capacity, _p(self.status), _p(bump_counter), self._stream()),
            "pcg_plan_epochs")

    def _enqueue_plan_one(self, ids, labels, B, train_flag) -> int:
        """plan of ONE batch into this batch size's own slot (+ its touched-row map); returns the slot's address"""
        slot = self._plan_slot(B)
        self._enqueue_plan(ids, labels, B, B, slot, slot.numel(), train_flag)
        if self.touched_on and train_flag:
            if self._touch_one is None:
                if torch.cuda.is_current_stream_capturing():
                    raise _lib.PcgnnLibraryError("first use of the touched-row map inside a graph capture: warm up before capturing")
                self._touch_one = torch.zeros(self._touch_stride, dtype=torch.uint8, device=self.dev)
            self._enqueue_mark(ids, B, B, self._touch_one, slot.data_ptr(), slot.numel())
        return slot.data_ptr()

    def _enqueue_mark(self, ids, n_total, B, maps: torch.Tensor, plans: int, stride: int):
        """byte maps of the rows each batch's selection can read, from the batches' plans (`plans`: address of the first batch's
        slot): two launches for all batches (pcg_mark_touched_planned)"""
        _lib.check(self.lib.pcg_mark_touched_planned(self.g.desc_ref(), _p(ids), n_total, B, C.c_void_p(plans), stride, self.list_capacity,
                                                     _p(maps), self._touch_stride, self._stream()), "pcg_mark_touched_planned")

    def _enqueue_scores(self, train_flag):
        """label-aware score table + per-step sort of the train positives (the calls of their own: evaluation, parity)."""
        g = self.g
        self._fresh = False                         # (s0 / the keys are overwritten: theta's classifier, sorted keys)
        ops.score_table(g, self.w_clf, self.b_clf, out=self.s0)
        return ops.pos_sort(g, self.s0, self.keys) if (train_flag and g.n_pos) else None

    def _enqueue_scores_train(self, touched: Optional[int] = None):
        """the front of a training step: score pass || train-pos keys (unsorted) || the previous step's deferred Adam update.
        touched: address of the batch's byte map - only the rows it marks are scored."""
        g = self.g
        b1, b2 = self.betas
        self._fresh = False                         # (the four-launch step: scores of theta's classifier, every step)
        _lib.check(self.lib.pcg_step_scores_train(
            g.desc_ref(), _p(self.theta), _p(self.m), _p(self.v), self.E, _p(self.s0), _p(self.keys) if g.n_pos else None,
            _p(self.slabs), _p(self.step_counter), _p(self.sync), self.lr, b1, b2, self.eps, self.wd,
            None if touched is None else C.c_void_p(touched), self._stream()), "pcg_step_scores_train")
        return self.keys if g.n_pos else None

    def params_changed(self):
        """Tell the engine that the parameters were written from outside (a state dict loaded into the flat buffer's views):
        the stepped copy of the label classifier and the score table are taken from theta again."""
        self.flush()
        self.clf_next.copy_(self.theta[self.n_rest:])
        self._fresh = False

    def _enqueue_refresh(self, touched: Optional[int] = None):
        """scores + unsorted train-pos keys of the classifier the next select launch uses (clf_next), one launch (no update of
        anything): the first training step after anything else has run, and every first step of an epoch of a touched-rows
        engine (whose previous step could not know this batch's map)."""
        g = self.g
        F = self.F
        _lib.check(self.lib.pcg_step_scores(
            g.desc_ref(), _p(self.clf_next), C.c_void_p(self.clf_next.data_ptr() + 8 * F), 0, g.n_nodes, _p(self.s0), None,
            _p(self.keys) if g.n_pos else None, -1, _p(self.sync), None if touched is None else C.c_void_p(touched),
            self._stream()), "pcg_step_scores")
        if self.presort:                             # (the steps that follow are told the keys are sorted)
            ops.pos_sort(g, self.s0, self.keys)
        self._fresh = True

    def _enqueue_choose_train(self, ids, labels, B, plan: int, score_next: bool, next_touched: Optional[int] = None):
        """select (+ the label classifier's step for this batch) and gather (+ the deferred update of the other parameters,
        + the next step's scores if score_next) of a training step: pcg_choose_gather_train."""
        g = self.g
        agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
        cnt = self.cnt.view(-1)[:g.R * B].view(g.R, B)
        b1, b2 = self.betas
        timed = self._prof is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _lib.check(self.lib.pcg_choose_gather_train(
            g.desc_ref(), _p(ids), _p(labels), B, _p(self.s0), _p(self.keys) if g.n_pos else None, self._thr, self._rhos, 0,
            _p(agg), agg.stride(-2), _p(cnt), _p(self.data), C.c_void_p(plan), self.list_capacity, _p(self.status), _p(self.sync),
            _p(self.theta), _p(self.m), _p(self.v), self.E, _p(self.clf_next), _p(self.slabs), _p(self.step_counter),
            self.lambda_1, 1.0 / (B * self.scale), self.lr, b1, b2, self.eps, self.wd, 1 if score_next else 0,
            None if next_touched is None else C.c_void_p(next_touched), _p(self.acts), self.act_ld, _p(self.wg_scratch),
            1 if self.presort else 0, self._stream()), "pcg_choose_gather_train")
        if timed:
            ev[1].record()
            self._prof.append(ev)
        self.last_counts = cnt
        self._fresh = bool(score_next)
        return agg, cnt

    def _enqueue_choose(self, ids, labels, B, keys, train_flag, plan: int, sort_in_kernel: bool):
        """select + gather over the batch's plan (rows of several chunks are left as partial sums for the dense kernel).
        sort_in_kernel: the launch follows ``_enqueue_scores_train`` - `keys` holds this step's UNSORTED train-pos keys (the
        select kernel sorts them itself unless there are too many: then they are sorted already) and the launch clears the
        "deferred update pending" word."""
        g = self.g
        agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
        cnt = self.cnt.view(-1)[:g.R * B].view(g.R, B)
        timed = self._prof is not None and not torch.cuda.is_current_stream_capturing()
        if timed:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        _lib.check(self.lib.pcg_choose_gather_planned(
            g.desc_ref(), _p(ids), _p(labels if train_flag else None), B, _p(self.s0), None, _p(keys), self._thr, self._rhos,
            1 if train_flag else 0, 0, _p(agg), agg.stride(-2), _p(cnt), _p(self.data), C.c_void_p(plan), self.list_capacity,
            _p(self.status), _p(self.sync) if sort_in_kernel else None, self._stream()), "pcg_choose_gather_planned")
        if timed:
            ev[1].record()
            self._prof.append(ev)
        self.last_counts = cnt
        return agg, cnt

    def _enqueue_tail(self, ids, labels, B, agg, plan: int, train: bool, combined=None, adam_clf: Optional[bool] = None):
        """dense tail reading the gather's partial sums (no combine launch); training: the gradient slabs are left pending
        (adam_clf 3, the default: no slabs at all - the kernel leaves its transposed activations and the next step's gather launch
        or flush() runs the weight-gradient GEMMs + Adam; 2: the same with per-tile gradient slabs; either way the label
        classifier has been stepped by this step's select launch).  adam_clf=0 (training): gradient slabs only - the caller
        reduces / all-reduces them itself; 1: the label classifier's Adam by the kernel's last workgroup (the four-launch step of
        pcg_step_scores_train); 4: transposed activations only, nothing marked as waiting (gradients())."""
        g = self.g
        cnt = self.cnt.view(-1)[:g.R * B]
        b1, b2 = self.betas
        adam_clf = (3 if train else 0) if adam_clf is None else int(adam_clf)
        _lib.check(self.lib.pcg_train_dense(
            g.desc_ref(), _p(self.theta), _p(self.m), _p(self.v), self.E, _p(ids), _p(labels), B, _p(agg), agg.stride(1),
            _p(cnt), _p(self.data), C.c_void_p(plan), self.list_capacity, self.lambda_1, 1.0 / (B * self.scale), _p(self.logits),
            _p(self.center), _p(combined), _p(self.row_loss) if labels is not None else None, _p(self.slabs) if train else None,
            _p(self.step_counter) if train else None, _p(self.sync), self.lr, b1, b2, self.eps, self.wd, adam_clf,
            _p(self.acts) if adam_clf in (3, 4) else None, self.act_ld,
            # (the next step's keys: formed by the gather launch before this one if it scored ahead - then sorted here)
            _p(self.keys) if (adam_clf == 3 and self.presort and self._fresh) else None, self._stream()), "pcg_train_dense")
        if adam_clf == 1:                           # (the four-launch step updates theta's classifier in place)
            self.clf_next.copy_(self.theta[self.n_rest:])

    def flush(self):
        """Apply a deferred Adam update now (no-op on the device if none is pending).  Enqueued, not synchronised."""
        b1, b2 = self.betas
        _lib.check(self.lib.pcg_adam_flush(
            _p(self.theta), _p(self.m), _p(self.v), _p(self.slabs), 0, self.n_params, self.n_rest, _p(self.step_counter),
            _p(self.sync), self.lr, b1, b2, self.eps, self.wd, _p(self.clf_next), _p(self.acts), self.act_ld, self.F, self.E, self.R,
            _p(self.wg_scratch), self._stream()), "pcg_adam_flush")

    def _enqueue_step(self, ids, labels, B, plan: int, defer: bool, touched: Optional[int] = None,
                      next_touched: Optional[int] = None):
        """the three launches of one training step over an existing plan (+ the flush unless deferred; + a score launch first
        if the scores at hand are not those of the current classifier).  touched / next_touched: the byte maps of this batch
        and of the one that follows (a touched-rows engine scores the next step's rows only if it is told which they are)."""
        if not self._fresh:
            self._enqueue_refresh(touched)
        score_next = (not self.touched_on) or next_touched is not None
        agg, _ = self._enqueue_choose_train(ids, labels, B, plan, score_next, next_touched)
        self._enqueue_tail(ids, labels, B, agg, plan, True)
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

    def _enqueue_grad_slabs(self, ids, labels, B):
        """plan, scores, sort, select, gather, dense forward + backward partials: the gradient is left in the slabs (no Adam)."""
        plan = self._enqueue_plan_one(ids, labels, B, True)
        keys = self._enqueue_scores(True)
        agg, _ = self._enqueue_choose(ids, labels, B, keys, True, plan, sort_in_kernel=False)
        self._enqueue_tail(ids, labels, B, agg, plan, True, adam_clf=False)

    # ------------------------------------------------------------------
    def train_step(self, ids: torch.Tensor, labels: torch.Tensor, allreduce=None, defer: bool = False, plan: Optional[int] = None,
                   touched: Optional[int] = None, next_touched: Optional[int] = None):
        """zero_grad + loss + backward + Adam step for one batch (model_handler.py:149-153).
        ids / labels: int32 device tensors.  Nothing is returned and nothing syncs;
        ``last_loss()`` reads the batch loss afterwards.  ``allreduce(flat_grad)`` (data-parallel
        ranks) is called between the gradient reduction and the Adam update.  ``defer``: leave the Adam update of
        everything but the label classifier to the next step's front (or ``flush()``) - inside an epoch.
        ``plan``: address of this batch's plan slot if it has been planned already (a staged epoch); else it is planned here."""
        B = ids.numel()
        if B == 0:
            return
        if B > self.maxB:
            self.flush()
            self._alloc(B)
        self._lastB = B
        if allreduce is None:
            if plan is None:
                plan = self._enqueue_plan_one(ids, labels, B, True)
                touched = self._touch_one.data_ptr() if self.touched_on else None
            self._enqueue_step(ids, labels, B, plan, defer, touched, next_touched)
            return
        # data-parallel ranks: gradient of the local batch -> all-reduce -> the same Adam on every rank
        self.flush()
        self._enqueue_grad_slabs(ids, labels, B)
        self._enqueue_adam(B, apply=False, want_grad=True)
        allreduce(self.grad)
        self._enqueue_adam(B, apply=True, from_grad=True)
        self.clf_next.copy_(self.theta[self.n_rest:])      # (the label classifier is stepped with everything else here)

    def train_step_graph(self, ids: torch.Tensor, labels: torch.Tensor, timed: bool = False):
        """Same as train_step through captured hipGraphs (one set per batch size): one graph launch
        per step.  timed=True (bench.py, every Nth step) replays the step as [plan + scores graph] ->
        choose+aggregate bracketed by HIP events -> [dense+Adam graph]; same kernels, same order."""
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
        """Warm up (kernel attributes, per-batch-size plan slots: nothing may allocate inside a capture) on a side
        stream, capture every fn into a hipGraph of its own, and put the optimizer state back.  A deferred Adam update
        of real steps is applied first (the saved state then includes it); the warm-up's own is flushed and undone."""
        self.flush()
        state = (self.theta.clone(), self.m.clone(), self.v.clone(), self.step_counter.clone(), self.clf_next.clone())
        prof, self._prof = self._prof, None
        s = torch.cuda.Stream(self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for fn in (warm if warm is not None else fns):
                fn()
            self.flush()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        graphs = []
        # (no garbage collection while a stream is capturing: a collected engine of the caller's - tensors, events, other graphs -
        #  is torn down with HIP calls that are not allowed during a capture and end the process)
        import gc
        gc.collect()
        was_on = gc.isenabled()
        gc.disable()
        try:
            for fn in fns:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    fn()
                graphs.append(gr)
        finally:
            if was_on:
                gc.enable()
        for dst, src in zip((self.theta, self.m, self.v, self.step_counter, self.clf_next), state):
            dst.copy_(src)
        self._fresh = False                          # (s0 holds the warm-up's scores)
        self._prof = prof
        return graphs

    def _capture(self, B):
        ids, lab = self.ids_buf[:B], self.lab_buf[:B]
        keys = self.keys if self.g.n_pos else None
        g = self.g
        agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
        plan = self._plan_slot(B).data_ptr()

        def pre():                     # (a step on its own: nothing is known about the batch before or after it)
            self._enqueue_plan_one(ids, lab, B, True)
            self._enqueue_refresh(self._touch_one.data_ptr() if self.touched_on else None)

        def post():
            self._enqueue_tail(ids, lab, B, agg, plan, True)
            self.flush()

        def choose():
            self._enqueue_choose_train(ids, lab, B, plan, False)

        def full():
            pre()
            choose()
            post()

        names = ("full", "pre", "choose", "post")
        graphs = dict(zip(names, self._capture_graphs([full, pre, choose, post], warm=[full])))
        self._graphs[B] = graphs
        return graphs

    # -- whole-epoch path: ids / labels of an epoch live in static buffers, one plan slot and one graph per batch ------
    # There are TWO sets of epoch buffers (ids | labels | plan slots).  The picks of an epoch and its plans depend on nothing
    # a training step computes, so with ``epoch_run(prefetch=True)`` the sampler and the plan launches of epoch e + 1 run on a
    # parallel branch of epoch e's graph, into the other set - off the steps' serial chain altogether.
    @property
    def _ep_ids(self):
        return None if self._ep_sets is None else self._ep_sets[self._cur]["ids"]

    @property
    def _ep_lab(self):
        return self._ep_sets[self._cur]["lab"]

    @property
    def _ep_plans(self):
        return None if self._ep_sets is None else self._ep_sets[self._cur]["plans"]

    def stage_epoch(self, n: int, batch_size: int, n_epochs: int = 1):
        """Static id / label buffers and plan slots of n_epochs epochs of n picks each (the ids are filled by begin_epoch or by a
        sampler; ``plan_staged()`` must follow before any of the staged steps runs).  Several epochs staged together are sampled
        and planned by ONE launch each - the sampler's and the plan's latency chains are paid once per n_epochs epochs - and are
        walked as one sequence of batches 0 .. n_epochs * ceil(n / batch_size) - 1 (every epoch's last batch may be the shorter
        one).  Returns the CURRENT set's (ids, labels), n_epochs * n long."""
        if batch_size > self.maxB:
            self.flush()
            self._alloc(batch_size)
        nb_e = -(-n // batch_size)
        nb, total = nb_e * n_epochs, n * n_epochs
        stride = self._plan_bytes(batch_size)
        sets = self._ep_sets
        if sets is None or sets[0]["ids"].numel() < total or self._ep_stride != stride or sets[0]["plans"].numel() < nb * stride:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.PcgnnLibraryError("stage_epoch with a new shape inside a graph capture")
            self._ep_sets = [dict(ids=torch.zeros(total, dtype=torch.int32, device=self.dev),
                                  lab=torch.zeros(total, dtype=torch.int32, device=self.dev),
                                  plans=torch.zeros(nb * stride, dtype=torch.uint8, device=self.dev),
                                  touched=torch.zeros(nb * self._touch_stride, dtype=torch.uint8, device=self.dev)
                                  if self.touched_on else None) for _ in range(2)]
            self._ep_stride = stride
            self._cur, self._cur_ready = 0, False
            self._ep_graphs.clear()
        shape = (n, batch_size, n_epochs)
        if getattr(self, "_ep_shape", shape) != shape:
            # another epoch shape inside the same buffers: the last batch's (shorter) plan layout moves to another slot, and the
            # plan launch's look-back tags (a small integer per workgroup) would sit on top of whatever the old layout left
            # there - zero the slots so that no stale word can pass for a published total
            if torch.cuda.is_current_stream_capturing():
                raise _lib.PcgnnLibraryError("stage_epoch with a new epoch size inside a graph capture")
            for st in self._ep_sets:
                st["plans"].zero_()
            self._ep_graphs.clear()
        self._ep_shape = shape
        self._ep_n, self._ep_bs, self._ep_k = n, batch_size, n_epochs
        # batch j of the staged sequence: (first pick, size)
        self._ep_batches = [(e * n + b * batch_size, min(batch_size, n - b * batch_size)) for e in range(n_epochs) for b in range(nb_e)]
        return self._ep_ids[:total], self._ep_lab[:total]

    def take_prefetched(self) -> bool:
        """True if the current set already holds a sampled and planned epoch that no step has used yet (left by
        ``epoch_run(prefetch=True)``) - the caller then starts on it instead of sampling; the set counts as used from here on."""
        ready, self._cur_ready = self._cur_ready, False
        if ready and getattr(self, "_ev_ready", None) is not None:       # (prepared on the side stream: epoch_run(prefetch="stream"))
            torch.cuda.current_stream(self.dev).wait_event(self._ev_ready)
            self._ev_ready = None
        return ready

    def plan_staged(self, bump_counter: Optional[torch.Tensor] = None, which: Optional[int] = None):
        """Plan every batch of the staged epoch (set `which`, default the current one): one launch (after the sampler, before
        the epoch's first step).  bump_counter: the sampler's device epoch counter, incremented by it."""
        st = self._ep_sets[self._cur if which is None else which]
        self._enqueue_plan(st["ids"], st["lab"], self._ep_n, self._ep_bs, st["plans"], self._ep_stride, True, bump_counter, self._ep_k)
        if self.touched_on:
            nb_e = -(-self._ep_n // self._ep_bs)
            for e in range(self._ep_k):              # (an epoch's last batch may be the shorter one: the maps are made epoch by epoch)
                self._enqueue_mark(st["ids"][e * self._ep_n:], self._ep_n, self._ep_bs, st["touched"][e * nb_e * self._touch_stride:],
                                   st["plans"].data_ptr() + e * nb_e * self._ep_stride, self._ep_stride)
            self._fresh = False                      # (new maps: the rows scored so far need not cover the new first batch)

    def _ep_plan(self, b: int, which: Optional[int] = None) -> int:
        return self._ep_sets[self._cur if which is None else which]["plans"].data_ptr() + b * self._ep_stride

    def _ep_touch(self, b: int, which: Optional[int] = None) -> Optional[int]:
        if not self.touched_on:
            return None
        return self._ep_sets[self._cur if which is None else which]["touched"].data_ptr() + b * self._touch_stride

    def _ep_next_touch(self, b: int, which: Optional[int] = None) -> Optional[int]:
        """the map of the batch after b, if the staged sequence has one"""
        if not self.touched_on or b + 1 >= len(self._ep_batches):
            return None
        return self._ep_touch(b + 1, which)

    def begin_epoch(self, ids: torch.Tensor, labels: torch.Tensor, batch_size: int):
        """Stage an epoch's (already shuffled) ids and labels and plan its batches; afterwards ``epoch_step(b)`` is exactly
        one graph launch - no copies, no indexing kernels (model_handler.py:142-148 slices a Python list here)."""
        n = ids.numel()
        ep_ids, ep_lab = self.stage_epoch(n, batch_size)
        ep_ids.copy_(ids)
        ep_lab.copy_(labels)
        self._cur_ready = False
        self.plan_staged()

    def epoch_step(self, b: int, defer: bool = False):
        """Batch b of the staged (and planned) epoch as one graph replay.  defer: as inside epoch_run - the Adam update of
        everything but the label classifier is left to the next batch's front launch (the epoch's last batch flushes it), so
        the parameters are complete only after the last batch or a flush()."""
        if b >= len(self._ep_batches):
            return
        lo, B = self._ep_batches[b]
        self._lastB = B
        defer = defer and b + 1 < len(self._ep_batches)
        key = (self._cur, lo, B, self._ep_shape, "deferred") if defer else (self._cur, lo, B, self._ep_shape)
        gr = self._ep_graphs.get(key)
        if gr is None:
            ids, lab = self._ep_ids[lo:lo + B], self._ep_lab[lo:lo + B]
            gr = self._capture_graphs([lambda: self.train_step(ids, lab, defer=defer, plan=self._ep_plan(b), touched=self._ep_touch(b),
                                                               next_touched=self._ep_next_touch(b))])[0]
            self._ep_graphs[key] = gr
        if not self._fresh:                          # (the graph was captured with the scores at hand, or scores them itself)
            self._enqueue_refresh(self._ep_touch(b))
        gr.replay()
        self._fresh = (not self.touched_on) or self._ep_next_touch(b) is not None

    def epoch_step_timed(self, b: int, eager: bool = True, flush: bool = True):
        """Batch b of the staged epoch with HIP events around the select + gather call (appended to ``_prof``): the same
        kernels in the same order as one step of ``epoch_run``, reading the staged ids in place (no copies, no label
        gather).  eager: the four kernels launched one by one with the two event records between them (the host stays
        ahead of the GPU, so the events bracket the two kernels and nothing else); otherwise three graphs - scores | select +
        gather | dense - whose launch latency lands inside the bracket."""
        if b >= len(self._ep_batches):
            return
        lo, B = self._ep_batches[b]
        self._lastB = B
        key = (self._cur, lo, B, self._ep_shape, "timed")
        grs = self._ep_graphs.get(key)
        ids, lab = self._ep_ids[lo:lo + B], self._ep_lab[lo:lo + B]
        g = self.g
        plan = self._ep_plan(b)
        agg = self.agg.view(-1)[:g.R * B * g.feat_dim].view(g.R, B, g.feat_dim)
        keys = self.keys if g.n_pos else None
        # the staged sequence's last batch: nothing follows that would apply the deferred update (flush=False: the caller's next
        # launches - the next group's first gather launch, or its own flush - do)
        last = flush and b + 1 >= len(self._ep_batches)
        touched, nxt = self._ep_touch(b), self._ep_next_touch(b)
        score_next = (not self.touched_on) or nxt is not None
        if not self._fresh:
            self._enqueue_refresh(touched)
        if eager:
            self._enqueue_choose_train(ids, lab, B, plan, score_next, nxt)           # (records the two events itself)
            self._enqueue_tail(ids, lab, B, agg, plan, True)
            if last:
                self.flush()
            self.last_counts = self.cnt.view(-1)[:g.R * B].view(g.R, B)
            return
        if grs is None:
            parts = (lambda: self._enqueue_choose_train(ids, lab, B, plan, score_next, nxt),
                     lambda: (self._enqueue_tail(ids, lab, B, agg, plan, True), self.flush() if last else None))
            grs = self._capture_graphs(list(parts), warm=[lambda: (self._enqueue_refresh(touched), parts[0](), parts[1]())])
            self._ep_graphs[key] = grs
            self._enqueue_refresh(touched)
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
        grs[0].replay()
        ev[1].record()
        if self._prof is not None:
            self._prof.append(ev)
        self.last_counts = self.cnt.view(-1)[:g.R * B].view(g.R, B)
        grs[1].replay()
        self._fresh = score_next

    def epoch_run(self, n_steps: Optional[int] = None, sample=None, bump_counter: Optional[torch.Tensor] = None,
                  flush: bool = True, prefetch: bool = False, first_step: int = 0):
        """All batches of the staged epoch as ONE graph launch: the host latency between two graph launches (~8 us)
        is paid once per epoch instead of once per batch.  ``sample(ids, labels)``, if given, is enqueued (and captured): it
        fills the given id / label buffers on the device (pick + shuffle + labels), so a replay is a whole new epoch;
        the plans of all batches follow (which also bumps ``bump_counter``, the sampler's epoch number).
        flush=False: the last batch's deferred Adam update (everything but the label classifier) is left to whatever comes
        next - the next epoch's first front launch applies it, any other public call flushes it first.
        prefetch=True (needs ``sample``): the sampler and the plans of the NEXT epoch run on a parallel branch of this epoch's
        graph, into the other buffer set; afterwards that set is the current one and ready (``take_prefetched``).  The first
        such call - or one after the ready set was used up by other calls - samples its own epoch first.
        prefetch="stream": the same division of labour without a fork inside the graph: the sampler and plan launches of the next
        epoch are enqueued - by the host, behind this epoch's graph launch - on a second stream, where they run beside this
        epoch's kernels; the only cross-stream waits are two events per epoch (the other buffer set is free / is ready).
        first_step > 0: the REST of an epoch that is staged and planned already and whose first batches have been run some other
        way (batches first_step .. n_steps - 1): no sampler, no plans."""
        nb = len(self._ep_batches)
        n_steps = nb if n_steps is None else min(n_steps, nb)
        n = self._ep_n * self._ep_k
        batches = list(self._ep_batches)
        if first_step > 0:
            sample, prefetch = None, False
        on_stream = prefetch == "stream" and sample is not None
        prefetch = prefetch is True and sample is not None
        cur = self._cur
        primed = (prefetch or on_stream) and self._cur_ready
        key = ("epoch", cur, self._ep_shape, n_steps, sample is not None, None if bump_counter is None else bump_counter.data_ptr(),
               flush, prefetch, primed, on_stream, first_step)
        gr = self._ep_graphs.get(key)
        if gr is None:
            st = self._ep_sets[cur]
            nxt = self._ep_sets[cur ^ 1]
            def run():
                for b in range(first_step, n_steps):
                    lo, B = batches[b]
                    self.train_step(st["ids"][lo:lo + B], st["lab"][lo:lo + B], defer=True, plan=self._ep_plan(b, cur),
                                    touched=self._ep_touch(b, cur), next_touched=self._ep_next_touch(b, cur))
                if flush:
                    self.flush()
            def warm_run():                      # (the warm-up leaves the staged ids - and the epoch counter - as they are)
                if first_step == 0:
                    self.plan_staged(which=cur)
                elif not self._fresh:
                    self._enqueue_refresh(self._ep_touch(first_step, cur))
                run()
                if prefetch or on_stream:        # (the other set's plan slots and kernels get their first use outside a capture)
                    self.plan_staged(which=cur ^ 1)
            def sampled_run():
                if not primed and first_step == 0:
                    if sample is not None:
                        sample(st["ids"][:n], st["lab"][:n])
                    self.plan_staged(bump_counter, which=cur)
                if prefetch:                     # fork: the next epoch's sampler + plans beside this epoch's steps
                    main = torch.cuda.current_stream(self.dev)
                    side = torch.cuda.Stream(self.dev)
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        sample(nxt["ids"][:n], nxt["lab"][:n])
                        self.plan_staged(bump_counter, which=cur ^ 1)
                    run()
                    main.wait_stream(side)
                else:
                    run()
            gr = self._capture_graphs([sampled_run], warm=[warm_run])[0]
            self._ep_graphs[key] = gr
        self._lastB = batches[n_steps - 1][1]
        # (whole-table engine: the graph was captured with the scores at hand - it holds no score launch of its own; a
        #  touched-rows engine's graph scores its first batch's rows itself, behind its sampler and maps)
        if first_step > 0:
            if not self._fresh:                      # (the graph was captured with the scores at hand)
                self._enqueue_refresh(self._ep_touch(first_step, cur))
        elif not self.touched_on and not self._fresh:
            self._enqueue_refresh()
        main = torch.cuda.current_stream(self.dev)
        if primed and getattr(self, "_ev_ready", None) is not None:
            main.wait_event(self._ev_ready)          # the side stream has sampled and planned this set
            self._ev_ready = None
        gr.replay()
        self._fresh = (not self.touched_on) or self._ep_next_touch(n_steps - 1, cur) is not None
        if on_stream:
            # the next epoch's sampler + plans, on the side stream: behind the graph launched BEFORE this one (the last reader of
            # the other set), beside this one
            if getattr(self, "_side", None) is None:
                self._side, self._ev_free = torch.cuda.Stream(self.dev), None
            nxt = self._ep_sets[cur ^ 1]
            if self._ev_free is not None:
                self._side.wait_event(self._ev_free)
            with torch.cuda.stream(self._side):
                sample(nxt["ids"][:n], nxt["lab"][:n])
                self.plan_staged(bump_counter, which=cur ^ 1)
                self._ev_ready = torch.cuda.Event()
                self._ev_ready.record(self._side)
            self._ev_free = torch.cuda.Event()
            self._ev_free.record(main)               # (this epoch's graph: the last reader of the set it ran on)
            self._cur, self._cur_ready = cur ^ 1, True
        elif prefetch:
            self._cur, self._cur_ready = cur ^ 1, True
        else:
            self._cur_ready = False
        return n_steps

    def read_batch_lists(self, b: int):
        """The selection lists batch b of the staged epoch has in the workspace RIGHT NOW - what the last select launch on that
        batch's plan slot wrote (the data part is shared by all batches: call it before another batch is selected) - copied to
        the host: sets[r][i] for relation r, centre i.  Synchronises.  (bench.py's `verified`, tests.)"""
        import numpy as np
        g, lib = self.g, self.lib
        B = self._ep_batches[b][1]
        rows = g.R * B
        torch.cuda.synchronize(self.dev)
        off = lambda which: int(lib.pcg_choose_workspace_offset(g.desc_ref(), B, self.list_capacity, which))
        slot = self._ep_plans[b * self._ep_stride:(b + 1) * self._ep_stride]
        begin = slot[off(0):off(0) + 8 * (rows + 1)].view(torch.int64).cpu().numpy()
        length = slot[off(1):off(1) + 4 * rows].view(torch.int32).cpu().numpy()
        d0 = off(2) - self._plan_bytes(B)
        lst = self.data[d0:d0 + 4 * max(int(begin[-1]), 1)].view(torch.int32).cpu().numpy()
        sets = []
        for r in range(g.R):
            row_sets = []
            for i in range(B):
                row = r * B + i
                seg = lst[begin[row]:begin[row] + length[row]]
                row_sets.append(set(seg[seg >= 0].tolist()))
            sets.append(row_sets)
        return sets

    def check(self):
        """Raise if any batch since the last check did not fit its selection list (the kernels then select nothing and
        only set the device status word), or a list named a row outside its table, or an in-kernel wait ran out.  Reads one
        word: synchronises - call it where the host waits anyway (``last_loss``, the end of an evaluation pass or of an epoch,
        after a timed region)."""
        st = int(self.status.item())
        if st == 0:
            return
        self.status.zero_()
        what = []
        if st & _lib.PCG_ST_SEL_OVERFLOW:
            what.append("selection list overflow: a batch needed more list entries than the workspace holds - raise "
                        "FusedPCGNN(list_capacity=...)")
        if st & _lib.PCG_ST_LIST_ID_RANGE:
            what.append("a selection list named a row outside the feature table")
        if st & _lib.PCG_ST_SYNC_TIMEOUT:
            what.append("a bounded in-kernel wait ran out: the select kernel's wait for its own train-pos sort, or a plan "
                        "workgroup's wait for its predecessors' totals")
        if st & _lib.PCG_ST_SORT_OVERFLOW:
            what.append("the one-launch sort of the train positives met a bucket of more than 4096 keys: minority picks may be wrong")
        raise _lib.PcgnnLibraryError("; ".join(what) or f"device status {st}")

    def last_loss(self) -> torch.Tensor:
        self.check()
        B = self._lastB
        return self.row_loss[:B].sum() / (B * self.scale)

    def gradients(self, ids: torch.Tensor, labels: torch.Tensor, via: str = "acts") -> Dict[str, torch.Tensor]:
        """loss.backward() without the optimizer step: per-parameter gradients (parity tests).  via="acts": the training
        step's kernels (dense kernel -> transposed activations -> the weight-gradient GEMMs of pcg_wgrad, the label
        classifier's tiles included); via="slabs": per-tile gradient slabs summed in tile order (the data-parallel paths')."""
        B = ids.numel()
        self.flush()
        if B > self.maxB:
            self._alloc(B)
        if via == "slabs":
            self._enqueue_grad_slabs(ids, labels, B)
            self.step_counter -= 1               # the dense kernel counted a step that is not taken
            self._enqueue_adam(B, apply=False, want_grad=True)
        else:
            plan = self._enqueue_plan_one(ids, labels, B, True)
            keys = self._enqueue_scores(True)
            agg, _ = self._enqueue_choose(ids, labels, B, keys, True, plan, sort_in_kernel=False)
            self._enqueue_tail(ids, labels, B, agg, plan, True, adam_clf=4)
            self.step_counter -= 1
            b1, b2 = self.betas
            _lib.check(self.lib.pcg_wgrad(_p(self.acts), self.act_ld, B, self.F, self.E, self.R, None, None, None, None, self.lr,
                                          b1, b2, self.eps, self.wd, _p(self.grad), 0, 1, _p(self.wg_scratch), None, self._stream()), "pcg_wgrad")
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
        plan = self._enqueue_plan_one(ids, labels, B, train_flag)
        keys = self._enqueue_scores(train_flag)
        agg, _ = self._enqueue_choose(ids, labels, B, keys, train_flag, plan, sort_in_kernel=False)
        comb = torch.empty(B, self.E, dtype=torch.float32, device=self.dev) if want_combined else None
        self._enqueue_tail(ids, None, B, agg, plan, False, combined=comb)
        res = (self.logits[:B].clone(), self.center[:B].clone())
        return res + (comb,) if want_combined else res
