"""Training harness: the counterpart of ``ModelHandler.train`` (src/model_handler.py:77-178)
for the HIP path.  Same config keys, same loop structure:

    for epoch:  pick 2*|train_pos| nodes (:130) -> shuffle (:133) -> batches of batch_size (:134-148)
                per batch: zero_grad, loss, backward, Adam step (:149-153)   <- the reference's timed window

but ids / labels stay on the device for the whole epoch (one pick kernel, one
permutation, batch = a slice), and the empty trailing batch the reference's
``int(len/B)+1`` produces (and crashes on) is not run.
"""
import time
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn

from .graph import DeviceGraph
from .layers import InterAgg, IntraAgg
from .model import PCALayer
from .sampler import PickSampler

DEFAULTS = dict(model="PCGNN", emb_size=64, rho=0.5, alpha=2.0, lr=0.01, weight_decay=0.001, batch_size=1024,
                epochs=1, seed=0, engine="graph")   # engine: "graph" (fused kernels in a hipGraph) | "fused" | "torch"


class PCGNNTrainer:
    def __init__(self, workload, config: Optional[Dict] = None, device=None):
        cfg = dict(DEFAULTS)
        cfg.update(config or {})
        self.cfg = cfg
        self.w = workload
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        torch.manual_seed(cfg["seed"])
        w = workload
        self.graph = DeviceGraph(w.X, w.csr, w.train_pos, self.device)
        feats = nn.Embedding(w.n, w.X.shape[1])
        feats.weight = nn.Parameter(torch.from_numpy(w.X), requires_grad=False)      # model_handler.py:85-86
        f = w.X.shape[1]
        intras = [IntraAgg(feats, f, cfg["emb_size"], w.train_pos, cfg["rho"], cuda=True) for _ in w.csr]
        inter = InterAgg(feats, f, cfg["emb_size"], w.train_pos, self.graph, intras, cuda=True)   # :103-113
        self.model = PCALayer(2, inter, cfg["alpha"]).to(self.device)                 # :114,:122
        self.engine = cfg["engine"]
        self.fused = None
        self.opt = None
        if self.engine == "torch":      # dense tail + Adam through torch autograd (generic, slower)
            self.opt = torch.optim.Adam([p for p in self.model.parameters() if p.requires_grad], lr=cfg["lr"],
                                        weight_decay=cfg["weight_decay"])             # :124
        else:
            from .fused import FusedPCGNN
            self.fused = FusedPCGNN(self.model, cfg["lr"], cfg["weight_decay"], max_batch=cfg["batch_size"],
                                    global_batch_scale=cfg.get("world_size", 1), list_capacity=cfg.get("list_capacity"))
        self.labels_dev = torch.from_numpy(w.labels).to(self.device)
        self.labels_i32 = self.labels_dev.to(torch.int32)
        self.sampler = PickSampler(w.idx_train, w.labels[w.idx_train], w.homo_deg[w.idx_train], self.device,
                                   seed=cfg["seed"])
        self.pick_size = 2 * len(w.train_pos)                                          # :130
        self.batch_size = cfg["batch_size"]
        self._epoch_dev = torch.zeros(2, dtype=torch.int64, device=self.device)      # [epoch of the staged sampler, kernel scratch]
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(cfg["seed"])

    # ------------------------------------------------------------------
    def batches_per_epoch(self) -> int:
        return (self.pick_size + self.batch_size - 1) // self.batch_size

    def start_epoch(self, epoch: int) -> torch.Tensor:
        """pick + shuffle (model_handler.py:130-133) -> int32 ids on the device."""
        picked = self.sampler.pick(self.pick_size, epoch)
        perm = torch.randperm(self.pick_size, device=self.device, generator=self._gen)
        return picked[perm]

    def start_epoch_staged(self, n_epochs: int = 1) -> torch.Tensor:
        """pick + shuffle + label lookup in ONE launch (pcg_pick_shuffled_epochs), straight into the fused engine's epoch
        buffers, then every batch's plan; the epoch number lives on the device and is moved on by the plan launch.  If a
        previous ``run_epoch_one_graph(prefetch=True)`` has prepared this epoch already, nothing is launched.
        n_epochs > 1: that many epochs are sampled and planned together (one sampler launch, one plan launch) and walked as
        one sequence of n_epochs * batches_per_epoch() batches.  Returns the staged ids (n_epochs * pick_size)."""
        ids, lab = self.fused.stage_epoch(self.pick_size, self.batch_size, n_epochs)
        if not self.fused.take_prefetched():
            self._sample_into(ids, lab)
            self.fused.plan_staged(self._epoch_dev)      # every batch's plan; it also moves the epoch number on
        return ids

    def _sample_into(self, ids: torch.Tensor, lab: torch.Tensor):
        """fill ids / lab (a whole number of epochs of pick_size draws) with consecutive epochs' shuffled picks"""
        self.sampler.pick_shuffled(self.pick_size, ids, self.labels_i32, lab, epoch_counter=self._epoch_dev, bump=False,
                                   n_epochs=ids.numel() // self.pick_size)

    def nodes_of_steps(self, n_steps: int) -> int:
        """sampled nodes of the first n_steps batches of a staged sequence of epochs (every epoch's last batch is the short one)"""
        nb = self.batches_per_epoch()
        full, rest = divmod(n_steps, nb)
        return full * self.pick_size + min(self.pick_size, rest * self.batch_size)

    def run_epoch_one_graph(self, flush: bool = True, prefetch: bool = False, n_steps: Optional[int] = None, n_epochs: int = 1) -> int:
        """A whole epoch - pick, shuffle, labels, every batch's plan and every batch's training step - as one graph launch
        (n_epochs > 1: that many epochs, sampled and planned together, in one launch).
        flush=False / prefetch=True: see FusedPCGNN.epoch_run (back-to-back epochs: the next epoch's first launch applies the
        last update; the next epoch's sampler and plans run beside this epoch's steps)."""
        self.fused.stage_epoch(self.pick_size, self.batch_size, n_epochs)
        done = self.fused.epoch_run(n_steps=n_steps, sample=self._sample_into, bump_counter=self._epoch_dev, flush=flush, prefetch=prefetch)
        return self.nodes_of_steps(done)

    def step(self, batch_ids: torch.Tensor, timed: bool = False) -> torch.Tensor:
        """One iteration of the batch loop (model_handler.py:147-153)."""
        if self.fused is not None:
            labels = self.labels_i32[batch_ids.long()]
            if self.engine == "graph":
                self.fused.train_step_graph(batch_ids, labels, timed)
            else:
                self.fused.train_step(batch_ids, labels)
            return None
        labels = self.labels_dev[batch_ids.long()]
        self.opt.zero_grad(set_to_none=True)
        loss = self.model.loss(batch_ids, labels)
        loss.backward()
        self.opt.step()
        return loss

    def run_epoch_graph(self, epoch: int) -> int:
        """One epoch through the per-slot graphs: pick + shuffle + label gather once, then one graph
        launch per batch.  Returns the number of sampled nodes."""
        ids = self.start_epoch(epoch)
        self.fused.begin_epoch(ids, self.labels_i32[ids.long()], self.batch_size)
        for b in range(self.batches_per_epoch()):
            self.fused.epoch_step(b)
        return self.pick_size

    def train_epoch(self, epoch: int):
        """Returns (#sampled nodes, seconds inside the reference's per-batch window,
        seconds including pick + shuffle)."""
        torch.cuda.synchronize(self.device)
        t_all = time.perf_counter()
        ids = self.start_epoch(epoch)
        torch.cuda.synchronize(self.device)
        t0 = time.perf_counter()
        for b in range(self.batches_per_epoch()):
            self.step(ids[b * self.batch_size:(b + 1) * self.batch_size])
        torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        return self.pick_size, t1 - t0, t1 - t_all


# ---------------------------------------------------------------------------------------------------------------------
# ModelHandler: the reference's experiment driver (src/model_handler.py:24-178) over the HIP path
# ---------------------------------------------------------------------------------------------------------------------
EPOCHS_PER_LAUNCH = 4        # epochs ModelHandler.train samples, plans and replays per launch (FusedPCGNN.stage_epoch(n_epochs))
FIRST_LABELED = {"amazon": 3305, "amazon_new": 2013}      # model_handler.py:38-40: the unlabeled Amazon users come first


def _csr_of(adj, n):
    """one relation in any of the accepted forms -> (indptr, indices): a (indptr, indices) pair, or the reference's
    dict[int -> set[int]] (src/utils.py:226-239)"""
    if isinstance(adj, (tuple, list)) and len(adj) == 2 and hasattr(adj[0], "shape"):
        return np.asarray(adj[0], dtype=np.int64), np.asarray(adj[1], dtype=np.int32)
    from .graph import adj_to_csr
    return adj_to_csr(adj, n)


class ModelHandler(object):
    """``ModelHandler(config).train() -> (auc_test, recall_test, f1_macro_test)`` with the reference's config keys
    (generate_exp_config.ipynb:50-66): data_name, model (PCGNN | SAGE | GCN), seed, train_ratio, test_ratio, emb_size, rho,
    alpha, lr, weight_decay, batch_size, epochs, valid_epochs, patience (+ exp_num, kept for bookkeeping).

    The reference reads its datasets from files that exist nowhere offline (``load_data``, src/utils.py:66-207), so the
    graph is handed in: ``dataset = (homo, relation_list, feat_data, labels)`` - exactly what ``load_data`` returns
    (dict-of-sets adjacencies, or (indptr, indices) pairs) - or a ``synth.Workload``.  Everything after that follows the
    reference: seeded stratified train / valid / test split (:36-48, ``utils.split_dataset``), ``pos_neg_split`` (:56),
    Amazon feature normalisation (:59-60), model construction (:85-122), Adam (:124), the epoch loop with the pick sampler
    (:128-156), validation every ``valid_epochs`` with the gain rule and the best checkpoint (:158-169), patience (:170-173),
    restore + final test (:175-178), ``ResultManager`` logs.  One difference: pick and shuffle draw from the device
    generator (Philox), not from Python's ``random`` - the same distribution, not the same draws."""

    def __init__(self, config, dataset=None, device=None):
        import argparse
        import random
        from .result_manager import EXP_RES_DIR, ResultManager
        from . import utils as U
        self.result = ResultManager(args=config, root=config.get("result_dir", EXP_RES_DIR))
        args = argparse.Namespace(**config)
        if dataset is None:
            raise FileNotFoundError(
                f"dataset '{args.data_name}': the reference loads ./data/pyg/... files (src/utils.py:66-207) that are not part of "
                f"either tree; pass dataset=(homo, relation_list, feat_data, labels) or a synth.Workload")
        if hasattr(dataset, "csr"):                 # synth.Workload
            w = dataset
            homo_deg = np.asarray(w.homo_deg, dtype=np.int64)
            relation_list, feat_data, labels = list(w.csr), np.asarray(w.X), np.asarray(w.labels)
            homo = None
        else:
            homo, relation_list, feat_data, labels = dataset
            feat_data, labels = np.asarray(feat_data), np.asarray(labels)
            homo_deg = None
        n = feat_data.shape[0]
        np.random.seed(args.seed)                                                                    # :35-36
        random.seed(args.seed)
        first = FIRST_LABELED.get(args.data_name, 3305 if str(args.data_name).startswith("amazon") else 0)
        idx_train, y_train, idx_valid, y_valid, idx_test, y_test = U.split_dataset(labels, args.train_ratio, args.test_ratio,
                                                                                    args.seed, first)   # :36-48
        print(f"Run on {args.data_name}, postive/total num: {np.sum(labels)}/{len(labels)}, train num {len(y_train)},"
              f"valid num {len(y_valid)}, test num {len(y_test)}, test positive num {np.sum(y_test)}")       # :50-51
        print(f"Feature dimension: {feat_data.shape[1]}")
        train_pos, train_neg = U.pos_neg_split(idx_train, y_train)                                    # :56
        if str(args.data_name).startswith("amazon"):                                                   # :59-60
            feat_data = np.asarray(U.normalize(feat_data), dtype=np.float32)
        self.relations = [_csr_of(a, n) for a in relation_list]
        if homo is not None:
            self.homo = _csr_of(homo, n)
            homo_deg = np.diff(self.homo[0])
        else:                                       # union of the relations (what the reference's homo pickle holds)
            keys = np.unique(np.concatenate([np.repeat(np.arange(n, dtype=np.int64), np.diff(ip)) * n + ix for ip, ix in self.relations]))
            rows = keys // n
            hp = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(np.bincount(rows, minlength=n), out=hp[1:])
            self.homo = (hp, (keys - rows * n).astype(np.int32))
        print(f"Model: {args.model}, emb_size: {args.emb_size}.")                                       # :68
        self.args = args
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dataset = {"feat_data": feat_data.astype(np.float32), "labels": labels, "homo_deg": homo_deg,
                        "idx_train": idx_train, "idx_valid": idx_valid, "idx_test": idx_test,
                        "y_train": y_train, "y_valid": y_valid, "y_test": y_test,
                        "train_pos": train_pos, "train_neg": train_neg}

    def _build(self):
        """model construction (:85-122)"""
        from . import graphsage as GS
        from .fused import FusedPCGNN
        from .layers import InterAgg1, InterAgg3, InterAgg5
        args, ds, dev = self.args, self.dataset, self.device
        feat_data = ds["feat_data"]
        n, f = feat_data.shape
        features = nn.Embedding(n, f)
        features.weight = nn.Parameter(torch.from_numpy(feat_data), requires_grad=False)                # :85-86
        if args.model == "PCGNN":
            graph = DeviceGraph(feat_data, self.relations, ds["train_pos"], dev)
            intras = [IntraAgg(features, f, args.emb_size, ds["train_pos"], args.rho, cuda=True) for _ in self.relations]
            cls = {1: InterAgg1, 3: InterAgg3, 5: InterAgg5}.get(len(intras), InterAgg)                 # :103-113
            inter = cls(features, f, args.emb_size, ds["train_pos"], graph, intras, cuda=True)
            model = PCALayer(2, inter, args.alpha).to(dev)                                              # :114, :122
            engine = FusedPCGNN(model, args.lr, args.weight_decay, max_batch=args.batch_size)          # :124 (Adam, fused)
            return model, engine, None
        graph = DeviceGraph(feat_data, [self.homo], [], dev)
        if args.model == "SAGE":                                                                        # :96-98, :115-118
            enc = GS.Encoder(features, f, args.emb_size, graph, GS.MeanAggregator(features, cuda=True), gcn=True, cuda=True)
            model = GS.GraphSage(2, enc)
        elif args.model == "GCN":                                                                       # :99-101, :119-120
            enc = GS.GCNEncoder(features, f, args.emb_size, graph, GS.GCNAggregator(features, cuda=True), cuda=True)
            model = GS.GCN(2, enc)
        else:
            raise ValueError(f"model {args.model!r}: PCGNN, SAGE or GCN")
        model = model.to(dev)
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=args.lr, weight_decay=args.weight_decay)
        return model, None, opt

    def train(self):
        import random
        from . import utils as U
        args, ds, dev = self.args, self.dataset, self.device
        idx_train, y_train = ds["idx_train"], ds["y_train"]
        model, engine, opt = self._build()
        labels_dev = torch.from_numpy(np.asarray(ds["labels"]).astype(np.int32)).to(dev)
        auc_best, f1_mac_best, epoch_best = 1e-10, 1e-10, 0                                              # :125
        saved = False
        if engine is not None:
            sampler = PickSampler(idx_train, y_train, ds["homo_deg"][np.asarray(idx_train)], dev, seed=args.seed)
            pick_size = 2 * len(ds["train_pos"])                                                        # :130
            epoch_dev = torch.zeros(2, dtype=torch.int64, device=dev)
        self.epoch_time = []
        epoch = -1
        group_left = 0                                   # epochs of the current group (one launch) still to be accounted for
        for epoch in range(args.epochs):                                                                 # :128
            t0 = time.perf_counter()
            if engine is not None and group_left == 0:
                # pick + shuffle + label lookup on the device, then every batch (an epoch's last one partial; the empty batch
                # the reference's int(len / B) + 1 can produce is not run): one sampler launch, one plan launch and one graph
                # launch for a GROUP of up to EPOCHS_PER_LAUNCH epochs - never across a validation point, the end of training
                # or the first epoch at which the patience rule (:170-173) can stop the run, so the loop below behaves epoch by
                # epoch exactly as the reference's
                k = min(EPOCHS_PER_LAUNCH, args.valid_epochs - epoch % args.valid_epochs, args.epochs - epoch,
                        max(1, epoch_best + args.patience + 2 - epoch))
                engine.stage_epoch(pick_size, args.batch_size, k)
                engine.epoch_run(sample=lambda ids, lab: sampler.pick_shuffled(pick_size, ids, labels_dev, lab, epoch_counter=epoch_dev,
                                                                                bump=False, n_epochs=ids.numel() // pick_size),
                                 bump_counter=epoch_dev)
                group_left = k
            if engine is not None:
                group_left -= 1
            else:
                sampled = list(idx_train)                                                                # :132-133
                random.shuffle(sampled)
                for b0 in range(0, len(sampled), args.batch_size):
                    batch_nodes = sampled[b0:b0 + args.batch_size]
                    opt.zero_grad()
                    loss = model.loss(batch_nodes, labels_dev[torch.as_tensor(batch_nodes, device=dev)].long())
                    loss.backward()
                    opt.step()
            self.epoch_time.append(time.perf_counter() - t0)                                              # (:155; enqueue time only)
            if (epoch + 1) % args.valid_epochs == 0:                                                      # :158-169
                print("Valid at epoch {}".format(epoch))
                auc_val, recall_val, f1_mac_val, precision_val = U.test(ds["idx_valid"], ds["y_valid"], engine or model,
                                                                        args.batch_size, self.result, epoch, epoch_best, flag="val")
                gain_auc = (auc_val - auc_best) / auc_best
                gain_f1_mac = (f1_mac_val - f1_mac_best) / f1_mac_best
                if (gain_auc + gain_f1_mac) > 0:
                    f1_mac_best, auc_best, epoch_best = f1_mac_val, auc_val, epoch
                    if engine is not None:
                        engine.flush()
                    torch.save(model.state_dict(), self.result.model_path)
                    saved = True
            if (epoch - epoch_best) > args.patience:                                                      # :170-173
                print(f"Early stopping at epoch {epoch}")
                break
        if engine is not None:
            engine.check()
        print("Restore model from epoch {}".format(epoch_best))                                          # :175-177
        if saved:
            if engine is not None:
                engine.flush()
            model.load_state_dict(torch.load(self.result.model_path, weights_only=True))
            if engine is not None:
                engine.params_changed()          # (the engine's own copy of the label classifier follows the restored one)
        auc_test, recall_test, f1_mac_test, precision_test = U.test(ds["idx_test"], ds["y_test"], engine or model, args.batch_size,
                                                                    self.result, epoch_best=epoch_best, flag="test")
        self.model, self.engine, self.epoch_best, self.last_epoch = model, engine, epoch_best, epoch
        return auc_test, recall_test, f1_mac_test
