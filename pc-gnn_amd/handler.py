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
                                    global_batch_scale=cfg.get("world_size", 1))
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

    def start_epoch_staged(self) -> torch.Tensor:
        """pick + shuffle + label lookup in ONE launch (pcg_pick_shuffled), straight into the fused engine's epoch
        buffers; the epoch number lives on the device and is incremented by the call.  Returns the staged ids."""
        ids, lab = self.fused.stage_epoch(self.pick_size, self.batch_size)
        self.sampler.pick_shuffled(self.pick_size, ids, self.labels_i32, lab, epoch_counter=self._epoch_dev, bump=True)
        return ids

    def run_epoch_one_graph(self) -> int:
        """A whole epoch - pick, shuffle, labels and every batch's training step - as one graph launch."""
        self.fused.stage_epoch(self.pick_size, self.batch_size)
        self.fused.epoch_run(sample=self.start_epoch_staged)
        return self.pick_size

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
