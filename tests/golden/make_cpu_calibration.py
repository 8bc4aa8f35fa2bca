#!/usr/bin/env python3
"""Calibrate bench.py's CPU baseline: how much faster (or slower) is the oracle port than THE REFERENCE ITSELF?

Runs only in the build container, where the reference tree is mounted read-only at /root/reference:

    python tests/golden/make_cpu_calibration.py

bench.py times ``oracle/pcgnn_oracle.py`` (``cpu_baseline.kind = "port"``) on the GPU box, where the reference cannot
travel.  This script times the imported reference (``src/layers.py`` + ``src/model.py``, ``cuda=False``) and the oracle
on the same graphs (BASELINE configs[0]/[1] shapes), the same batches, the same parameters and the same thread count,
and stores the ratio in ``cpu_calibration.json`` (data only).  bench.py then reports
``cpu_baseline.port_over_reference`` and ``cpu_baseline.reference_equivalent = value / ratio``.
"""
import json
import os
import random
import sys
import time

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import src.layers as RL  # noqa: E402
from src.model import PCALayer  # noqa: E402

from oracle import pcgnn_oracle as O  # noqa: E402
from pcgnn_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def calibrate(name, w, batch, lr, wd, n_batches, threads):
    torch.set_num_threads(threads)
    n, f = w.X.shape
    adj = [{v: set(idx[ip[v]:ip[v + 1]].tolist()) for v in range(n)} for ip, idx in w.csr]
    rng = random.Random(0)
    weights = np.diff(np.concatenate([[0.0], synth.pick_cum_weights(w)]))
    picked = rng.choices(w.idx_train.tolist(), weights=weights.tolist(), k=n_batches * batch)
    batches = [picked[i * batch:(i + 1) * batch] for i in range(n_batches)]
    # the reference model (src/model_handler.py:85-114,124 restated with cuda=False)
    torch.manual_seed(0)
    features = torch.nn.Embedding(n, f)
    features.weight = torch.nn.Parameter(torch.FloatTensor(w.X), requires_grad=False)
    intras = [RL.IntraAgg(features, f, 64, w.train_pos, 0.5, cuda=False) for _ in adj]
    inter = RL.InterAgg3(features, f, 64, w.train_pos, adj, intras, cuda=False)
    ref = PCALayer(2, inter, 2.0)
    params = {k: v.detach().clone() for k, v in ref.state_dict().items() if "features" not in k}
    opt = torch.optim.Adam(filter(lambda p: p.requires_grad, ref.parameters()), lr=lr, weight_decay=wd)
    om = O.OraclePCGNN(torch.from_numpy(w.X), adj, w.train_pos, params, 0.5, 2.0, dense_mask=True)
    oopt = O.make_adam(om, lr, wd)
    t_ref = t_or = 0.0
    losses = []
    for b in batches:
        lab = w.labels[np.array(b)]
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = ref.loss(b, torch.LongTensor(lab))
        loss.backward()
        opt.step()
        t_ref += time.perf_counter() - t0
        t0 = time.perf_counter()
        lo = O.train_step(om, oopt, b, lab)
        t_or += time.perf_counter() - t0
        losses.append((float(loss.item()), lo))
    nodes = n_batches * batch
    out = {"workload": f"{w.name} N={n} F={f} batch={batch}", "batches": n_batches, "threads": threads,
           "reference_nodes_per_s": nodes / t_ref, "oracle_nodes_per_s": nodes / t_or,
           "port_over_reference": (nodes / t_or) / (nodes / t_ref),
           "max_loss_difference": max(abs(a - b) for a, b in losses)}
    print(name, json.dumps(out))
    return out


if __name__ == "__main__":
    threads = min(os.cpu_count() or 1, 16)
    res = {"torch": torch.__version__, "host_cores": os.cpu_count(),
           "note": "reference = /root/reference src/layers.py + src/model.py with cuda=False driven as src/model_handler.py:149-153 "
                   "does; oracle = oracle/pcgnn_oracle.py train_step (dense-mask formulation); same graphs, batches, parameters, threads"}
    res["yelp"] = calibrate("yelp", synth.yelp_like(0), 1024, 0.01, 0.001, 3, threads)
    res["amazon"] = calibrate("amazon", synth.amazon_like(0), 256, 0.005, 0.0005, 4, threads)
    with open(os.path.join(HERE, "cpu_calibration.json"), "w") as f:
        json.dump(res, f, indent=1)
