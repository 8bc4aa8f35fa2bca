#!/usr/bin/env python3
"""bench.py - sampled-nodes/sec of the PC-GNN train step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload yelp|amazon|powerlaw]

A "step" is one mini-batch through the whole hot path: label-aware scoring,
train-pos sort, choose + aggregate, dense tail, loss, backward, Adam - the
reference's timed window (src/model_handler.py:143-155) - plus, at every epoch
start, the pick + shuffle (:130-133).  Steps walk real epochs: 2*|train_pos|
picked nodes -> ceil(./B) batches, the last one partial; `value` counts the nodes
actually processed.  Inputs (graph, features, labels) are resident in HBM before
the timed region.

N=1 workload: BASELINE.json configs[1] - YelpChi-shaped synthetic graph
(N=45,954, F=32, 3 relations with 49,315 / 573,616 / 3,402,743 undirected edges,
14.53 % positives, 40 % train), emb 64, batch 1024, rho 0.5.

Graph engine (default at N=1): a GROUP of `--epochs-per-launch` epochs - one sampler launch (pick +
shuffle + labels of all of them), one plan launch (every batch's plan), every batch's three launches
(select_rows | gather_train_kernel | dense_step) - is ONE hipGraph launch.  The second group of the
timed region and every `--event-every`-th after it (the only group, if the region holds but one) is
event-bracketed: sampler and plans as launches of their own, the group's first batch kernel by kernel
with HIP events around the pcg_choose_gather_train call (same kernels, same order), the rest of the
group one graph launch; `--post-brackets` further epochs after the clock has stopped add one such
bracket each.  Those events give the `roofline` object (dominant call = select_rows +
gather_train_kernel: selection, gather, the next step's score pass, the deferred Adam update; bytes:
algorithmic_bytes()).  N>1: the destination-node partitioned path (pc-gnn_amd/dist.py), weak scaling.

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline` (the oracle port timed
on the host cores on the event-bracketed batches of the same run; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=24)
    ap.add_argument("--workload", default="yelp", choices=["yelp", "amazon", "powerlaw"])
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--relabel-by-degree", action="store_true", help="power-law workload: renumber the nodes by descending degree "
                    "(diagnostic for the select kernel's score-gather over-fetch: DESIGN.md section 9)")
    ap.add_argument("--emb", type=int, default=64)
    ap.add_argument("--rho", type=float, default=0.5)
    ap.add_argument("--nodes", type=int, default=2_000_000, help="powerlaw only")
    ap.add_argument("--edges", type=int, default=40_000_000, help="powerlaw only")
    ap.add_argument("--cpu-batches", type=int, default=24, help="oracle batches timed for cpu_baseline (0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-epoch-graphs", action="store_true", help="graph engine: copy each batch into a static buffer "
                    "instead of staging the whole epoch (A/B switch)")
    ap.add_argument("--force-partitioned", action="store_true",
                    help="run the node-partitioned RCCL path even at world size 1 (rehearsal of the N>1 code)")
    ap.add_argument("--window", type=int, default=8, help="partitioned path: steps per halo prefetch (feature rows never change, "
                    "so the remote rows of a window's centres are fetched once)")
    ap.add_argument("--epochs-per-launch", type=int, default=0, help="(0 = four, or as many as make a group of >= 24 steps: epochs of two or three batches) graph engine: epochs sampled, planned and replayed together (one "
                    "sampler launch, one plan launch, one graph launch per group; 1 = epoch by epoch)")
    ap.add_argument("--event-every", type=int, default=3,
                    help="graph engine: bracket the select+aggregate launch of the first batch of every Nth group of epochs with HIP events")
    ap.add_argument("--list-capacity", type=int, default=None, help="entries of the selection list (default: the graph's worst case); "
                    "a value too small for a batch makes the run exit non-zero (the status word is checked after the timed region)")
    ap.add_argument("--report-epochs", type=int, default=7, help="graph engine: epochs timed one by one AFTER the timed region "
                    "(sampler and batches as two graph launches with events in between) for the reference-window / pick-inclusive "
                    "medians (SURVEY 8d); 0 = skip")
    ap.add_argument("--post-brackets", type=int, default=8, help="graph engine: epochs run AFTER the timed region whose first batch is "
                    "launched kernel by kernel with HIP events around pcg_choose_gather_train: further samples for `roofline` "
                    "(the timed region itself brackets one batch every --event-every groups of epochs); 0 = none")
    ap.add_argument("--verify-batches", type=int, default=1, help="batches whose chosen sets are checked after the clock stops "
                    "(count law on every row, the oracle's sets on a strided sample); 0 = skip")
    ap.add_argument("--engine", default=None, choices=["graph", "fused", "torch", "dp"],
                    help="graph: fused HIP step replayed from a hipGraph (default at 1 GPU); fused: same kernels "
                         "launched eagerly (default at N>1, gradient all-reduce in between); torch: torch dense tail")
    return ap.parse_args()


def make_workload(args):
    from pcgnn_amd import synth
    if args.workload == "yelp":
        return synth.yelp_like(args.seed), 1024, 0.01, 0.001
    if args.workload == "amazon":
        return synth.amazon_like(args.seed), 256, 0.005, 0.0005
    w = synth.power_law(args.nodes, args.edges, args.seed)
    if args.relabel_by_degree:       # diagnostic: nodes renumbered by descending degree (synth.relabel_by_degree)
        w = synth.relabel_by_degree(w)
    return w, 4096, 0.01, 0.001


def unique_rows(csr, n_nodes, ids_host):
    """U of SURVEY 8(d): |batch + every neighbour it has in any relation| (src/layers.py:226-227's unique_nodes)."""
    mark = np.zeros(n_nodes, dtype=bool)
    mark[ids_host] = True
    for indptr, idx in csr:
        for v in np.unique(ids_host):
            mark[idx[indptr[v]:indptr[v + 1]]] = True
    return int(mark.sum())


def algorithmic_bytes(graph, ids_host, counts_host, riders=None):
    """HBM bytes the choose+aggregate call must move for one batch:
    CSR row bounds + neighbour ids + neighbour class-0 scores + chosen feature rows + output.
    riders (the training call pcg_choose_gather_train, whose two launches also carry the next step's score pass, the label
    classifier's step and the deferred Adam update): dict(table_rows | csr + n_nodes, n_pos, n_params) - + SURVEY 8(d)'s U x F
    term: the feature rows of the batch's unique_nodes read once for scoring, and their scores (U is counted from the CSR when
    `table_rows` is not given; an engine that scores the whole table streams N >= U rows - the surplus is its own traffic, not
    algorithmic bytes), + the train positives' rows and keys, + the centres' rows, + one read of the gradient and a read and a
    write of theta, m, v (the per-tile slabs the gradient is summed from are an implementation's traffic, not counted)."""
    B = len(ids_host)
    total = 0
    for r in range(graph.R):
        D = int(graph.deg_host[r][ids_host].sum())
        S = int(counts_host[r].sum())
        total += 4 * (2 * B + D) + 4 * D + 4 * graph.feat_dim * S + 4 * B * graph.feat_dim
    if riders:
        F = graph.feat_dim
        U = riders["table_rows"] if "table_rows" in riders else unique_rows(riders["csr"], riders["n_nodes"], ids_host)
        total += U * (4 * F + 4) + riders["n_pos"] * (4 * F + 8) + B * (4 * F + 8) + 28 * riders["n_params"]
    return total


def cpu_baseline(w, trainer, cfg, batches, n_batches):
    """Time the oracle (CPU port of the reference algorithm, same Python/torch shape) on the
    first `n_batches` batches the GPU run used.  Test infrastructure used as the baseline
    leg only - nothing here feeds the GPU path."""
    from oracle import pcgnn_oracle as O
    torch.set_num_threads(min(os.cpu_count() or 1, 16))     # the GPU box gives one GPU a 16-core share
    adj = []
    for indptr, idx in w.csr:
        adj.append({v: set(idx[indptr[v]:indptr[v + 1]].tolist()) for v in range(w.n)})
    params = {k: v.detach().cpu().clone() for k, v in trainer.model.state_dict().items() if "features" not in k}
    om = O.OraclePCGNN(torch.from_numpy(w.X), adj, w.train_pos, params, cfg["rho"], cfg["alpha"], dense_mask=True)
    opt = O.make_adam(om, cfg["lr"], cfg["weight_decay"])
    nodes, spent = 0, 0.0
    batches = batches[:n_batches]
    n_batches = len(batches)
    for ids in batches:
        t0 = time.perf_counter()
        O.train_step(om, opt, ids.tolist(), w.labels[ids])
        spent += time.perf_counter() - t0
        nodes += len(ids)
    out = {"value": nodes / spent, "unit": "nodes/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{n_batches} batches ({nodes} nodes) of the same workload picked by the same sampler (the run's event-bracketed batches first), "
                     f"oracle/pcgnn_oracle.py train_step (dense-mask formulation as in the reference), {spent:.1f} s"}
    # calibration against the reference itself (tests/golden/make_cpu_calibration.py, run where the reference can be imported:
    # the build container, 8 cores): the port's speed relative to the reference on the same graph shape and batches
    cpath = os.path.join(ROOT, "tests", "golden", "cpu_calibration.json")
    key = {"yelp": "yelp", "amazon": "amazon"}.get(cfg.get("workload_key", ""), None)
    if key and os.path.exists(cpath):
        cal = json.load(open(cpath)).get(key)
        if cal:
            out["port_over_reference"] = cal["port_over_reference"]
            out["reference_equivalent"] = out["value"] / cal["port_over_reference"]
            out["calibration"] = (f"reference {cal['reference_nodes_per_s']:.0f} vs port {cal['oracle_nodes_per_s']:.0f} nodes/s on "
                                  f"{cal['workload']}, {cal['threads']} threads, build container")
    return out


def epoch_report(tr, n_epochs):
    """SURVEY 8(d)'s reporting form, measured AFTER the timed region: every epoch as two graph launches - the sampler (pick +
    shuffle + labels, src/model_handler.py:130-133) and the epoch's batches (the reference's per-batch window :143-155, summed
    as the reference sums it) - with HIP events before, between and after; medians over the epochs."""
    fz = tr.fused
    fz.stage_epoch(tr.pick_size, tr.batch_size)
    fz.take_prefetched()                         # (an epoch prepared ahead by the timed region is not used here: every epoch below samples)
    evs = []
    for e in range(n_epochs + 1):                # (the first one captures the batches-only graph: not counted)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        tr.start_epoch_staged()
        e1.record()
        fz.epoch_run()
        e2.record()
        evs.append((e0, e1, e2))
    torch.cuda.synchronize(tr.device)
    fz.check()
    win = np.array([a.elapsed_time(b) for _, a, b in evs[1:]]) * 1e-3
    inc = np.array([a.elapsed_time(b) for a, _, b in evs[1:]]) * 1e-3
    n = tr.pick_size
    return {"epochs": n_epochs, "nodes_per_epoch": n, "batches_per_epoch": tr.batches_per_epoch(),
            "reference_window_nodes_per_s_median": float(n / np.median(win)), "pick_inclusive_nodes_per_s_median": float(n / np.median(inc)),
            "reference_window_ms_per_epoch_median": float(np.median(win) * 1e3), "pick_inclusive_ms_per_epoch_median": float(np.median(inc) * 1e3),
            "how": "after the timed region; per epoch: event | sampler launch | event | one graph launch of all batches | event"}


def post_region_brackets(w, tr, cfg, n_pairs, verify, stride=61):
    """After the clock has stopped: `n_pairs` further epochs whose FIRST batch is launched kernel by kernel with HIP events around
    pcg_choose_gather_train (select_rows + gather_train_kernel), the rest of each epoch one graph launch - more samples for the
    roofline object than the timed region's few brackets.  verify: the first of these launches is also CHECKED - the scores it
    selects by are copied beforehand, the selection lists it wrote are read back from the workspace (plan slot + data part) right
    after it: the count law on every row (kept = deg > k + 1 ? k : deg neighbours, plus at most m = int(k * rho) train positives for
    a positive centre, src/layers.py:662-694; |set| == the kernel's count) and, on a strided sample of rows, the oracle's sets for
    the same scores (bit-exact).  The oracle is the checker: nothing it computes is timed or fed back.
    Returns (events, [(ids, counts)], verified | None)."""
    fz = tr.fused
    fz.stage_epoch(tr.pick_size, tr.batch_size)
    fz.take_prefetched()
    fz._prof = []
    used, verified = [], None
    B = tr.batch_size
    for i in range(n_pairs + (1 if verify else 0)):
        ids_all = tr.start_epoch_staged()
        ids = ids_all[:B]
        checked = verify and i == 0                   # the checked launch starts from a drained device: its timing is not a sample
        s0_before = None
        if checked:
            if not fz._fresh:
                fz._enqueue_refresh(fz._ep_touch(0))
            s0_before = fz.s0.clone()
        fz.epoch_step_timed(0, eager=True)
        if checked:
            fz._prof.pop()
            sets = fz.read_batch_lists(0)             # (synchronises; the lists of THIS launch: nothing has selected since)
            verified = verify_lists(w, tr, cfg, ids, sets, fz.last_counts.cpu().numpy(), s0_before, stride)
        else:
            used.append((ids.clone(), fz.last_counts.clone()))
        fz.epoch_run(first_step=1, flush=False)
    fz.flush()
    torch.cuda.synchronize(tr.device)
    fz.check()
    events, fz._prof = fz._prof, None
    return events, used, verified


def verify_lists(w, tr, cfg, ids, sets, cnt_h, s0_dev, stride):
    from oracle import pcgnn_oracle as O
    g = tr.graph
    s0_h = torch.from_numpy(s0_dev.cpu().numpy())
    pos = list(w.train_pos)
    pos_s = s0_h[torch.as_tensor(pos, dtype=torch.long)]
    rho, thr = float(cfg["rho"]), 0.5
    B = int(ids.numel())
    lab_h = tr.labels_i32[ids.long()].cpu().numpy()
    ids_h = ids.cpu().numpy().astype(np.int64)
    rows_law = rows_oracle = 0
    for r in range(g.R):
        deg = g.deg_host[r][ids_h].astype(np.int64)
        k = np.ceil(deg * thr).astype(np.int64)
        kept = np.where(deg > k + 1, k, deg)
        m = np.where(lab_h == 1, np.minimum((k * rho).astype(np.int64), len(pos)), 0)
        sizes = np.array([len(x) for x in sets[r]])
        if not (np.array_equal(sizes, cnt_h[r]) and (sizes >= kept).all() and (sizes <= kept + m).all()):
            raise SystemExit(f"verify: the count law fails in relation {r}")
        rows_law += B
        probe = list(range(0, B, stride))
        indptr, idx = w.csr[r]
        lists = [idx[indptr[v]:indptr[v + 1]].tolist() for v in ids_h[probe]]
        want = O.choose_sets(s0_h[torch.as_tensor(ids_h[probe], dtype=torch.long)], [int(lab_h[b]) for b in probe], lists,
                             [s0_h[torch.as_tensor(l, dtype=torch.long)] for l in lists], pos, pos_s, thr, rho, True)
        for b, ws_ in zip(probe, want):
            if sets[r][b] != ws_:
                raise SystemExit(f"verify: relation {r} row {b} (node {ids_h[b]}, degree {deg[b]}) differs from the oracle's set")
        rows_oracle += len(probe)
    return {"batches": 1, "rows_count_law": rows_law, "rows_vs_oracle": rows_oracle,
            "how": "after the timed region: the selection lists an event-bracketed pcg_choose_gather_train launch wrote (plan slot + "
                   "data part read back), against the scores that launch selected by (copied before it)"}


def run_partitioned(args, w, B, lr, wd, dev, dist, world, rank):
    """N > 1: destination-node partition, RCCL all-to-all of remote neighbour rows (once per window of steps) +
    gradient all-reduce (pc-gnn_amd/dist.py).  Weak scaling: every rank trains batches of B centres it owns."""
    from pcgnn_amd.dist import DistributedPCGNN
    cfg = dict(emb_size=args.emb, rho=args.rho, alpha=2.0, lr=lr, weight_decay=wd, batch_size=B, seed=args.seed)
    W = max(1, args.window)
    gloo = dist.get_backend() == "gloo"          # rehearsal of the N > 1 flow on one GPU (PCG_BENCH_BACKEND=gloo): collectives staged through the host
    d = DistributedPCGNN(w, cfg, dev, window=W, stage_host=gloo)
    try:
        return _run_partitioned(args, w, B, d, dev, dist, world, rank, W, gloo)
    finally:
        d.close()                           # (captured collectives go before their communicator does - also on an error)
        dist.destroy_process_group()


def _run_partitioned(args, w, B, d, dev, dist, world, rank, W, gloo):
    n_nodes, feat, n_rel = w.n, d.F, d.R

    def all_reduce(t, op=None):
        kw = {} if op is None else {"op": op}
        if gloo:
            c = t.cpu()
            dist.all_reduce(c, **kw)
            t.copy_(c)
        else:
            dist.all_reduce(t, **kw)

    def run_steps(first, n):
        """n steps in windows of W: one pick + one halo prefetch (two all-to-alls) per window, then per step one graph
        replay + the gradient all-reduce + Adam.  No host synchronisation anywhere."""
        k = 0
        while k < n:
            m = min(W, n - k)
            ids = d.pick_epoch(m * B, first + k)
            lab = d.labels_of(ids)
            d.train_window(ids, lab)
            k += m

    def barrier():
        dist.barrier()
        torch.cuda.synchronize(dev)

    # every batch position of a window has a graph of its own: the warm-up covers one whole window at least, so that none is
    # captured inside the timed region
    warmup = max(args.warmup, W)
    run_steps(0, warmup)
    # (rank 0 runs every 6th window step by step, its steps bracketed by events - the other windows are one graph launch each)
    prof = d.profile_select(max(args.event_every, 6)) if rank == 0 else None
    barrier()
    t0 = time.perf_counter()
    run_steps(warmup, args.steps)
    ta = time.perf_counter()
    d.flush()                               # the last step's Adam update (it otherwise rides at the head of the next step's graph)
    tb = time.perf_counter()
    torch.cuda.synchronize(dev)
    tc = time.perf_counter()
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("PCG_BENCH_DEBUG"):
        print(f"[debug] enqueue {ta - t0:.4f} flush {tb - ta:.4f} sync {tc - tb:.4f} barrier {t0 + elapsed - tc:.4f}", file=sys.stderr)
    d.check()                               # raises on every rank if any rank's exchange / lists went over capacity
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    fr, seen = d.feature_rows, d.halo.max_seen
    # rows a step's exchange moves per rank: the fixed-pitch buffers (what the all-to-alls carry) and the rows really asked for
    stats = torch.tensor([seen["halo_rows"], int((d.halo.req_in >= 0).sum().item()), fr["halo"] * d.g.X.shape[1] * 4],
                         dtype=torch.float64, device=dev)
    mem = torch.tensor([fr["owned"] + fr["train_pos"] + fr["halo"], seen["halo_rows"], fr["halo"], seen["rows_from_one_owner"],
                        fr["halo_pitch"]], dtype=torch.float64, device=dev)
    all_reduce(t, dist.ReduceOp.MAX)
    all_reduce(stats)
    all_reduce(mem, dist.ReduceOp.MAX)
    # what every rank believes the job to be: the smallest and the largest world size any rank's process group reports
    seen = torch.tensor([dist.get_world_size(), -dist.get_world_size()], dtype=torch.float64, device=dev)
    all_reduce(seen, dist.ReduceOp.MAX)
    world_seen = {"max": int(seen[0].item()), "min": int(-seen[1].item()), "backend": dist.get_backend()}
    elapsed = float(t.item())
    nodes_total = args.steps * B * world
    if rank == 0:
        hr, rm, en = (float(x) / world for x in stats.tolist())
        # roofline of this path's one launch per step - the captured graph (score pass over the rank's table, plan, train-pos
        # sort, select, look-up, gather, dense step, slab sum) - on rank 0: the select + gather bytes of algorithmic_bytes()
        # plus the score pass's table stream
        ms, ab = [], []
        g = d.g
        for e0, e1, ids_l, cnt in prof:
            ms.append(e0.elapsed_time(e1))
            ids_h = ids_l.cpu().numpy().astype(np.int64)
            # (SURVEY 8(d)'s U x F: the feature rows of the batch's unique nodes - centres and every neighbour, counted from this
            #  rank's CSR rows by global id - read once for scoring; not the whole extended table the rank's score pass streams)
            glob = [ids_h + d.part.lo] + [idx[ip[v]:ip[v + 1]] for ip, idx in d.csr_host for v in np.unique(ids_h)]
            U = int(np.unique(np.concatenate(glob)).size)
            ab.append(algorithmic_bytes(g, ids_h, cnt.view(g.R, -1).cpu().numpy()) + U * (4 * g.feat_dim + 4))
        avg_ms = float(np.mean(ms)) if ms else float("nan")
        achieved = float(np.mean(ab)) / (avg_ms * 1e-3) / 1e9 if ms else float("nan")
        # HBM-side bytes of the step graph's kernels from two PMC passes of this command at world size 1 (profiles/pmc_traffic.json)
        traffic, traffic_commit = None, None
        try:
            # (collected at world size 1: a rank's step graph does the same work at any world size but for the share of remote rows)
            tkey = f"{args.workload}_partitioned_w1" if (args.batch_size in (0, None, 1024) and args.emb == 64 and args.workload == "yelp") else None
            entry = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(tkey, {}) if tkey else {}
            traffic, traffic_commit = entry.get("choose_agg_bytes_per_launch"), entry.get("commit")
        except Exception:
            traffic, traffic_commit = None, None
        roofline = {"bound": "hbm", "kernel": "the step graph of rank 0: front (Adam from the all-reduced gradient || train-pos keys || score "
                                              "pass over owned + train-pos + halo rows), select_rows (sorts the keys), gather_chunks_dist "
                                              "(translates node ids to table rows as it reads the lists), dense_step (transposed activations, "
                                              "no slabs), wgrad (weight-gradient GEMMs over the batch)"
                                              + (", the gradient all-reduce (captured in the graph)" if d.collectives_in_graph else "")
                                              + ": one launch per step; bytes = select + gather + U x F scored rows (U = the batch's unique nodes)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_note": (f"profiles/pmc_traffic.json: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of the world-size-1 step graph's "
                                     f"kernels at commit {traffic_commit} (rank 0's step graph at world size 1)") if traffic is not None else
                                    "no PMC pass of this partitioned configuration has been collected (profiles/): null",
                    "avg_launch_ms": avg_ms, "algorithmic_bytes_per_launch": float(np.mean(ab)) if ab else None,
                    "launches_timed": len(ms)}
        out = {
            "metric": "sampled-nodes/sec", "value": nodes_total / elapsed, "unit": "nodes/s",
            "n_gpus": world, "steps": args.steps, "warmup": warmup, "warmup_requested": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "status_clean": True,             # d.check() above raised on every rank otherwise
            "world_size_seen": world_seen,
            "config": {"workload": f"{w.name} N={n_nodes} F={feat} R={n_rel} "
                                   f"edges={'/'.join(str(e) for e in w.meta['rel_edges'])}, PCGNN emb={args.emb} "
                                   f"batch={B}/GPU rho={args.rho}",
                       "global_batch": B * world, "parallelism": f"node-partition x{world} (in-edge balanced): id + feature-row "
                       f"all-to-all once per window of {W} steps (feature rows are immutable), every rank scores the rows it holds, "
                       "grad all-reduce per step (RCCL)", "engine": ("one graph replay per step (five launches + the captured all-reduce)"
                       if d.collectives_in_graph else "one graph replay (five launches) + one eager all-reduce per step") + "; no host "
                       "synchronisation", "collectives_in_graph": bool(d.collectives_in_graph), "window_steps": W,
                       "nodes_processed": int(nodes_total),
                       "per_rank_per_window": {"halo_rows_needed_max_seen": hr, "rows_served_last_window": rm,
                                               "all_to_all_row_bytes_fixed_pitch": en},
                       "feature_rows_per_rank_max": {"owned+train_pos+halo": int(mem[0].item()), "halo_capacity": int(mem[2].item()),
                                                     "halo_pitch": int(mem[4].item()), "rows_from_one_owner_max_seen": int(mem[3].item()),
                                                     "halo_rows_max_seen": int(mem[1].item()), "unpartitioned_table": int(n_nodes)}},
            "roofline": roofline,
        }
        print(json.dumps(out), flush=True)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N fresh processes - one per GPU - under
    torch.distributed.run and hand back their exit status.  This parent never touches the GPU (importing torch does not
    initialise it), it only waits; rank 0's JSON line goes straight to the inherited stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.run(cmd, env=env).returncode


def dry_run(world, rank):
    """PCG_BENCH_DRY=1: the launch plumbing alone (process group up, one all-reduce, one line) - no GPU, no library; what
    the CPU test of `--gpus N` exercises."""
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(1)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "ranks_seen": int(t.item())}))
    dist.destroy_process_group()


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and not launched:
        raise SystemExit(self_launch(args))          # (before any GPU call)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's process count and --gpus disagree")
    if os.environ.get("PCG_BENCH_DRY") == "1":
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        return dry_run(world, rank)
    # PCG_BENCH_BACKEND=gloo: rehearsal of the N > 1 flow on a box with fewer GPUs than ranks (ranks share GPUs, collectives
    # are staged through the host) - a correctness rehearsal, its numbers mean nothing
    backend = os.environ.get("PCG_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.force_partitioned:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import pcgnn_amd  # noqa: F401  (raises if libpcgnn_hip.so is missing - no fallback)
    from pcgnn_amd.handler import PCGNNTrainer

    partitioned = (world > 1 or args.force_partitioned) and args.engine != "dp"
    if partitioned and args.workload == "powerlaw":
        # every rank generates ITS shard of the graph (pure functions of the seed): no rank ever holds the whole graph
        from pcgnn_amd import synth
        w = synth.power_law_shard(args.nodes, args.edges, args.seed, world, rank)
        return run_partitioned(args, w, args.batch_size or 4096, 0.01, 0.001, dev, dist, world, rank)
    w, default_b, lr, wd = make_workload(args)
    B = args.batch_size or default_b
    if partitioned:
        return run_partitioned(args, w, B, lr, wd, dev, dist, world, rank)
    engine = args.engine or ("graph" if world == 1 else "fused")
    if engine == "dp":          # N>1 only: replicated graph, data-parallel batches (not the default)
        engine = "fused"
    cfg = dict(emb_size=args.emb, rho=args.rho, alpha=2.0, lr=lr, weight_decay=wd, batch_size=B,
               seed=args.seed + 1000 * rank, engine=engine, world_size=world,
               workload_key=args.workload if (B == default_b and args.emb == 64) else "", list_capacity=args.list_capacity)
    tr = PCGNNTrainer(w, cfg, dev)
    torch.manual_seed(args.seed)                     # identical initial weights on every rank
    with torch.no_grad():
        for p in tr.model.parameters():
            if p.requires_grad:
                torch.nn.init.xavier_uniform_(p) if p.dim() > 1 else torch.nn.init.zeros_(p)

    params = [p for p in tr.model.parameters() if p.requires_grad]

    def allreduce(flat):                             # gradients are already scaled by 1/(B*world)
        dist.all_reduce(flat)

    def one_step(ids, timed=False):
        if engine == "torch":
            labels = tr.labels_dev[ids.long()]
            tr.opt.zero_grad(set_to_none=True)
            loss = tr.model.loss(ids, labels)
            if dist is not None:
                loss = loss / world
            loss.backward()
            if dist is not None:
                flat = torch.cat([p.grad.reshape(-1) for p in params])
                dist.all_reduce(flat)
                o = 0
                for p in params:
                    p.grad.copy_(flat[o:o + p.numel()].view_as(p))
                    o += p.numel()
            tr.opt.step()
        elif dist is None:        # (graph engine without epoch graphs: one per-batch graph replay)
            tr.step(ids, timed)
        else:
            tr.fused.train_step(ids, tr.labels_i32[ids.long()], allreduce=allreduce)

    state = {"epoch": 0, "ids": None, "b": 0}
    nb = tr.batches_per_epoch()
    epoch_graphs = engine == "graph" and dist is None and not args.no_epoch_graphs   # ids/labels of whole epochs in static buffers, 1 launch per group of epochs

    def next_batch():
        if state["ids"] is None or state["b"] == nb:
            state["ids"] = tr.start_epoch(state["epoch"])      # pick + shuffle on the device
            state["epoch"] += 1
            state["b"] = 0
        b = state["b"]
        state["b"] += 1
        return state["ids"][b * B:(b + 1) * B]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    warmup = args.warmup
    inter = tr.model.inter1
    prof = tr.fused if tr.fused is not None else inter
    used_ev, counted = [], {"nodes": 0}
    # graph engine: K epochs are sampled, planned and replayed together - one sampler launch, one plan launch and one graph launch
    # per K * nb steps (the plan slots of a group must fit a few GB: large graphs fall back to K = 1)
    K = 1
    if epoch_graphs:
        slot = tr.fused._plan_bytes(B)
        # (default: four epochs, or as many as make a group of >= 24 steps - an Amazon-like epoch is two batches: measured 4 / 12 /
        #  24 epochs per launch 39.7 / 38.5 / 38.1 us per step; YelpChi-like, six batches an epoch, 4 / 8: 43.4 / 42.9)
        want = args.epochs_per_launch if args.epochs_per_launch > 0 else max(4, -(-24 // max(nb, 1)))
        K = max(1, min(want, (3 << 30) // max(2 * nb * slot, 1)))
    G = K * nb

    def run_steps(n_steps, measure):
        """n_steps training steps, batch by batch (every engine but the whole-epoch graph engine)."""
        for _ in range(n_steps):
            ids = next_batch()
            one_step(ids, True)
            if measure:
                used_ev.append((ids.clone(), prof.last_counts.clone()))
                counted["nodes"] += int(ids.numel())

    def run_groups(n_steps, measure):
        """n_steps training steps of the graph engine, in groups of K epochs (G = K * nb steps; the last group may be cut short).
        A group is ONE graph launch: sampler (pick + shuffle + labels of K epochs), plans of all its batches, every batch's three
        launches.  The second group of a phase and every `event_every`-th after it (the only group, if there is but one) is
        event-bracketed instead: one graph launch for everything but the group's LAST batch, which follows kernel by kernel with
        HIP events around pcg_choose_gather_train (the same kernels in the same order)."""
        fz = tr.fused
        k, gi = 0, 0
        first_timed = 1 if n_steps > G else 0
        while k < n_steps:
            r = min(G, n_steps - k)
            if (gi - first_timed) % args.event_every == 0:
                # (the bracketed batch is the group's LAST: its three kernels and the two event records are enqueued while the
                #  device is busy with the graph in front of them, so the bracket holds the two kernels and nothing else - and the
                #  host's launch latency is not paid on an idle device, as it was with the bracket at the head of the group)
                if r > 1:
                    tr.run_epoch_one_graph(flush=False, n_steps=r - 1, n_epochs=K)
                else:
                    tr.start_epoch_staged(K)
                fz.epoch_step_timed(r - 1, eager=True, flush=False)
                if measure:
                    lo, bsz = fz._ep_batches[r - 1]
                    used_ev.append((fz._ep_ids[lo:lo + bsz].clone(), fz.last_counts.clone()))
            else:
                tr.run_epoch_one_graph(flush=False, n_steps=r, n_epochs=K)
            if measure:
                counted["nodes"] += tr.nodes_of_steps(r)
            k += r
            gi += 1

    prof._prof = []
    if epoch_graphs:
        # one untimed pass through the timed region's own sequence: every graph the region replays (whole groups, the bracketed
        # group's tail, the last, shorter group) is captured here
        run_groups(args.steps, False)
        tr.fused.flush()
        warmup = args.steps
        if args.warmup > warmup:
            run_groups(args.warmup - warmup, False)
            warmup = args.warmup
    else:
        run_steps(warmup, False)
    prof._prof = []                                    # (start, end) HIP events around the select + aggregate launch
    barrier()
    t0 = time.perf_counter()
    (run_groups if epoch_graphs else run_steps)(args.steps, True)
    if tr.fused is not None:
        tr.fused.flush()                               # every step's update is applied inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0
    events, prof._prof = prof._prof, None
    # the status word, once, after the clock has stopped: a batch that did not fit its selection list makes the kernels select
    # nothing - a throughput for empty work must not be printed (raises -> non-zero exit)
    status_clean = None
    if tr.fused is not None:
        tr.fused.check()
        status_clean = True
    report = epoch_report(tr, args.report_epochs) if (epoch_graphs and args.report_epochs > 0 and rank == 0) else None
    post_ev, post_used, verified = [], [], None
    if epoch_graphs and rank == 0 and (args.post_brackets > 0 or args.verify_batches > 0):
        post_ev, post_used, verified = post_region_brackets(w, tr, cfg, args.post_brackets, args.verify_batches > 0)

    nodes_local = counted["nodes"]
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    n = torch.tensor([nodes_local], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    elapsed, nodes_total = float(t.item()), float(n.item())

    if rank == 0:
        n_in_region = len(events)
        events = list(events) + list(post_ev)
        used_ev = list(used_ev) + list(post_used)
        kern_ms = [a.elapsed_time(b) for a, b in events]
        batches_host = [i.cpu().numpy().astype(np.int64) for i, _ in used_ev]   # the event-bracketed batches: the CPU sample
        # (the score rider's term is SURVEY 8(d)'s U x F - the unique_nodes of the bracketed batch, counted from the host CSR -
        #  whether the engine scores the whole table or only the rows the next batch marks)
        riders = None
        if tr.fused is not None:
            riders = dict(csr=w.csr, n_nodes=w.n, n_pos=tr.graph.n_pos, n_params=tr.fused.n_params)
        abytes = [algorithmic_bytes(tr.graph, i.cpu().numpy().astype(np.int64), c.cpu().numpy(), riders) for i, c in used_ev]
        avg_ms = float(np.mean(kern_ms))
        achieved = float(np.mean(abytes)) / (avg_ms * 1e-3) / 1e9
        traffic, traffic_commit = None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                # the PMC passes were run on one batch size per workload: yelp 1024, amazon 256 (their defaults)
                tkey = (f"powerlaw_{args.nodes}_{args.edges}_b{B}" if args.workload == "powerlaw"
                        else args.workload if B == default_b else f"{args.workload}_b{B}")
                entry = json.load(open(tpath)).get(tkey, {})
                traffic, traffic_commit = entry.get("choose_agg_bytes_per_launch"), entry.get("commit")
            except Exception:
                traffic, traffic_commit = None, None
        out = {
            "metric": "sampled-nodes/sec", "value": nodes_total / elapsed, "unit": "nodes/s",
            "n_gpus": world, "steps": args.steps, "warmup": warmup, "warmup_requested": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "status_clean": status_clean,
            "timed_window": "every step = the reference's per-batch window (src/model_handler.py:143-155); the sampler - pick + "
                            "shuffle + label lookup (:130-133), one launch per epoch - runs INSIDE the timed region at every "
                            "epoch start, so `value` is the pick-inclusive figure (there is no separate pick-exclusive one); "
                            "warm-up = one untimed pass through the timed region's own launch sequence (so that every hipGraph it replays is captured before the clock starts), extended to --warmup if that is longer",
            "config": {"workload": f"{w.name} N={w.n} F={w.X.shape[1]} R={len(w.csr)} "
                                   f"edges={'/'.join(str(e) for e in w.meta['rel_edges'])} "
                                   f"endpoints={w.meta['endpoints']}, PCGNN emb={args.emb} batch={B} rho={args.rho}, "
                                   f"pick 2*|train_pos|={tr.pick_size}/epoch",
                       "global_batch": B * world, "parallelism": "single" if world == 1 else f"dp{world}-replicated-graph",
                       "engine": engine, "epochs_per_launch": K if epoch_graphs else None,
                       "nodes_processed": int(nodes_total)},
            "roofline": {"bound": "hbm", "kernel": "pcg_choose_gather_train: select_rows (sorts the train positives, steps the label classifier for this batch) + gather_train_kernel (gather || the other parameters' deferred Adam update || the NEXT step's score pass and train-pos keys); the plan is made per epoch beside the sampler, multi-chunk sums are finished in the dense kernel's prologue; bytes = select + gather + U x F feature rows scored + optimizer state (algorithmic_bytes)", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_from": ("profiles/pmc_traffic.json: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) of this command at commit "
                                          f"{traffic_commit}") if traffic is not None else None,
                         "avg_launch_ms": avg_ms, "min_launch_ms": float(np.min(kern_ms)), "max_launch_ms": float(np.max(kern_ms)),
                         "algorithmic_bytes_per_launch": float(np.mean(abytes)),
                         "launches_timed": len(kern_ms), "launches_timed_inside_the_timed_region": n_in_region,
                         "launches_how": "HIP events around the call's two kernels; the timed region brackets the first batch of its "
                                         "second group of epochs (its only one, if there is but one) and of every --event-every-th after it, --post-brackets further epochs "
                                         "after the clock has stopped add one bracket each (same launches, same graphs around them)"},
        }
        if report is not None:
            out["epoch_report"] = report
        if verified is not None:
            out["verified"] = verified
        if world == 1 and args.cpu_batches > 0:
            # the CPU sample: the event-bracketed batches of the run, topped up with further epochs of the same sampler (same
            # weights, same batch size) when a short run has bracketed fewer than --cpu-batches
            ep = 1 << 20
            while len(batches_host) < args.cpu_batches:
                extra = tr.start_epoch(ep).cpu().numpy().astype(np.int64)
                batches_host += [extra[i:i + B] for i in range(0, len(extra), B)]
                ep += 1
            out["cpu_baseline"] = cpu_baseline(w, tr, cfg, batches_host, args.cpu_batches)
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
