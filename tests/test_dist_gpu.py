"""Two and three ranks of the node-partitioned path on ONE GPU (collectives staged through the host over gloo)
against the single-GPU path on the same global batch.  -m gpu.

Shape: BASELINE configs[2]'s model (emb 128: the dense kernel variant that streams its weights) at batch 1024 per rank,
in-edge-balanced partition, a halo region much smaller than the node count (n_ext < N), a scenario whose halo is too small
on purpose and one that steps outside its prefetch window: every rank must raise together."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
EMB, BATCH, NODES = 128, 1024, 120000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pcgnn_amd
        from pcgnn_amd import ops, synth
        from pcgnn_amd.dist import DistributedPCGNN
        dev = torch.device("cuda", 0)
        w = synth.make_workload("t", NODES, 32, (20000, 100000, 400000), 0.15, seed=5, skew=1.5)
        B, E, F = BATCH, EMB, 32
        cfg = dict(emb_size=E, rho=0.5, alpha=2.0, lr=0.01, weight_decay=0.001, batch_size=B, seed=11)
        d = DistributedPCGNN(w, cfg, dev, stage_host=True, window=1)  # default: halo sized from the demand of a window of one batch
        part = d.part
        rows = d.feature_rows
        assert rows["owned"] + rows["train_pos"] + rows["halo"] < w.n, "the extended table must be smaller than the full one"
        assert abs(part.n_local - w.n / world) < 0.25 * w.n          # (in-edge balanced: not the equal split, not degenerate)
        ids_local = d.pick_epoch(B, 0)                       # duplicates included
        labels = d.labels_of(ids_local)
        theta0 = d.theta.clone()

        # ---- single-GPU reference on the full graph, same parameters ----
        g = pcgnn_amd.DeviceGraph(w.X, w.csr, w.train_pos, dev)
        o3 = d.lib.pcg_dense_param_offset(F, E, 3, 3, 0)
        o4 = d.lib.pcg_dense_param_offset(F, E, 3, 4, 0)
        Wc, bc = theta0[o3:o3 + 2 * F].view(2, F), theta0[o4:o4 + 2]
        s0 = ops.score_table(g, Wc, bc)
        keys = ops.pos_sort(g, s0)
        gids = (ids_local.long() + part.lo).to(torch.int32)
        agg_ref, cnt_ref = ops.choose_aggregate(g, gids, labels, s0, keys, [0.5] * 3, 0.5, True)

        # ---- partitioned: halo prefetch for these centres, scores of every held row, select, look-up, aggregate ----
        agg, cnt = d.forward_sample(ids_local, labels, True)
        torch.cuda.synchronize()
        held = d.row_gid[d.row_gid >= 0].long()
        assert held.numel() > part.n_local + len(w.train_pos), "the halo holds fetched rows"
        assert torch.equal(d.s0_full[held], s0[held]), "a row's score is the same on whichever rank computes it"
        assert torch.equal(cnt.view(3, B), cnt_ref)
        assert torch.equal(agg, agg_ref), "same lists, same order of summation: bitwise equal"
        seen = d.halo.max_seen
        assert 0 < seen["rows_from_one_owner"] <= rows["halo_pitch"] and seen["halo_rows"] <= rows["halo"]
        assert int((d.halo.req_in >= 0).sum()) > 0, "the other rank asked this one for rows"
        d.check()                                                   # nothing over capacity

        # ---- one train step: gradient = all-reduced; compare with the single-GPU gradient on the global batch ----
        d.train_step(ids_local, labels, use_graphs=False)           # (every rank in the same mode: a rank's first graph step runs a whole warm-up step, collectives included)
        torch.cuda.synchronize()
        all_ids = [torch.empty(B, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(all_ids, gids.cpu())
        all_lab = [torch.empty(B, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(all_lab, labels.cpu())
        gb_ids, gb_lab = torch.cat(all_ids).to(dev), torch.cat(all_lab).to(dev)
        agg_g, _ = ops.choose_aggregate(g, gb_ids, gb_lab, s0, keys, [0.5] * 3, 0.5, True)
        n_tiles = d.lib.pcg_dense_n_tiles(world * B)
        slabs = torch.empty(n_tiles, d.n_params, device=dev)
        grad = torch.empty(d.n_params, device=dev)
        lg, ce, rl = torch.empty(world * B, 2, device=dev), torch.empty(world * B, 2, device=dev), torch.empty(world * B, device=dev)
        st = ops._stream(dev)
        P_ = ops._p
        assert d.lib.pcg_dense_step(g.desc_ref(), P_(theta0), E, P_(gb_ids), P_(gb_lab), world * B, P_(agg_g), F, 2.0,
                                    1.0 / (world * B), P_(lg), P_(ce), None, P_(rl), P_(slabs), None, st) == 0
        assert d.lib.pcg_adam_step(None, None, None, P_(slabs), n_tiles, d.n_params, None, 0.01, 0.9, 0.999, 1e-8, 0.001,
                                   P_(grad), 0, st) == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(d.grad.cpu().numpy(), grad.cpu().numpy(), rtol=0, atol=2e-6)
        # logits of this rank's rows agree with the same rows in the global batch
        np.testing.assert_allclose(d.logits.cpu().numpy(), lg[rank * B:(rank + 1) * B].cpu().numpy(), rtol=0, atol=1e-6)
        # every rank holds the same parameters after the step
        th = [torch.empty(d.n_params) for _ in range(world)]
        d.flush()                                   # (the step's Adam update otherwise rides at the head of the next step)
        dist.all_gather(th, d.theta.cpu())
        assert all(torch.equal(th[0], t) for t in th[1:])
        assert not torch.equal(th[0], theta0.cpu())
        d.flush()                                   # a second flush without a new gradient changes nothing
        assert torch.equal(d.theta.cpu(), th[rank])
        # a window of two more batches through the captured step on both ranks (fresh centres): the first step warms up and
        # captures, the second replays
        ids2 = d.pick_epoch(2 * B, 1)
        d.train_window(ids2, d.labels_of(ids2))
        d.flush()
        torch.cuda.synchronize()
        dist.all_gather(th, d.theta.cpu())
        assert all(torch.equal(th[0], t) for t in th[1:]) and torch.isfinite(th[0]).all()

        d.check()
        # ---- a halo that is too small: the step itself never waits for the host (ids that do not fit become holes, a device
        # flag is raised); check() then raises on BOTH ranks (the flags are all-reduced: nobody is left inside a collective).
        # Rank 1 asks for a pitch of 16, rank 0 for plenty: the ranks agree on the larger - no overflow; then 16 on both. ----
        roomy = DistributedPCGNN(w, dict(cfg, batch_size=256), dev, stage_host=True, halo_pitch=(16 if rank == 1 else 30000))
        assert roomy.feature_rows["halo_pitch"] == 30000
        small = DistributedPCGNN(w, dict(cfg, batch_size=256), dev, stage_host=True, halo_pitch=16)
        ids3 = small.pick_epoch(256, 0)
        small.begin_window(ids3)
        small.train_step(ids3, small.labels_of(ids3), use_graphs=False)
        with pytest.raises(RuntimeError, match="over capacity.*halo exchange"):
            small.check()
        small.check()                                               # the flags were cleared by the look
        # ---- a step on centres its window does not cover: holes + a flag, and check() says so on both ranks ----
        ids4 = d.pick_epoch(B, 9)
        d.train_step(ids4, d.labels_of(ids4))
        with pytest.raises(RuntimeError, match="window did not fetch"):
            d.check()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_match_single_gpu(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def _worker_w1(port, q):
    """world size 1 through RCCL: the captured collectives and the whole-window graph against the step-by-step path"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    engines = []
    try:
        from pcgnn_amd import synth
        from pcgnn_amd.dist import DistributedPCGNN
        w = synth.make_workload("t", 40000, 32, (8000, 60000, 200000), 0.15, seed=5, skew=1.5)
        B, W = 256, 4
        cfg = dict(emb_size=64, rho=0.5, alpha=2.0, lr=0.01, weight_decay=0.001, batch_size=B, seed=11)
        os.environ["PCG_DIST_WINDOW_GRAPH"] = "1"
        a = DistributedPCGNN(w, cfg, dev, window=W)            # captured all-reduce; a window = one graph launch
        os.environ["PCG_DIST_WINDOW_GRAPH"] = "0"
        b = DistributedPCGNN(w, cfg, dev, window=W)            # captured all-reduce; one graph per step
        os.environ["PCG_DIST_GRAPH_COLLECTIVES"] = "0"
        c = DistributedPCGNN(w, cfg, dev, window=W)            # eager all-reduce behind every step's graph
        del os.environ["PCG_DIST_GRAPH_COLLECTIVES"], os.environ["PCG_DIST_WINDOW_GRAPH"]
        engines = [a, b, c]
        assert a.collectives_in_graph and a.window_graphs and b.collectives_in_graph and not b.window_graphs
        assert not c.collectives_in_graph
        assert torch.equal(a.theta, b.theta) and torch.equal(a.theta, c.theta)
        for d in engines:
            for k in range(4):                                  # the first window of a size runs step by step, the others replay
                ids = d.pick_epoch(W * B, k)
                d.train_window(ids, d.labels_of(ids))
            ids = d.pick_epoch(2 * B + 100, 9)                  # a shorter window with a partial last batch: another graph
            for _ in range(2):
                d.train_window(ids, d.labels_of(ids))
            d.flush()
            d.check()
        torch.cuda.synchronize()
        assert ("window", W * B) in a._graphs and ("window", 2 * B + 100) in a._graphs
        for name in ("theta", "m", "v", "step_counter", "grad"):
            assert torch.equal(getattr(a, name), getattr(b, name)), name + " (window graph vs one graph per step)"
            assert torch.equal(getattr(a, name), getattr(c, name)), name + " (captured vs eager all-reduce)"
        assert not torch.isnan(a.theta).any() and int(a.step_counter.item()) == 4 * W + 2 * 3
        q.put((0, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((0, traceback.format_exc()))
    finally:
        for d in engines:
            d.close()                                           # (captured collectives go before their communicator does)
        dist.destroy_process_group()


def test_window_graph_and_captured_collectives_world_size_1():
    """RCCL at world size 1 (what a one-GPU box can run): the step's all-reduce captured inside its hipGraph, and a whole window -
    halo exchange, plans, steps, all-reduces - as ONE graph launch, leave bit for bit what one graph per step + an eager all-reduce
    leave; windows of two sizes (the second with a partial last batch)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_w1, args=(_free_port(), q))
    p.start()
    rank, msg = q.get(timeout=600)
    p.join(timeout=120)
    if p.is_alive():        # (a teardown that does not return must not hang the test run)
        p.kill()
        raise AssertionError("the worker did not exit after its result: " + msg)
    assert msg == "ok", msg
