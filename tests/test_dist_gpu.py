"""Two ranks of the node-partitioned path on ONE GPU (collectives staged through the host over gloo)
against the single-GPU path on the same global batch.  -m gpu."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


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
        w = synth.make_workload("t", 3001, 32, (4000, 30000, 90000), 0.15, seed=5)
        B = 96
        cfg = dict(emb_size=32, rho=0.5, alpha=2.0, lr=0.01, weight_decay=0.001, batch_size=B, seed=11)
        d = DistributedPCGNN(w, cfg, dev, stage_host=True)
        part = d.part
        ids_local = d.pick_epoch(B, 0)                       # duplicates included
        labels = d.labels_of(ids_local)
        theta0 = d.theta.clone()

        # ---- single-GPU reference on the full graph, same parameters ----
        g = pcgnn_amd.DeviceGraph(w.X, w.csr, w.train_pos, dev)
        o3 = d.lib.pcg_dense_param_offset(32, 32, 3, 3, 0)
        o4 = d.lib.pcg_dense_param_offset(32, 32, 3, 4, 0)
        Wc, bc = theta0[o3:o3 + 64].view(2, 32), theta0[o4:o4 + 2]
        s0 = ops.score_table(g, Wc, bc)
        keys = ops.pos_sort(g, s0)
        gids = (ids_local.long() + part.lo).to(torch.int32)
        agg_ref, cnt_ref = ops.choose_aggregate(g, gids, labels, s0, keys, [0.5] * 3, 0.5, True)

        # ---- partitioned: scores all-gather, select, halo exchange, aggregate ----
        agg, cnt = d.forward_sample(ids_local, labels, True)
        torch.cuda.synchronize()
        assert torch.equal(d.s0_full[:w.n], s0), "all-gathered scores must equal the single-GPU score table"
        assert torch.equal(cnt.view(3, B), cnt_ref)
        assert torch.equal(agg, agg_ref), "same lists, same order of summation: bitwise equal"
        stats = d.halo.last_stats
        assert stats["halo_rows"] > 0 and stats["rows_served"] > 0

        # ---- one train step: gradient = all-reduced; compare with the single-GPU gradient on the global batch ----
        d.train_step(ids_local, labels, use_graphs=(rank == 0))     # one rank through the captured graphs, one eagerly
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
        assert d.lib.pcg_dense_step(g.desc_ref(), P_(theta0), 32, P_(gb_ids), P_(gb_lab), world * B, P_(agg_g), 32, 2.0,
                                    1.0 / (world * B), P_(lg), P_(ce), None, P_(rl), P_(slabs), None, st) == 0
        assert d.lib.pcg_adam_step(None, None, None, P_(slabs), n_tiles, d.n_params, None, 0.01, 0.9, 0.999, 1e-8, 0.001,
                                   P_(grad), 0, st) == 0
        torch.cuda.synchronize()
        np.testing.assert_allclose(d.grad.cpu().numpy(), grad.cpu().numpy(), rtol=0, atol=2e-6)
        # logits of this rank's rows agree with the same rows in the global batch
        np.testing.assert_allclose(d.logits.cpu().numpy(), lg[rank * B:(rank + 1) * B].cpu().numpy(), rtol=0, atol=1e-6)
        # every rank holds the same parameters after the step
        th = [torch.empty(d.n_params) for _ in range(world)]
        dist.all_gather(th, d.theta.cpu())
        assert torch.equal(th[0], th[1])
        assert not torch.equal(th[0], theta0.cpu())
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_two_ranks_match_single_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"
