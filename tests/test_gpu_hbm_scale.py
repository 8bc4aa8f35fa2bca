"""An HBM-resident workload inside ``-m gpu``: power-law graph, 2 M nodes / 40 M edges (256 MB feature table, ~690 MB of CSR:
far beyond the 4 MiB L2s, beyond the Infinity Cache with the workspace), batch 4096 - the shape of BASELINE.json configs[3] at
the largest size whose generation takes seconds.  Everything the small graphs of test_gpu_parity.py cannot reach runs here at
the size it was written for: the touched-rows score engine (forced on), the grid-wide marking pass of hub rows, rows beyond
the LDS key capacity (select_long_rows), the bucket sort of more than 16 384 train positives, two whole epochs through the
graph engine - checked by size-independent properties (the count law of src/layers.py:662-694 on every row) and the oracle's
sets on a strided sample of rows, from the lists a training launch itself wrote."""
import numpy as np
import pytest
import torch

from oracle import pcgnn_oracle as O

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda", 0)


def check_lists(w, g, ids, lab_h, sets, cnt_h, s0_h, rho, stride):
    """count law on every row; the oracle's sets (same scores) on every stride-th row.  Returns the rows checked both ways."""
    pos = list(w.train_pos)
    pos_s = s0_h[torch.as_tensor(pos, dtype=torch.long)]
    thr = 0.5
    ids_h = ids.cpu().numpy().astype(np.int64)
    B = len(ids_h)
    n_law = n_oracle = 0
    for r in range(g.R):
        deg = g.deg_host[r][ids_h].astype(np.int64)
        k = np.ceil(deg * thr).astype(np.int64)
        kept = np.where(deg > k + 1, k, deg)
        m = np.where(lab_h == 1, np.minimum((k * rho).astype(np.int64), len(pos)), 0)
        sizes = np.array([len(x) for x in sets[r]])
        assert np.array_equal(sizes, cnt_h[r]), f"relation {r}: |set| != the kernel's count"
        assert (sizes >= kept).all() and (sizes <= kept + m).all(), f"relation {r}: the count law"
        n_law += B
        # the longest rows of the batch always, then a strided sample
        probe = sorted(set(np.argsort(-deg)[:3].tolist()) | set(range(0, B, stride)))
        indptr, idx = w.csr[r]
        lists = [idx[indptr[v]:indptr[v + 1]].tolist() for v in ids_h[probe]]
        want = O.choose_sets(s0_h[torch.as_tensor(ids_h[probe], dtype=torch.long)], [int(lab_h[b]) for b in probe], lists,
                             [s0_h[torch.as_tensor(l, dtype=torch.long)] for l in lists], pos, pos_s, thr, rho, True)
        for b, ws_ in zip(probe, want):
            assert sets[r][b] == ws_, f"relation {r} row {b} (node {ids_h[b]}, degree {deg[b]}) differs from the oracle's set"
        n_oracle += len(probe)
    return n_law, n_oracle


def test_powerlaw_2m_touched_rows_long_rows_bucket_sort(monkeypatch):
    import pcgnn_amd  # noqa: F401
    from pcgnn_amd import synth
    from pcgnn_amd.handler import PCGNNTrainer
    monkeypatch.setenv("PCG_TOUCHED", "1")                  # (the engine's own switch is at 512 MB of table: this one has 256)
    B, rho = 4096, 0.5
    # 2.5 % positives -> ~20 000 train positives (> 16 384: the bucket sort); hubs up to 5e-4 of a relation's endpoints
    w = synth.power_law(2_000_000, 40_000_000, seed=5, pos_rate=0.025, max_share=5e-4)
    tr = PCGNNTrainer(w, dict(engine="graph", batch_size=B, emb_size=64, rho=rho, seed=5), dev())
    fz, g = tr.fused, tr.graph
    assert fz.touched_on
    assert g.n_pos > 16384, g.n_pos
    assert g.max_degree > 10240, g.max_degree               # rows for select_long_rows exist in the graph
    nb = tr.batches_per_epoch()
    assert nb >= 4
    theta0 = fz.theta.clone()
    # two whole epochs, one graph launch each (sampler, plans, touched-row maps, every batch's three launches)
    for e in range(2):
        tr.run_epoch_one_graph(flush=(e == 1))
    torch.cuda.synchronize()
    fz.check()                                              # no list overflow, no id out of range, no in-kernel wait ran out
    assert torch.isfinite(fz.theta).all() and not torch.equal(theta0, fz.theta)
    assert int(fz.step_counter.item()) == 2 * nb
    # a third epoch: every batch kernel by kernel, its lists read back right after its own select launch
    ids_all = tr.start_epoch_staged()
    seen_long = seen_hub_map = 0
    for b in range(nb):
        lo = b * B
        ids = ids_all[lo:lo + B]
        if not fz._fresh:
            fz._enqueue_refresh(fz._ep_touch(b))
        s0_before = fz.s0.clone()
        fz.epoch_step_timed(b, eager=True)
        if b in (0, 1, nb - 1):                             # (reading lists back costs seconds: three batches, the partial last one included)
            sets = fz.read_batch_lists(b)
            lab_h = tr.labels_i32[ids.long()].cpu().numpy()
            cnt_h = fz.last_counts.cpu().numpy()
            s0_h = torch.from_numpy(s0_before.cpu().numpy())
            n_law, n_or = check_lists(w, g, ids, lab_h, sets, cnt_h, s0_h, rho, stride=97)
            assert n_law == g.R * ids.numel() and n_or >= 3 * g.R
            degs = np.stack([g.deg_host[r][ids.cpu().numpy()] for r in range(g.R)])
            seen_long += int((degs > 10240).sum())
            seen_hub_map += int((degs > 4096).sum())
    torch.cuda.synchronize()
    fz.check()
    # the paths this test exists for were populated: rows beyond the LDS key capacity, hub rows of the marking pass
    assert seen_long > 0 and seen_hub_map > 0, (seen_long, seen_hub_map)
    # touched-rows scoring == whole-table scoring on every row a batch can read: the last batch's map against a full pass
    from pcgnn_amd import ops
    fz.flush()
    full = ops.score_table(g, fz.w_clf, fz.b_clf)
    ids = ids_all[(nb - 1) * B:]
    fz.clf_next.copy_(fz.theta[fz.n_rest:])
    fz._enqueue_refresh(fz._ep_touch(nb - 1))               # scores of the last batch's touched rows with the same classifier
    torch.cuda.synchronize()
    stride = fz._touch_stride
    marks = fz._ep_sets[fz._cur]["touched"][(nb - 1) * stride:(nb - 1) * stride + g.n_nodes].bool()
    assert bool(marks[ids.long()].all()) and int(marks.sum()) > ids.numel()
    assert torch.equal(fz.s0[marks], full[marks]), "a touched row's score is bit for bit the whole-table pass's"
