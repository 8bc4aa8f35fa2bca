"""Multi-process CPU tests (gloo, world size 2) of the N>1 path's host logic: node partition (equal and in-edge
balanced), shard construction, pick weights with the global label frequencies, and the halo exchange of a window of
steps (request list in fixed per-owner ranges -> ids all-to-all -> rows all-to-all; list look-up; capacity flags).
The HIP kernels are not involved (they have no CPU path); GPU coverage is tests/test_dist_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def parts_owner(part, ids):
    return part.owner(np.maximum(ids, 0))


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import pcgnn_amd  # noqa: F401
        from pcgnn_amd import synth
        from pcgnn_amd.dist import HaloExchange, Partition, shard_pick_weights, shard_workload, total_degree
        w = synth.make_workload("t", 601, 25, (900, 4000), 0.2, seed=3)      # odd N: last shard is shorter
        for balanced in (False, True):
            part = Partition.balanced(total_degree(w.csr), world, rank) if balanced else Partition(w.n, world, rank)
            if not balanced and world == 2:
                assert part.n_max == 301 and part.n_local == (301 if rank == 0 else 300)
            sh = shard_workload(w, part)
            # shard CSR rows == the global rows, neighbour ids stay global
            for (ip, ix), (gip, gix) in zip(sh["csr"], w.csr):
                for v in (part.lo, part.lo + 7, part.hi - 1):
                    assert np.array_equal(ix[ip[v - part.lo]:ip[v - part.lo + 1]], gix[gip[v]:gip[v + 1]])
            assert np.array_equal(sh["X_local"], w.X[part.lo:part.hi])
            # pick weights: every rank uses the GLOBAL label frequencies (utils.py:276), so the ranks' weights are the
            # single-GPU sampler's weights restricted to their nodes
            y_all = w.labels[w.idx_train]
            mine = shard_pick_weights(w.labels[sh["idx_train_local"]], sh["homo_deg_train"], len(y_all), int(y_all.sum()))
            glob = np.diff(np.concatenate([[0.0], synth.pick_cum_weights(w)]))
            sel = (w.idx_train >= part.lo) & (w.idx_train < part.hi)
            np.testing.assert_allclose(mine, glob[sel], rtol=1e-12)
            P = len(sh["train_pos"])
            pitch = 200
            X_ext = torch.zeros(part.n_local + P + (world - 1) * pitch, 28)
            X_ext[:part.n_local, :25] = torch.from_numpy(sh["X_local"])
            X_ext[part.n_local:part.n_local + P, :25] = torch.from_numpy(sh["X_pos"])
            hx = HaloExchange(part, X_ext, sh["train_pos"], pitch)
            csr_t = [(torch.from_numpy(ip), torch.from_numpy(ix.astype(np.int64))) for ip, ix in sh["csr"]]
            tp = np.asarray(sh["train_pos"])
            rs = np.random.RandomState(10 + rank)
            for trial in range(3):
                # a window of centres (uneven across ranks, duplicates, once an empty one) -> the halo holds their remote neighbourhood
                n_c = [40, 7, 0][trial] if rank == 0 else [25, 0, 3 + rank][trial]
                centres = torch.from_numpy(rs.randint(0, part.n_local, size=n_c).astype(np.int32))
                hx.prefetch(csr_t, centres)
                nbrs = np.concatenate([ix[ip[c]:ip[c + 1]] for ip, ix in sh["csr"] for c in centres.tolist()] + [np.zeros(0, np.int32)])
                rem = ~((nbrs >= part.lo) & (nbrs < part.hi)) & ~np.isin(nbrs, tp)
                want = np.unique(nbrs[rem])
                req = hx.req_out.numpy()
                assert np.array_equal(np.sort(req[req >= 0]), want), "request list = the distinct remote neighbours"
                for j, o in enumerate(r for r in range(world) if r != rank):   # the j-th other rank's requests sit in its fixed range
                    mine = req[j * pitch:(j + 1) * pitch]
                    assert np.all((mine < 0) | (parts_owner(part, mine) == o))
                got = X_ext[hx.halo_base:][req >= 0][:, :25].numpy()
                assert np.array_equal(got, w.X[req[req >= 0]]), "halo rows = the requested global rows"
                # a step's list: any subset of the window's neighbourhood (+ owned ids, train positives, holes)
                pool = np.concatenate([nbrs, np.arange(part.lo, part.hi)[:50], tp[:20]]).astype(np.int32)
                orig = rs.choice(pool, size=min(300, 4 * len(pool)), replace=True) if len(pool) else np.zeros(0, np.int32)
                orig = orig.copy()
                orig[rs.rand(orig.size) < 0.1] = -1
                lst = torch.from_numpy(orig.copy())
                hx.lookup(lst)
                new = lst.numpy()
                keep = orig >= 0
                assert np.array_equal(new[~keep], orig[~keep])                       # holes untouched
                rows = X_ext[torch.from_numpy(new[keep]).long(), :25].numpy()
                assert np.array_equal(rows, w.X[orig[keep]]), "re-indexed rows must be the requested global rows"
                assert int(hx.overflow_word) == 0
            # an id outside the window: a hole + bit 4
            far = np.setdiff1d(np.arange(w.n), np.concatenate([np.arange(part.lo, part.hi), tp, hx.req_out.numpy()]))[:1].astype(np.int32)
            lst = torch.from_numpy(far.copy())
            hx.lookup(lst)
            assert int(lst[0]) == -1 and int(hx.overflow_word) == 4
            hx.overflow_word.zero_()
            # a pitch too small: the ids that do not fit get no slot (bit 2, holes at lookup) - and nobody waits for anybody:
            # the all-to-alls have fixed sizes
            small = HaloExchange(part, X_ext[:part.n_local + P + (world - 1) * 3], sh["train_pos"], 3)
            centres = torch.from_numpy(rs.randint(0, part.n_local, size=60).astype(np.int32))
            small.prefetch(csr_t, centres)
            assert int(small.overflow_word) & 2 and small.max_seen["rows_from_one_owner"] > 3
            req = small.req_out.numpy()
            assert (req >= 0).sum() <= (world - 1) * 3
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_halo_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_partition_covers_every_node_once():
    from pcgnn_amd.dist import Partition
    for n, world in ((10, 3), (45954, 8), (7, 8), (64, 8)):
        parts = [Partition(n, world, r) for r in range(world)]
        owned = np.concatenate([np.arange(p.lo, p.hi) for p in parts])
        assert np.array_equal(owned, np.arange(n))
        ids = np.arange(n)
        assert all(parts[int(o)].lo <= v < parts[int(o)].hi for v, o in zip(ids, parts[0].owner(ids)))


def test_balanced_partition_equalises_edges():
    """Partition.balanced: contiguous ranges whose CSR entry counts are nearly equal, covering every node once;
    owner() agrees with the ranges on numpy and torch inputs."""
    from pcgnn_amd import synth
    from pcgnn_amd.dist import Partition, expected_halo_rows, total_degree
    w = synth.make_workload("t", 5000, 8, (3000, 40000), 0.1, seed=1, skew=1.2)
    deg = total_degree(w.csr)
    for world in (2, 3, 8):
        parts = [Partition.balanced(deg, world, r) for r in range(world)]
        owned = np.concatenate([np.arange(p.lo, p.hi) for p in parts])
        assert np.array_equal(owned, np.arange(w.n))
        loads = np.array([deg[p.lo:p.hi].sum() + p.n_local for p in parts], dtype=np.float64)
        assert loads.max() <= loads.mean() * 1.05 + deg.max() + 1, loads
        eq = [Partition(w.n, world, r) for r in range(world)]
        eq_loads = np.array([deg[p.lo:p.hi].sum() for p in eq], dtype=np.float64)
        assert loads.max() - loads.min() <= eq_loads.max() - eq_loads.min() + deg.max()
        ids = np.arange(w.n)
        own = parts[0].owner(ids)
        assert all(parts[int(o)].lo <= v < parts[int(o)].hi for v, o in zip(ids[::37], own[::37]))
        assert np.array_equal(parts[0].owner(torch.from_numpy(ids)).numpy(), own)
    # halo capacity: from the batch's demand, never more than the remote nodes, a single row at world size 1
    d = [np.full(100, 40), np.full(100, 10)]
    wts = np.ones(100)
    assert expected_halo_rows(d, wts, 64, 1, 1000) == 1
    assert abs(expected_halo_rows(d, wts, 64, 2, 10 ** 9) - int(np.ceil(25 * 64 * 0.5 * 1.5 + 1024))) <= 1      # (few draws from many nodes: all distinct)
    assert expected_halo_rows(d, wts, 64, 8, 300) == 300
    e = 25 * 6400 * 0.5                     # as many draws as there are remote nodes: 1 - 1/e of them distinct
    assert expected_halo_rows(d, wts, 6400, 2, int(e)) == int(np.ceil(e * (1 - np.exp(-1.0)) * 1.5 + 1024))


def test_sharded_generator_agrees_across_ranks():
    """synth.power_law_shard: every rank builds its own rows of one global graph from pure functions of the seed - the shards
    of a 3-rank run are exactly the corresponding rows of the 1-rank run (CSR rows, features, labels, replicated train-pos
    block), the partition is in-edge balanced, and the ranks' pick weights use the same global label counts."""
    from pcgnn_amd import synth
    full = synth.power_law_shard(20000, 300000, 3, 1, 0, chunk=1 << 16)
    parts = [synth.power_law_shard(20000, 300000, 3, 3, r, chunk=1 << 16) for r in range(3)]
    assert full.bounds.tolist() == [0, 20000]
    for r, p in enumerate(parts):
        assert np.array_equal(p.bounds, parts[0].bounds) and p.rank == r
        lo, hi = int(p.bounds[r]), int(p.bounds[r + 1])
        for (ip, ix), (fip, fix) in zip(p.csr, full.csr):
            assert np.array_equal(ix, fix[fip[lo]:fip[hi]]) and np.array_equal(ip, fip[lo:hi + 1] - fip[lo])
            rows = np.repeat(np.arange(lo, hi), np.diff(ip))
            assert np.all(np.diff(ix)[np.diff(rows) == 0] > 0), "neighbour ids ascending inside a row"
            assert np.all(np.isin(np.arange(lo, hi), ix)), "self-loops kept"
        assert np.array_equal(p.X_local, full.X_local[lo:hi]) and np.array_equal(p.labels_local, full.labels_local[lo:hi])
        assert p.train_pos == full.train_pos and np.array_equal(p.X_pos, full.X_pos)
        assert (p.n_train, p.n_train_pos) == (full.n_train, full.n_train_pos)
        assert np.array_equal(p.idx_train_local, full.idx_train_local[(full.idx_train_local >= lo) & (full.idx_train_local < hi)])
    loads = np.array([sum(len(ix) for _, ix in p.csr) for p in parts], dtype=np.float64)
    assert loads.max() < 1.05 * loads.mean()
    assert abs(full.X_local.std() - 1.0) < 0.02 and 0.005 < full.labels_local.mean() < 0.02
