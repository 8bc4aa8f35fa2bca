"""Multi-process CPU tests (gloo, world size 2) of the N>1 path's host logic: node partition (equal and in-edge
balanced), shard construction, pick weights with the global label frequencies, and the halo exchange (ids all-to-all ->
rows all-to-all -> list re-index; capacity errors raised by every rank together).
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
            if not balanced:
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
            halo_rows = 400
            X_ext = torch.zeros(part.n_local + P + halo_rows, 28)
            X_ext[:part.n_local, :25] = torch.from_numpy(sh["X_local"])
            X_ext[part.n_local:part.n_local + P, :25] = torch.from_numpy(sh["X_pos"])
            hx = HaloExchange(part, X_ext, sh["train_pos"])
            tp = np.asarray(sh["train_pos"])
            rs = np.random.RandomState(10 + rank)
            for trial in range(3):
                n_list = [500, 37, 0][trial] if rank == 0 else [300, 0, 5][trial]   # uneven, incl. an empty list
                orig = rs.randint(0, w.n, size=n_list).astype(np.int32)
                orig[rs.rand(n_list) < 0.1] = -1                                    # holes
                lst = torch.from_numpy(orig.copy())
                n_halo = hx.fetch_and_remap(lst)
                new = lst.numpy()
                keep = orig >= 0
                assert np.array_equal(new[~keep], orig[~keep])                       # holes untouched
                got = X_ext[torch.from_numpy(new[keep]).long(), :25].numpy()
                assert np.array_equal(got, w.X[orig[keep]]), "re-indexed rows must be the requested global rows"
                rem = keep & ~((orig >= part.lo) & (orig < part.hi)) & ~np.isin(orig, tp)
                assert n_halo == len(np.unique(orig[rem])) and hx.last_stats["remote_entries"] == int(rem.sum())
            # a halo too small on ONE rank: every rank raises (same count matrix, same verdict) - nobody is left in a collective
            small = HaloExchange(part, X_ext[:part.n_local + P + (3 if rank == 0 else 400)], sh["train_pos"])
            lst = torch.from_numpy(rs.randint(0, w.n, size=200).astype(np.int32))
            with pytest.raises(RuntimeError, match="rank 0 needs"):
                small.fetch_and_remap(lst)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_halo_exchange_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
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
    assert expected_halo_rows(d, wts, 64, 2, 10 ** 9) == int(np.ceil(25 * 64 * 0.5 * 1.5 + 1024))
    assert expected_halo_rows(d, wts, 64, 8, 300) == 300


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
