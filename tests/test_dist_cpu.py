"""Multi-process CPU tests (gloo, world size 2) of the N>1 path's host logic: node partition,
shard construction, and the halo exchange (ids all-to-all -> rows all-to-all -> list re-index).
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
        from pcgnn_amd.dist import HaloExchange, Partition, shard_workload
        w = synth.make_workload("t", 601, 25, (900, 4000), 0.2, seed=3)      # odd N: last shard is shorter
        part = Partition(w.n, world, rank)
        assert part.n_per == 301 and part.n_local == (301 if rank == 0 else 300)
        sh = shard_workload(w, part)
        # shard CSR rows == the global rows, neighbour ids stay global
        for (ip, ix), (gip, gix) in zip(sh["csr"], w.csr):
            for v in (part.lo, part.lo + 7, part.hi - 1):
                assert np.array_equal(ix[ip[v - part.lo]:ip[v - part.lo + 1]], gix[gip[v]:gip[v + 1]])
        assert np.array_equal(sh["X_local"], w.X[part.lo:part.hi])
        P = len(sh["train_pos"])
        halo_rows = 400
        X_ext = torch.zeros(part.n_local + P + halo_rows, 28)
        X_ext[:part.n_local, :25] = torch.from_numpy(sh["X_local"])
        X_ext[part.n_local:part.n_local + P, :25] = torch.from_numpy(sh["X_pos"])
        posmap = torch.full((w.n,), -1, dtype=torch.int32)
        posmap[torch.as_tensor(sh["train_pos"])] = torch.arange(P, dtype=torch.int32)
        hx = HaloExchange(part, X_ext, P, posmap)
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
            rem = keep & ~((orig >= part.lo) & (orig < part.hi)) & (posmap.numpy()[np.clip(orig, 0, None)] < 0)
            assert n_halo == len(np.unique(orig[rem])) and hx.last_stats["remote_entries"] == int(rem.sum())
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
