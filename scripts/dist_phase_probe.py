"""Phase timing of the partitioned step at world size 1 (RCCL), with a sync after every phase."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from pcgnn_amd import synth
from pcgnn_amd.dist import DistributedPCGNN
w = synth.yelp_like(0)
d = DistributedPCGNN(w, dict(batch_size=1024), dev)
ops, g, part = d.ops, d.g, d.part
def T():
    torch.cuda.synchronize(); return time.perf_counter()
acc = {}
for it in range(12):
    ids = d.pick_epoch(1024, it); lab = d.labels_of(ids)
    t = [T()]
    ops.score_table(g, d.w_clf, d.b_clf, out=d.s0_send, row_begin=0, row_end=part.n_local); d._all_gather(d.s0_full, d.s0_send); t.append(T())
    keys = ops.pos_sort(g, d.s0_full, d.keys); center = d.s0_full[(ids.long() + part.lo)]; t.append(T())
    cnt = d.cnt[:g.R * 1024]
    ops.choose_select(g, ids, lab, d.s0_full, keys, d.thresholds, d.rho, True, d.ws, cnt, center_s0=center); t.append(T())
    total_dev = d.ws.view(0, torch.int64, g.R * 1024 + 1)[-1:]; t.append(T())
    d.halo.fetch_and_remap_device(d.ws.view(2, torch.int32, d.ws.list_capacity), total_dev, g); t.append(T())
    agg = d.agg.view(-1)[:g.R * 1024 * d.F].view(g.R, 1024, d.F)
    ops.aggregate_lists(g, g.X, 1024, d.ws, cnt, agg); t.append(T())
    t0 = T(); d.train_step(ids, lab); t1 = T()
    if it >= 2:
        for n, a, b in zip(["score+allgather", "sort+center", "select", "total.item", "halo exchange", "aggregate"], t[:-1], t[1:]):
            acc[n] = acc.get(n, 0) + (b - a)
        acc["whole train_step"] = acc.get("whole train_step", 0) + (t1 - t0)
for k, v in acc.items(): print(f"{k:18s} {v / 10 * 1e6:8.1f} us")
dist.destroy_process_group()
