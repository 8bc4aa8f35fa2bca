"""Where the partitioned step's time goes at world size 1 (RCCL): a window's prefetch, its plans, graph replays alone, the
collective alone - each timed over many repetitions with one synchronisation at the end."""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from pcgnn_amd import synth
from pcgnn_amd.dist import DistributedPCGNN
w = synth.yelp_like(0)
B, W = 1024, 8
d = DistributedPCGNN(w, dict(batch_size=B), dev, window=W)
print("collectives_in_graph", d.collectives_in_graph)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
def timed(name, fn, reps, per=1):
    fn(); fn()
    t0 = T()
    for _ in range(reps): fn()
    t1 = T()
    print(f"{name:44s} {(t1 - t0) / reps / per * 1e6:9.1f} us")
state = {"e": 0}
def window():
    ids = d.pick_epoch(W * B, state["e"]); state["e"] += W
    d.train_window(ids, d.labels_of(ids))
for _ in range(3): window()
timed("train_window / step", window, 20, W)
ids = d.pick_epoch(W * B, 1000); lab = d.labels_of(ids)
timed("pick_epoch + labels_of / window", lambda: d.labels_of(d.pick_epoch(W * B, 7)), 20)
timed("begin_window / window", lambda: d.begin_window(ids), 20)
d.win_ids[:W * B].copy_(ids); d.win_lab[:W * B].copy_(lab)
timed("plan of a window / window", lambda: d._plan(d.win_ids, d.win_lab, W * B, B, d.win_plans, True), 20)
grs = [d._graph_for(B, s)[0] for s in range(W)]
def replays():
    for g in grs: g.replay()
timed("graph replay alone / step", replays, 20, W)
timed("all_reduce(grad) alone", lambda: d._all_reduce(d.grad), 50)
def replays_ar():
    for g in grs:
        g.replay()
        if not d.collectives_in_graph: d._all_reduce(d.grad)
timed("replay + collective / step", replays_ar, 20, W)
d.check()
print("checked", flush=True)
d.close()
print("closed", flush=True)
dist.destroy_process_group()
print("destroyed", flush=True)
