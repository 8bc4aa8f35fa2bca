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
    t = [T(), T()]
    t0 = T(); d.train_step(ids, lab); t1 = T()
    t2 = T(); d.train_step(ids, lab, use_graphs=False); t3 = T()
    if it >= 2: acc['eager train_step'] = acc.get('eager train_step', 0) + (t3 - t2)
    if it >= 2:
        acc["whole train_step"] = acc.get("whole train_step", 0) + (t1 - t0)
for k, v in acc.items(): print(f"{k:18s} {v / 10 * 1e6:8.1f} us")
dist.destroy_process_group()
