"""debug: which rows of the power-law full-size batch come back without a list"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcgnn_amd
from pcgnn_amd import ops, synth
from pcgnn_amd.handler import PCGNNTrainer
dev = torch.device("cuda", 0)
w = synth.power_law(200_000, 4_000_000, 0, max_share=5e-3)
B = 4096
tr = PCGNNTrainer(w, dict(engine="fused", batch_size=B, rho=0.5), dev)
g, fz = tr.graph, tr.fused
ids = tr.start_epoch(0)[:B].contiguous()
top = np.unique(np.concatenate([np.argsort(np.diff(ip))[-3:] for ip, _ in w.csr])).astype(np.int32)
ids[:len(top)] = torch.from_numpy(top).to(dev)
lab = tr.labels_i32[ids.long()]
s0 = ops.score_table(g, fz.w_clf, fz.b_clf)
keys = ops.pos_sort(g, s0)
R = g.R
print("max_degree", g.max_degree, "n_pos", g.n_pos)
for train in (False, True):
    ws = ops.ChooseWorkspace(g, B)
    print("list_capacity", ws.list_capacity, "clipped", ws.clipped, "ws bytes", ws.buf.numel())
    cnt = torch.full((R, B), -7, dtype=torch.int32, device=dev)
    agg, cnt = ops.choose_aggregate(g, ids, lab if train else None, s0, keys if train else None, [0.5] * R, 0.5, train, ws=ws, cnt=cnt)
    torch.cuda.synchronize()
    print("status", int(ws.status.item()))
    begin = ws.view(0, torch.int64, R * B + 1).cpu().numpy()
    length = ws.view(1, torch.int32, R * B).cpu().numpy()
    cnt_h = cnt.cpu().numpy().reshape(-1)
    ids_h = ids.cpu().numpy()
    deg = np.concatenate([np.diff(ip)[ids_h] for ip, _ in w.csr])
    bad = np.flatnonzero((length == 0) | (cnt_h == -7))
    print("train", train, "rows", R * B, "bad", len(bad), "total entries", begin[-1])
    print("bad rows deg", deg[bad][:40], "rows", bad[:40])
    cnts = ws.view(5, torch.int32, 16).cpu().numpy()
    print("counters", cnts)
    print("tier census", (deg <= 128).sum(), ((deg > 128) & (deg <= 512)).sum(), ((deg > 512) & (deg <= 4096)).sum(), (deg > 4096).sum())
