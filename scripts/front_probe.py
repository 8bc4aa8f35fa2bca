"""Where does front_b's time go?  Times (HIP events, many repetitions) plan pass 2 alone, the rank sort alone and both."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcgnn_amd
from pcgnn_amd import ops, synth

def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = torch.device("cuda", 0)
for name, w, B in (("yelp", synth.yelp_like(0), 1024), ("powerlaw", synth.power_law(2_000_000, 40_000_000, 0), 4096)):
    g = pcgnn_amd.DeviceGraph(w.X, w.csr, w.train_pos, dev)
    W = torch.randn(2, g.feat_dim, device=dev) * 0.1
    b = torch.zeros(2, device=dev)
    s0 = ops.score_table(g, W, b)
    keys = torch.empty(ops._lib.load().pcg_pos_sort_capacity(g.n_pos), dtype=torch.int64, device=dev)
    ids = torch.from_numpy(w.idx_train[:B].astype("int32")).to(dev)
    lab = torch.from_numpy(w.labels[w.idx_train[:B]].astype("int32")).to(dev)
    ws = ops.ChooseWorkspace(g, B)
    thr, rho = [0.5] * g.R, [0.5] * g.R
    ops.step_front_a(g, W, b, s0, 0, g.n_nodes, ids, lab, thr, rho, True, ws)
    print(name, "n_pos", g.n_pos, "rows", g.R * B)
    print("  front_a            %.2f us" % timeit(lambda: ops.step_front_a(g, W, b, s0, 0, g.n_nodes, ids, lab, thr, rho, True, ws)))
    print("  front_b plan only  %.2f us" % timeit(lambda: ops.step_front_b(g, s0, keys, ids, lab, thr, rho, False, ws)))
    print("  front_b plan+sort  %.2f us" % timeit(lambda: ops.step_front_b(g, s0, keys, ids, lab, thr, rho, True, ws)))
    print("  pos_sort alone     %.2f us" % timeit(lambda: ops.pos_sort(g, s0, keys)))
