"""Time pcg_pos_sort alone (HIP events, many repetitions) for a few train-pos counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcgnn_amd
from pcgnn_amd import ops
dev = torch.device("cuda", 0)
n = 200000
X = np.zeros((n, 4), np.float32)
indptr = np.arange(n + 1, dtype=np.int64)
for P in (8000, 20000, 40000, 100000):
    rs = np.random.RandomState(P)
    g = pcgnn_amd.DeviceGraph(X, [(indptr, np.arange(n, dtype=np.int32))], rs.choice(n, size=P, replace=False).tolist(), dev)
    s0 = torch.from_numpy(rs.randn(n).astype(np.float32)).to(dev)
    keys = ops.pos_sort(g, s0)
    for _ in range(5):
        ops.pos_sort(g, s0, keys)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.pos_sort(g, s0, keys)
    e1.record()
    torch.cuda.synchronize()
    k = keys.cpu().numpy().view(np.uint64)[:P]
    assert np.all(k[1:] > k[:-1])
    print(f"P {P}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
