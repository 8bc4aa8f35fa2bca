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
    # the one-launch bucket sort over raw keys (16384 < P <= 131072)
    from pcgnn_amd import _lib
    lib = _lib.load()
    if lib.pcg_pos_sort_one_launch(P):
        cap = int(lib.pcg_pos_sort_capacity(P)) // 2
        want = keys.clone()
        k2 = torch.zeros_like(want)
        k2[cap:cap + P] = want[:P][torch.from_numpy(rs.permutation(P)).to(dev)]
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        run = lambda: _lib.check(lib.pcg_pos_sort_raw(g.desc_ref(), ops._p(k2), ops._p(status), ops._stream(dev)), "raw")
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        assert torch.equal(k2[:cap], want[:cap]) and int(status.item()) == 0
        print(f"P {P}: one launch {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
