"""Diagnostic: per-phase time of dense_step_kernel on one YelpChi-like batch (in-kernel stamps, 10-ns ticks)."""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcgnn_amd
from pcgnn_amd import synth, ops, _lib
from pcgnn_amd.handler import PCGNNTrainer
B = int(os.environ.get("PROBE_B", "1024"))
E = int(os.environ.get("PROBE_E", "64"))
w = synth.yelp_like(0)
tr = PCGNNTrainer(w, dict(engine="fused", batch_size=B, emb_size=E), torch.device("cuda", 0))
fz = tr.fused; g = fz.g; lib = _lib.load()
ids = tr.sampler.pick(B, 0); lab = tr.labels_i32[ids.long()]
n_tiles = lib.pcg_dense_n_tiles(B)
stamps = torch.zeros(n_tiles, 16, dtype=torch.int64, device="cuda")
TAIL = os.environ.get("PROBE_TAIL", "0") == "1"     # the training engine's launch (partial sums + in-kernel classifier Adam)
for it in range(4):
    if TAIL:
        keys = fz._enqueue_front_train(ids, lab, B)
        agg, _ = fz._enqueue_choose(ids, lab, B, keys, True, planned=True, combine=False)
    else:
        agg, _ = fz._enqueue_sample(ids, lab, B, True)
    torch.cuda.synchronize()
    if it == 3: lib.pcg_debug_set_dense_stamps(C.c_void_p(stamps.data_ptr()))
    if TAIL:
        fz._enqueue_tail(ids, lab, B, agg, True)
        fz.flush()
    else:
        fz._enqueue_dense(ids, lab, B, agg, True)
        fz._enqueue_adam(B)
    torch.cuda.synchronize()
lib.pcg_debug_set_dense_stamps(None)
st = stamps.cpu().numpy().astype(np.float64) * 0.01
t0 = st[:, 0].min()
names = {0: "start", 1: "staged", 2: "h_r", 11: "comb mfma", 3: "comb reduce", 4: "loss grads", 8: "dcomb", 9: "dW_cls", 10: "dW_clf", 5: "bias+sync", 6: "dh_r+dW_inter", 7: "dW_r"}
order = [0, 1, 2, 11, 3, 4, 8, 9, 10, 5, 6, 7]
print("tail" if TAIL else "plain", "B", B, "E", E, "tiles", n_tiles, "start skew %.2f us, last end %.2f us" % (st[:, 0].max() - t0, st[:, 7].max() - t0))
prev = st[:, 0]
for s in order[1:]:
    cur = st[:, s]
    print(f"  {names[s]:16s} mean {np.mean(cur - prev):6.2f} us   max {np.max(cur - prev):6.2f}")
    prev = cur
print("  total per tile   mean %.2f us  max %.2f" % (np.mean(st[:, 7] - st[:, 0]), np.max(st[:, 7] - st[:, 0])))
raw = stamps.cpu().numpy()
dc = (raw[:, 13] - raw[:, 12]).astype(np.float64)
dw = (raw[:, 7] - raw[:, 0]).astype(np.float64) * 10.0     # ns
print("  shader clock during the kernel: %.2f GHz (mean over tiles)" % np.mean(dc / dw))
