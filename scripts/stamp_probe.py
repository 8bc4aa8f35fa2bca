"""Diagnostic: per-phase time of the select kernel and of the dense kernel on one YelpChi-like batch (in-kernel stamps).
PROBE_SORT=select (default: the train positives are sorted inside select_rows) | separate (pcg_pos_sort first)."""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcgnn_amd
from pcgnn_amd import synth, ops, _lib
from pcgnn_amd.handler import PCGNNTrainer
B = int(os.environ.get("PROBE_B", "1024"))
E = int(os.environ.get("PROBE_E", "64"))
if os.environ.get("PROBE_WORKLOAD", "yelp") == "powerlaw":
    w = synth.power_law(int(os.environ.get("PROBE_NODES", "2000000")), int(os.environ.get("PROBE_EDGES", "40000000")), 0)
else:
    w = synth.yelp_like(0)
tr = PCGNNTrainer(w, dict(engine="fused", batch_size=B, emb_size=E), torch.device("cuda", 0))
fz = tr.fused; g = fz.g; lib = _lib.load()
ids_all = tr.start_epoch(0)
ids = ids_all[:B].contiguous(); lab = tr.labels_i32[ids.long()]
rows = g.R * B
in_select = os.environ.get("PROBE_SORT", "select") == "select"
stamps = torch.zeros(rows + 2, 8, dtype=torch.int64, device="cuda")
plan = fz._enqueue_plan_one(ids, lab, B, True)
torch.cuda.synchronize()
lib.pcg_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
plan = fz._enqueue_plan_one(ids, lab, B, True)
torch.cuda.synchronize()
lib.pcg_debug_set_stamps(None)
pl = stamps[rows].cpu().numpy().astype(np.float64) * 0.01
print("plan (last workgroup of the batch, us): own rows + totals published %.2f, predecessors' totals %.2f, writes %.2f" % (pl[1] - pl[0], pl[2] - pl[1], pl[3] - pl[2]))
stamps.zero_()
for it in range(3):
    if in_select: fz._enqueue_refresh()
    else: keys = fz._enqueue_scores(True)
    torch.cuda.synchronize()
    if it == 2: lib.pcg_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    if in_select: agg, _ = fz._enqueue_choose_train(ids, lab, B, plan, True)        # (select + the classifier's step | gather + the rest)
    else: agg, _ = fz._enqueue_choose(ids, lab, B, keys, True, plan, sort_in_kernel=False)
    torch.cuda.synchronize()
lib.pcg_debug_set_stamps(None)
extra = stamps[rows].cpu().numpy().astype(np.float64) * 0.01
raw = stamps[:rows].cpu().numpy()
blk = (raw[:, 0] >> 54) & 0x3FF                                   # workgroup that ran the row
raw[:, 0] &= (1 << 54) - 1
st = raw.astype(np.float64) * 0.01   # us
deg = np.stack([g.deg_host[r][ids.cpu().numpy()] for r in range(g.R)]).reshape(-1)
lab_h = lab.cpu().numpy()
pos = np.tile(lab_h == 1, g.R)
ran = st[:, 0] > 0                                                 # (groups of four short rows leave no stamps)
t0 = st[ran, 0].min()
print("sort:", "inside select_rows" if in_select else "separate launch", "| rows with stamps", int(ran.sum()), "of", rows)
if in_select:
    print("   sort workgroup 0 started at %.2f, the LAST key group was published at %.2f us (after the first row start)" % tuple(extra[4:6] - t0))
if in_select and extra[3] > 0:
    print("   the label classifier's step (one workgroup): from %.2f to %.2f us" % tuple(extra[2:4] - t0))
    ph = stamps[rows + 1].cpu().numpy().astype(np.float64) * 0.01
    print("      classifier, state, ids staged at %.2f | rows done %.2f | partials in LDS %.2f" % tuple(ph[0:3] - t0))
wt = st[:, 7][ran & pos & (st[:, 7] > 0)] - t0
if wt.size:
    w3 = (st[:, 7] - st[:, 3])[ran & pos & (st[:, 7] > 0)]
    print("   rows with minority picks left the wait for the sorted keys at: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f us; time inside the wait p50 %.2f max %.2f" % (
        wt.min(), np.percentile(wt, 10), np.percentile(wt, 50), np.percentile(wt, 90), wt.max(), np.percentile(w3, 50), w3.max()))
names = ["rec+keys(1)", "kth(2)", "compact+list(3)", "min-search(4)", "min-resolve(5)", "tail(6)"]
for lo, hi, tier in ((0, 64, "T0"), (64, 512, "T1"), (512, 4096, "T4"), (4096, 1 << 30, "T16")):
    for what, msk in (("neg", ~pos), ("pos", pos)):
        sel = (deg > lo) & (deg <= hi) & ran & msk
        if not sel.any(): continue
        s = st[sel]
        print(f"{tier} {what}: rows {sel.sum()}, deg mean {deg[sel].mean():.0f} max {deg[sel].max()}, start {s[:,0].min()-t0:.1f}..{s[:,0].max()-t0:.1f} us, end max {s[:,6].max()-t0:.1f} us")
        prev = s[:, 0]
        for i, n in enumerate(names, start=1):
            cur = np.where(s[:, i] > 0, s[:, i], prev)
            dt = cur - prev
            print(f"   {n:18s} mean {dt.mean():7.2f} us  p95 {np.percentile(dt,95):7.2f}  max {dt.max():7.2f}")
            prev = cur
        tot = s[:, 6] - s[:, 0]
        print(f"   total per row      mean {tot.mean():7.2f} us  max {tot.max():7.2f}")
end = st[ran, 6] - t0
print("row end times (us): p50 %.1f p90 %.1f p99 %.1f max %.1f" % (np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99), end.max()))
order = np.flatnonzero(ran)[np.argsort(-(st[ran, 6] - t0))][:10]
print("last rows to finish: row rel b deg label wg | start keys kth compact search resolve tail | end")
for rr in order:
    r_, b_ = rr // B, rr % B
    ph = [st[rr, i] - st[rr, i - 1] if st[rr, i] > 0 else 0.0 for i in range(1, 7)]
    print(f"  {rr:5d} {r_} {b_:4d} {deg[rr]:5d} {lab_h[b_]} wg {blk[rr]:3d} | {st[rr,0]-t0:6.1f} " + " ".join(f"{x:6.2f}" for x in ph) + f" | {st[rr,6]-t0:6.2f}")

# ---- dense_step phases (the training launch: partial sums + in-kernel classifier Adam)
tiles = (B + 15) // 16
dst = torch.zeros(tiles, 16, dtype=torch.int64, device="cuda")
for it in range(3):
    torch.cuda.synchronize()
    if it == 2: lib.pcg_debug_set_dense_stamps(C.c_void_p(dst.data_ptr()))
    fz._enqueue_tail(ids, lab, B, agg, plan, True)
    torch.cuda.synchronize()
lib.pcg_debug_set_dense_stamps(None)
fz.flush()
d = dst.cpu().numpy().astype(np.float64) * 0.01
t0d = d[:, 0].min()
names = {0: "start", 1: "staged", 2: "h_r", 11: "comb mfma", 3: "comb reduce", 4: "loss grads", 8: "dcomb", 9: "dW_cls", 10: "dW_clf", 5: "bias+sync", 6: "dh_r+dW_inter", 7: "dW_r"}
order = [0, 1, 2, 11, 3, 4, 8, 9, 10, 5, 6, 7]
print("dense_step: tiles %d, start skew %.2f us, last end %.2f us after the first start" % (tiles, d[:, 0].max() - t0d, d[:, 7].max() - t0d))
prev = d[:, 0]
for s_ in order[1:]:
    cur = d[:, s_]
    print(f"   {names[s_]:16s} mean {np.mean(cur - prev):6.2f} us   max {np.max(cur - prev):6.2f}")
    prev = cur
print("   total per tile   mean %.2f us  max %.2f" % (np.mean(d[:, 7] - d[:, 0]), np.max(d[:, 7] - d[:, 0])))
la = d[:, 14].max()
if la > 0:
    print("   the last arriver (tile %d): entered the classifier's Adam %.2f us after the first tile's start (last tile done at %.2f), finished at %.2f" % (
        int(d[:, 14].argmax()), la - t0d, d[:, 7].max() - t0d, d[:, 15].max() - t0d))
