"""Diagnostic: per-phase time of the select kernels on one YelpChi-like batch (in-kernel stamps)."""
import sys, os, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcgnn_amd
from pcgnn_amd import synth, ops, _lib
from pcgnn_amd.handler import PCGNNTrainer
B = int(os.environ.get("PROBE_B", "1024"))
if os.environ.get("PROBE_WORKLOAD", "yelp") == "powerlaw":
    w = synth.power_law(int(os.environ.get("PROBE_NODES", "2000000")), int(os.environ.get("PROBE_EDGES", "40000000")), 0)
else:
    w = synth.yelp_like(0)
tr = PCGNNTrainer(w, dict(engine="fused", batch_size=B), torch.device("cuda", 0))
fz = tr.fused; g = fz.g; lib = _lib.load()
ids_all = tr.start_epoch(0)
ids = ids_all[:B].contiguous(); lab = tr.labels_i32[ids.long()]
rows = g.R * B
stamps = torch.zeros(rows + 1, 8, dtype=torch.int64, device="cuda")
for it in range(3):
    keys = fz._enqueue_scores(True)
    if it == 2: lib.pcg_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    fz._enqueue_choose(ids, lab, B, keys, True)
    torch.cuda.synchronize()
lib.pcg_debug_set_stamps(None)
plan = stamps[rows].cpu().numpy().astype(np.float64) * 0.01
print('plan phases (us): load+rowplan %.2f, scans %.2f, writes %.2f; plan start -> first select start %.2f' % (plan[1]-plan[0], plan[2]-plan[1], plan[3]-plan[2], (stamps[:rows,0].cpu().numpy() & ((1 << 54) - 1)).min()*0.01 - plan[0]))
raw = stamps[:rows].cpu().numpy()
blk = (raw[:, 0] >> 54) & 0x3FF                                   # workgroup that ran the row
raw[:, 0] &= (1 << 54) - 1
st = raw.astype(np.float64) * 0.01   # us
deg = np.stack([g.deg_host[r][ids.cpu().numpy()] for r in range(g.R)]).reshape(-1)
t0 = st[:, 0].min()
names = ["rec+keys(1)", "kth(2)", "compact+list(3)", "min-search(4)", "min-resolve(5)", "tail(6)"]
for lo, hi, tier in ((0, 512, "T1"), (512, 4096, "T4"), (4096, 1 << 30, "T16")):
    sel = (deg > lo) & (deg <= hi)
    if not sel.any(): continue
    s = st[sel]
    print(f"{tier}: rows {sel.sum()}, deg mean {deg[sel].mean():.0f} max {deg[sel].max()}, start skew {s[:,0].min()-t0:.1f}..{s[:,0].max()-t0:.1f} us, end {s[:,6].max()-t0:.1f} us")
    prev = s[:, 0]
    for i, n in enumerate(names, start=1):
        cur = np.where(s[:, i] > 0, s[:, i], prev)
        dt = cur - prev
        print(f"   {n:18s} mean {dt.mean():7.2f} us  p95 {np.percentile(dt,95):7.2f}  max {dt.max():7.2f}")
        prev = cur
    tot = s[:, 6] - s[:, 0]
    print(f"   total per row      mean {tot.mean():7.2f} us  max {tot.max():7.2f}")

end = st[:, 6] - t0
print("row end times (us): p50 %.1f p90 %.1f p99 %.1f max %.1f; rows still running after 50%% of the kernel: %d" % (
    np.percentile(end, 50), np.percentile(end, 90), np.percentile(end, 99), end.max(), int((end > 0.5 * end.max()).sum())))
bend = np.zeros(1024); bstart = np.full(1024, 1e18); brows = np.zeros(1024, dtype=int); bwide = np.zeros(1024, dtype=int)
for r_ in range(rows):
    b_ = int(blk[r_]); bend[b_] = max(bend[b_], end[r_]); bstart[b_] = min(bstart[b_], st[r_, 0] - t0); brows[b_] += 1; bwide[b_] += int(deg[r_] > 512)
used_b = np.flatnonzero(brows)
print("workgroups used %d; end time p50 %.1f p90 %.1f max %.1f; first-row start max %.1f" % (len(used_b), np.percentile(bend[used_b], 50), np.percentile(bend[used_b], 90), bend[used_b].max(), bstart[used_b].max()))
late = used_b[np.argsort(-bend[used_b])][:8]
for b_ in late:
    sel_ = blk == b_
    print(f"   wg {b_}: rows {brows[b_]} (wide {bwide[b_]}), first start {bstart[b_]:.1f}, end {bend[b_]:.1f}, sum of row times {(st[sel_,6]-st[sel_,0]).sum():.1f}, wide row time {(st[sel_ & (deg>512),6]-st[sel_ & (deg>512),0]).sum():.1f}")
busy = (st[:, 6] - st[:, 0])
print("sum of row times %.0f us -> %.1f us if spread evenly over 768 x 8 wave slots (wide rows use 8 slots each)" % (
    busy.sum(), (busy * np.where(deg > 512, 8, 1)).sum() / (768 * 8)))
rounds = (stamps[:rows].cpu().numpy()[:, 7] & 0xFFFFFFFF).astype(np.int64)
ncand = (stamps[:rows].cpu().numpy()[:, 7] >> 32).astype(np.int64)
kth = st[:, 2] - st[:, 1]
print("kth phase by degree bucket: rows, mean us, max us, mean rounds, max rounds, mean ncand at exit")
for lo, hi in ((0, 3), (3, 64), (64, 128), (128, 256), (256, 512), (512, 1024), (1024, 4096), (4096, 1 << 30)):
    sel = (deg > lo) & (deg <= hi)
    if sel.any():
        print(f"  deg ({lo},{hi}]: {sel.sum():5d} rows  kth {kth[sel].mean():6.2f} / {kth[sel].max():6.2f} us   rounds {rounds[sel].mean():5.1f} / {rounds[sel].max():3d}   ncand {ncand[sel].mean():6.1f}")

# ---- dense_step phases
tiles = (B + 15) // 16
dst = torch.zeros(tiles, 16, dtype=torch.int64, device="cuda")
agg, _ = fz._enqueue_sample(ids, lab, B, True)
for it in range(3):
    if it == 2: lib.pcg_debug_set_dense_stamps(C.c_void_p(dst.data_ptr()))
    fz._enqueue_dense(ids, lab, B, agg, True)
    torch.cuda.synchronize()
lib.pcg_debug_set_dense_stamps(None)
d = dst.cpu().numpy().astype(np.float64) * 0.01
names = ["stage weights+self+agg", "h_r fwd", "combined", "logits + loss grads", "dcomb + small dW", "dh_r + dW_inter", "dW_r", ""]
print("dense_step per tile (us):", "start skew %.1f" % (d[:,0].max()-d[:,0].min()), " total mean %.1f max %.1f" % ((d[:,7]-d[:,0]).mean(), (d[:,7]-d[:,0]).max()))
for i in range(1, 8):
    dt = d[:, i] - d[:, i-1]
    print(f"   {names[i-1]:24s} mean {dt.mean():6.2f}  max {dt.max():6.2f}")
print("   inside phase 5 (thread 0): dcomb loop %.2f, dW_cls loop %.2f, dW_clf loop %.2f, rest+barrier %.2f" % (
    (d[:, 8] - d[:, 4]).mean(), (d[:, 9] - d[:, 8]).mean(), (d[:, 10] - d[:, 9]).mean(), (d[:, 5] - d[:, 10]).mean()))

# ---- the slowest T1 rows
tot = st[:, 6] - st[:, 0]
t1 = np.flatnonzero(deg <= 512)
order = t1[np.argsort(-tot[t1])][:14]
lab_h = lab.cpu().numpy()
print("slowest T1 rows: row rel b deg label | start(us) keys kth compact search resolve tail | total")
for rr in order:
    r_, b_ = rr // B, rr % B
    ph = [st[rr, i] - st[rr, i - 1] if st[rr, i] > 0 else 0.0 for i in range(1, 7)]
    print(f"  {rr:5d} {r_} {b_:4d} {deg[rr]:4d} {lab_h[b_]} | {st[rr,0]-t0:6.1f} " + " ".join(f"{x:6.2f}" for x in ph) + f" | {tot[rr]:6.2f}")

