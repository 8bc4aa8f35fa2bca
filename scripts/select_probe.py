"""Diagnostic: per-row phase times of select_rows (in-kernel stamps, 10-ns ticks) on one batch.
PROBE_WORKLOAD=yelp|powerlaw  PROBE_B=...  PROBE_NODES / PROBE_EDGES (powerlaw)"""
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
ids = tr.sampler.pick(B, 0); lab = tr.labels_i32[ids.long()]
rows = g.R * B
stamps = torch.zeros(rows + 1, 8, dtype=torch.int64, device="cuda")
for it in range(3):
    keys = fz._enqueue_front(ids, lab, B, True)
    torch.cuda.synchronize()
    if it == 2: lib.pcg_debug_set_stamps(C.c_void_p(stamps.data_ptr()))
    fz._enqueue_choose(ids, lab, B, keys, True, planned=True, combine=False)
    torch.cuda.synchronize()
lib.pcg_debug_set_stamps(None)
raw = stamps[:rows].cpu().numpy()
blk = (raw[:, 0] >> 54) & 0x3FF
raw[:, 0] &= (1 << 54) - 1
st = raw.astype(np.float64) * 0.01
deg = np.stack([g.deg_host[r][ids.cpu().numpy()] for r in range(g.R)]).reshape(-1)
lab_h = np.tile(lab.cpu().numpy(), g.R)
have = st[:, 0] > 0                     # (rows of <= 16 neighbours do not stamp their start)
t0 = st[have, 0].min()
end = np.where(st[:, 6] > 0, st[:, 6], 0)
print(f"rows {rows}  stamped {have.sum()}  kernel span (first start -> last end) {end.max() - t0:.1f} us")
names = ["keys(1)", "kth(2)", "compact(3)", "min-search(4)", "min-resolve(5)", "tail(6)"]
for lo, hi, tier in ((16, 64, "lane rows 17..64"), (64, 512, "wave rows 65..512"), (512, 4096, "wg rows 513..4096"), (4096, 1 << 30, "wg rows > 4096")):
    sel = have & (deg > lo) & (deg <= hi)
    if not sel.any(): continue
    for posflag, tag in ((0, "neg"), (1, "pos")):
        s2 = sel & (lab_h == posflag)
        if not s2.any(): continue
        s = st[s2]
        tot = s[:, 6] - s[:, 0]
        line = f"{tier:18s} {tag}: rows {s2.sum():5d} deg mean {deg[s2].mean():6.0f} max {deg[s2].max():5d} | start {s[:,0].min()-t0:5.1f}..{s[:,0].max()-t0:5.1f} end max {s[:,6].max()-t0:5.1f} | total mean {tot.mean():5.2f} max {tot.max():5.2f} |"
        prev = s[:, 0]
        for i, n in enumerate(names, start=1):
            cur = np.where(s[:, i] > 0, s[:, i], prev)
            line += f" {n} {np.mean(cur - prev):4.2f}"
            prev = cur
        print(line)
short = (deg <= 16)
se = st[short & (st[:, 6] > 0), 6]
if se.size:
    print(f"rows <= 16 with a tail: {se.size}, last end {se.max() - t0:.1f} us; rows <= 16: {short.sum()}")
order = np.argsort(-end)[:12]
print("last rows to finish: row deg label start end total")
for r_ in order:
    print(f"   {r_:6d} deg {deg[r_]:5d} lab {lab_h[r_]} wg {blk[r_]:4d} start {st[r_,0]-t0:6.1f} end {end[r_]-t0:6.1f} total {end[r_]-st[r_,0]:6.2f}  phases " +
          " ".join(f"{st[r_,i]-st[r_,i-1]:5.2f}" if st[r_, i] > 0 and st[r_, i-1] > 0 else "  -  " for i in range(1, 7)))
