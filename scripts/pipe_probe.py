"""Diagnostic: what does running the step's stages on SEPARATE streams buy?  (timing only)

A step's selection (scores -> select_rows) needs nothing of the dense tail - the label classifier is stepped on its own - and
the gather / dense tail of batch t need nothing of batch t + 1's selection.  So the chain select -> gather -> dense of a step can
be a software pipeline over batches: the selection of batch t + 1 beside the gather + dense of batch t.  This probe replays the
same kernels (the four-launch step's: pcg_step_scores | pcg_choose_select_planned | pcg_gather_lists_planned | pcg_train_dense)
 - V0: on one stream, one hipGraph of E epochs;
 - V1: on two streams inside one hipGraph (select chain | gather + dense chain), double-buffered lists / counts / aggregates;
 - V2: on three streams (select | gather | dense);
 - V3 / V4: V1 / V2 launched by the host without a graph.
and prints microseconds per step.  PROBE_WORKLOAD=yelp|powerlaw, PROBE_B, PROBE_E (emb), PROBE_EPOCHS (epochs per graph)."""
import sys, os, time, ctypes as C, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcgnn_amd
from pcgnn_amd import synth, ops, _lib
from pcgnn_amd.handler import PCGNNTrainer
_p = ops._p
B = int(os.environ.get("PROBE_B", "1024"))
E = int(os.environ.get("PROBE_E", "64"))
EPG = int(os.environ.get("PROBE_EPOCHS", "4"))
if os.environ.get("PROBE_WORKLOAD", "yelp") == "powerlaw":
    w = synth.power_law(int(os.environ.get("PROBE_NODES", "2000000")), int(os.environ.get("PROBE_EDGES", "40000000")), 0)
else:
    w = synth.yelp_like(0)
dev = torch.device("cuda", 0)
tr = PCGNNTrainer(w, dict(engine="fused", batch_size=B, emb_size=E), dev)
fz = tr.fused; g = fz.g; lib = _lib.load()
nb = tr.batches_per_epoch()
tr.start_epoch_staged()                       # ids / labels / plans of one epoch (set 0)
torch.cuda.synchronize()
n = tr.pick_size
ids_all, lab_all = fz._ep_ids[:n], fz._ep_lab[:n]
# double buffers: data part (list | partial sums), counts, aggregates
data = [fz.data, torch.zeros_like(fz.data)]
cnt = [torch.empty_like(fz.cnt), torch.empty_like(fz.cnt)]
agg = [torch.empty_like(fz.agg), torch.empty_like(fz.agg)]
b1, b2 = fz.betas


def st(s):
    return C.c_void_p(s.cuda_stream)


def front(s):
    _lib.check(lib.pcg_step_scores(g.desc_ref(), _p(fz.clf_next), C.c_void_p(fz.clf_next.data_ptr() + 8 * fz.F), 0, g.n_nodes, _p(fz.s0),
                                   None, _p(fz.keys), -1, _p(fz.sync), None, st(s)), "scores")


def batch(t):
    b = t % nb
    lo = b * B
    Bt = min(B, n - lo)
    return ids_all[lo:lo + Bt], lab_all[lo:lo + Bt], Bt, fz._ep_plan(b)


def select(t, s):
    ids, lab, Bt, plan = batch(t)
    _lib.check(lib.pcg_choose_select_planned(g.desc_ref(), _p(ids), _p(lab), Bt, _p(fz.s0), None, _p(fz.keys), fz._thr, fz._rhos, 1, 0,
                                             _p(cnt[t & 1]), _p(data[t & 1]), C.c_void_p(plan), fz.list_capacity, _p(fz.status), _p(fz.sync), 0,
                                             st(s)), "select")


def gather(t, s):
    ids, lab, Bt, plan = batch(t)
    a = agg[t & 1].view(-1)[:g.R * Bt * g.feat_dim].view(g.R, Bt, g.feat_dim)
    _lib.check(lib.pcg_gather_lists_planned(_p(g.X), g.feat_dim, g.X.stride(0), g.n_nodes, g.R * Bt, _p(cnt[t & 1]), g.desc_ref(), Bt,
                                            _p(data[t & 1]), C.c_void_p(plan), fz.list_capacity, _p(a), a.stride(-2), _p(fz.status), st(s)), "gather")


def dense(t, s):
    ids, lab, Bt, plan = batch(t)
    a = agg[t & 1].view(-1)[:g.R * Bt * g.feat_dim].view(g.R, Bt, g.feat_dim)
    _lib.check(lib.pcg_train_dense(g.desc_ref(), _p(fz.theta), _p(fz.m), _p(fz.v), fz.E, _p(ids), _p(lab), Bt, _p(a), a.stride(1),
                                   _p(cnt[t & 1]), _p(data[t & 1]), C.c_void_p(plan), fz.list_capacity, fz.lambda_1, 1.0 / Bt, _p(fz.logits),
                                   _p(fz.center), None, _p(fz.row_loss), _p(fz.slabs), _p(fz.step_counter), _p(fz.sync), fz.lr, b1, b2,
                                   fz.eps, fz.wd, 2, None, 0, None, st(s)), "dense")


def seq(T, main):
    for t in range(T):
        front(main); select(t, main); gather(t, main); dense(t, main)


def pipe2(T, A, Bs):
    """A: front + select of every batch; Bs: gather + dense.  gather(t) waits for select(t); select(t + 2) waits for dense(t)."""
    ev_sel, ev_den = {}, {}
    for t in range(T):
        if t >= 2:
            A.wait_event(ev_den[t - 2])
        front(A); select(t, A)
        ev_sel[t] = torch.cuda.Event(); ev_sel[t].record(A)
        Bs.wait_event(ev_sel[t])
        gather(t, Bs); dense(t, Bs)
        ev_den[t] = torch.cuda.Event(); ev_den[t].record(Bs)


def pipe3(T, A, G, D):
    ev_sel, ev_gat, ev_den = {}, {}, {}
    for t in range(T):
        if t >= 2:
            A.wait_event(ev_den[t - 2])          # counts / list buffer t & 1 are free
        front(A); select(t, A)
        ev_sel[t] = torch.cuda.Event(); ev_sel[t].record(A)
        G.wait_event(ev_sel[t])
        if t >= 2:
            G.wait_event(ev_den[t - 2])          # aggregates / partial sums t & 1 are free
        gather(t, G)
        ev_gat[t] = torch.cuda.Event(); ev_gat[t].record(G)
        D.wait_event(ev_gat[t])
        dense(t, D)
        ev_den[t] = torch.cuda.Event(); ev_den[t].record(D)


def timeit(fn, reps, steps_per_call, label):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label:70s} {dt / (reps * steps_per_call) * 1e6:8.2f} us/step  ({reps} x {steps_per_call} steps)", flush=True)


T = EPG * nb
main = torch.cuda.current_stream(dev)
side1, side2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
# warm every kernel outside a capture
seq(nb, main)
torch.cuda.synchronize()

def host2():
    side1.wait_stream(main)
    pipe2(T, main, side1)
    main.wait_stream(side1)


def host3():
    side1.wait_stream(main); side2.wait_stream(main)
    pipe3(T, main, side1, side2)
    main.wait_stream(side1); main.wait_stream(side2)


# ---- D: the theta-independent half (scores, select, gather) of a whole group of batches on SIDE streams, with no event
#         inside the group, beside the theta-dependent half (dense + Adam) on the main stream ------------------------------------
nside = int(os.environ.get("PROBE_SIDE", "4"))
sides = [torch.cuda.Stream(dev) for _ in range(nside)]
datas = [torch.zeros_like(fz.data) for _ in range(nside)] if fz.data.numel() < (2 << 30) else None
cnts = [torch.empty_like(fz.cnt) for _ in range(nside)]
aggs = [torch.empty_like(fz.agg) for _ in range(nside)]
s0s = [torch.empty_like(fz.s0) for _ in range(nside)]
keyss = [torch.zeros_like(fz.keys) for _ in range(nside)]
syncs = [torch.zeros_like(fz.sync) for _ in range(nside)]


def front_k(k, s):
    _lib.check(lib.pcg_step_scores(g.desc_ref(), _p(fz.clf_next), C.c_void_p(fz.clf_next.data_ptr() + 8 * fz.F), 0, g.n_nodes, _p(s0s[k]),
                                   None, _p(keyss[k]), -1, _p(syncs[k]), None, st(s)), "scores")


def select_k(t, k, s):
    ids, lab, Bt, plan = batch(t)
    _lib.check(lib.pcg_choose_select_planned(g.desc_ref(), _p(ids), _p(lab), Bt, _p(s0s[k]), None, _p(keyss[k]), fz._thr, fz._rhos, 1, 0,
                                             _p(cnts[k]), _p(datas[k]), C.c_void_p(plan), fz.list_capacity, _p(fz.status), _p(syncs[k]), 0,
                                             st(s)), "select")


def gather_k(t, k, s):
    ids, lab, Bt, plan = batch(t)
    a = aggs[k].view(-1)[:g.R * Bt * g.feat_dim].view(g.R, Bt, g.feat_dim)
    _lib.check(lib.pcg_gather_lists_planned(_p(g.X), g.feat_dim, g.X.stride(0), g.n_nodes, g.R * Bt, _p(cnts[k]), g.desc_ref(), Bt,
                                            _p(datas[k]), C.c_void_p(plan), fz.list_capacity, _p(a), a.stride(-2), _p(fz.status), st(s)), "gather")


def adam_only(s):
    _lib.check(lib.pcg_adam_step(_p(fz.theta), _p(fz.m), _p(fz.v), _p(fz.slabs), lib.pcg_dense_n_tiles(B), fz.n_params,
                                 _p(fz.step_counter), fz.lr, b1, b2, fz.eps, fz.wd, None, 1, st(s)), "adam")


def phase1(T, streams):
    """scores | select | gather of T batches, batch t on stream t % len(streams): independent chains"""
    for t in range(T):
        k = t % len(streams)
        front_k(k, streams[k]); select_k(t, k, streams[k]); gather_k(t, k, streams[k])


def phase2(T, s):
    for t in range(T):
        dense(t, s); adam_only(s)


def decoupled(T, main_s, side_list, overlap):
    if overlap:                                  # phase 1 (of the "next" group) beside phase 2 (of this one)
        for s in side_list:
            s.wait_stream(main_s)
        phase1(T, side_list)
        phase2(T, main_s)
        for s in side_list:
            main_s.wait_stream(s)
    else:
        phase1(T, [main_s]); phase2(T, main_s)


if datas is not None:
    phase1(nb, [main]); phase2(nb, main)
    torch.cuda.synchronize()
    fz.status.zero_()
    timeit(lambda: decoupled(T, main, sides, False), 10, T, "D3h phase 1 then phase 2, one stream, host launches")
    timeit(lambda: decoupled(T, main, sides, True), 10, T, "D4h phase 1 on %d side streams beside phase 2, host launches" % nside)
    timeit(lambda: phase2(T, main), 10, T, "D2h phase 2 alone, host launches")
    for label, fn in (("D0 phase 1 alone (scores|select|gather x T), one stream", lambda m_, sd: phase1(T, [m_])),
                      ("D1 phase 1 alone on %d streams" % nside, lambda m_, sd: (_fork(m_, sd), phase1(T, sd), _join(m_, sd))),
                      ("D2 phase 2 alone (dense|adam x T)", lambda m_, sd: phase2(T, m_)),
                      ("D3 phase 1 then phase 2, one stream", lambda m_, sd: decoupled(T, m_, sd, False)),
                      ("D4 phase 1 on %d side streams BESIDE phase 2 on the main stream" % nside, lambda m_, sd: decoupled(T, m_, sd, True))):
        def _fork(m_, sd):
            for s_ in sd:
                s_.wait_stream(m_)
        def _join(m_, sd):
            for s_ in sd:
                m_.wait_stream(s_)
        gq = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gq):
            cs = torch.cuda.current_stream(dev)
            sd = [torch.cuda.Stream(dev) for _ in range(nside)]
            fn(cs, sd)
        timeit(gq.replay, 30, T, label + " (graph)")
    print("status:", int(fz.status.item()), flush=True)
    fz.status.zero_()
if os.environ.get("PROBE_STREAMS") != "1":
    print("done")
    sys.exit(0)
timeit(lambda: seq(T, main), 10, T, "V0h one stream, host launches (python + ctypes)")
timeit(host2, 10, T, "V3 two streams, host launches")
timeit(host3, 10, T, "V4 three streams, host launches")
gr0 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr0):
    seq(T, torch.cuda.current_stream(dev))
timeit(gr0.replay, 40, T, f"V0 one stream, one graph of {T} steps (scores|select|gather|dense)")

gr1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr1):
    cs = torch.cuda.current_stream(dev)
    s1 = torch.cuda.Stream(dev)
    s1.wait_stream(cs)
    pipe2(T, cs, s1)
    cs.wait_stream(s1)
timeit(gr1.replay, 40, T, f"V1 two streams inside one graph of {T} steps")

if os.environ.get('PROBE_V2') == '1':
  gr2 = torch.cuda.CUDAGraph()
  with torch.cuda.graph(gr2):
    cs = torch.cuda.current_stream(dev)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    s1.wait_stream(cs); s2.wait_stream(cs)
    pipe3(T, cs, s1, s2)
    cs.wait_stream(s1); cs.wait_stream(s2)
  timeit(gr2.replay, 40, T, f"V2 three streams inside one graph of {T} steps")


# graphs of a single epoch (how much of a forked graph's cost is per replay?)
if EPG > 1:
    gr1b = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr1b):
        cs = torch.cuda.current_stream(dev)
        s1 = torch.cuda.Stream(dev)
        s1.wait_stream(cs)
        pipe2(nb, cs, s1)
        cs.wait_stream(s1)
    timeit(gr1b.replay, 160, nb, f"V1b two streams inside one graph of {nb} steps")
    gr0b = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr0b):
        seq(nb, torch.cuda.current_stream(dev))
    timeit(gr0b.replay, 160, nb, f"V0b one stream, one graph of {nb} steps")
fz.status.zero_()
print("done")
