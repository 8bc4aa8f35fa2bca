#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container, where the reference tree is mounted
read-only at /root/reference:

    python tests/golden/make_golden.py

It imports the reference's hot-path modules (``src/layers.py``, ``src/model.py``,
``src/utils.py``, ``src/graphsage.py``) with ``cuda=False``, drives them on small
seeded synthetic graphs and stores inputs + outputs as ``.npz`` (data only - no
reference source travels).  The reference ships no tests or golden vectors of
its own, so these files are what pins ``oracle/`` - and
through it the HIP path - to the reference's behaviour.

Nothing here is imported by the product or by the test-suite; the tests read
only the ``.npz`` files.
"""
import math
import os
import random
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

import src.layers as RL  # noqa: E402
from src.graphsage import GCNAggregator, GCNEncoder, Encoder, MeanAggregator  # noqa: E402
from src.model import PCALayer  # noqa: E402
from src.utils import pick_step  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = sys.argv[1] if len(sys.argv) > 1 else HERE       # `python make_golden.py /tmp/check` regenerates elsewhere (to compare)
torch.set_num_threads(1)


# ---------------------------------------------------------------------------
# synthetic multi-relation graph in the reference's input format
# (dict[int -> set[int]], symmetric, self-loops: src/utils.py:226-239)
# ---------------------------------------------------------------------------
def synth_graph(seed, n, feat_dim, rel_avg_deg, pos_rate, nonneg=False):
    rs = np.random.RandomState(seed)
    X = rs.randn(n, feat_dim).astype(np.float32)
    if nonneg:  # amazon-like: non-negative, row-normalised (utils.py:213-223)
        X = np.abs(X)
        X = (X / (X.sum(1, keepdims=True) + 0.01)).astype(np.float32)
    labels = (rs.rand(n) < pos_rate).astype(np.int64)
    # node "popularity" gives a skewed degree distribution + one hub (node 7)
    pop = rs.pareto(1.5, n) + 0.05
    pop[7] = pop.max() * 6
    pop /= pop.sum()
    rels = []
    for avg in rel_avg_deg:
        adj = {v: {v} for v in range(n)}
        m = int(n * avg / 2)
        src = rs.choice(n, size=m, p=pop)
        dst = rs.randint(0, n, size=m)
        for a, b in zip(src.tolist(), dst.tolist()):
            adj[a].add(b)
            adj[b].add(a)
        rels.append(adj)
    # a few rows with exactly the small degrees that hit the keep-all rule
    # (deg <= num_sample + 1, layers.py:662): isolate some nodes in relation 0
    for v, want in ((11, 1), (12, 2), (13, 3), (14, 4), (15, 5)):
        adj = rels[0]
        for u in list(adj[v]):
            if u != v:
                adj[v].discard(u)
                adj[u].discard(v)
        extra = [w for w in range(20, 20 + want - 1)]
        for u in extra:
            adj[v].add(u)
            adj[u].add(v)
    homo = {v: set().union(*[r[v] for r in rels]) for v in range(n)}
    return X, labels, rels, homo


def to_csr(adj, n):
    indptr = np.zeros(n + 1, dtype=np.int64)
    for v in range(n):
        indptr[v + 1] = indptr[v] + len(adj[v])
    idx = np.empty(indptr[-1], dtype=np.int32)
    for v in range(n):
        idx[indptr[v]:indptr[v + 1]] = sorted(adj[v])
    return indptr, idx


def sets_to_csr(sets):
    off = np.zeros(len(sets) + 1, dtype=np.int64)
    for i, s in enumerate(sets):
        off[i + 1] = off[i] + len(s)
    flat = np.empty(off[-1], dtype=np.int32)
    for i, s in enumerate(sets):
        flat[off[i]:off[i + 1]] = sorted(s)
    return off, flat


# ---------------------------------------------------------------------------
# recorders: wrap the reference's own functions, change nothing
# ---------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.sets = []      # one list[set] per IntraAgg call
        self.feats = []     # to_feats per IntraAgg call
        self.min_gap = math.inf   # smallest distance gap at any selection cut

    def install(self):
        self._train, self._test, self._fwd = RL.choose_step_neighs, RL.choose_step_test, RL.IntraAgg.forward
        rec = self

        def train_wrap(center_scores, center_labels, neigh_scores, neighs_list, minor_scores, minor_list,
                       sample_list, sample_rate):
            rec._gaps(center_scores, neigh_scores, sample_list, center_labels, minor_scores, sample_rate)
            out = rec._train(center_scores, center_labels, neigh_scores, neighs_list, minor_scores, minor_list,
                             sample_list, sample_rate)
            rec.sets.append([set(s) for s in out[0]])
            return out

        def test_wrap(center_scores, neigh_scores, neighs_list, sample_list):
            rec._gaps(center_scores, neigh_scores, sample_list, None, None, None)
            out = rec._test(center_scores, neigh_scores, neighs_list, sample_list)
            rec.sets.append([set(s) for s in out[0]])
            return out

        def fwd_wrap(self_, *a, **k):
            out = rec._fwd(self_, *a, **k)
            rec.feats.append(out[0].detach().numpy().copy())
            return out

        RL.choose_step_neighs, RL.choose_step_test, RL.IntraAgg.forward = train_wrap, test_wrap, fwd_wrap

    def remove(self):
        RL.choose_step_neighs, RL.choose_step_test, RL.IntraAgg.forward = self._train, self._test, self._fwd

    def _gaps(self, center_scores, neigh_scores, sample_list, labels, minor_scores, rate):
        """Track the smallest gap between the last kept and first dropped
        distance, so the fixture is known to be tie-free at every cut."""
        for b in range(len(neigh_scores)):
            c = center_scores[b][0]
            d = torch.abs(c - neigh_scores[b][:, 0]).sort().values
            k = sample_list[b]
            if len(d) > k + 1:
                self.min_gap = min(self.min_gap, float(d[k] - d[k - 1]))
            if labels is not None and int(labels[b]) == 1:
                m = int(k * rate)
                dm = torch.abs(c - minor_scores[:, 0]).sort().values
                if 0 < m < len(dm):
                    self.min_gap = min(self.min_gap, float(dm[m] - dm[m - 1]))


def build_model(X, rels, train_pos, emb, rho, alpha, seed):
    torch.manual_seed(seed)
    n, f = X.shape
    features = torch.nn.Embedding(n, f)
    features.weight = torch.nn.Parameter(torch.FloatTensor(X), requires_grad=False)
    intras = [RL.IntraAgg(features, f, emb, train_pos, rho, cuda=False) for _ in rels]
    if len(rels) == 3:
        inter = RL.InterAgg3(features, f, emb, train_pos, rels, intras, cuda=False)
    elif len(rels) == 1:
        inter = RL.InterAgg1(features, f, emb, train_pos, rels, intras, cuda=False)
    elif len(rels) == 5:
        inter = RL.InterAgg5(features, f, emb, train_pos, rels, intras, cuda=False)
    else:
        raise ValueError
    return PCALayer(2, inter, alpha)


PARAM_KEYS = lambda R: (["weight", "inter1.weight", "inter1.label_clf.weight", "inter1.label_clf.bias"]
                        + [f"inter1.intra_agg{r + 1}.weight" for r in range(R)])


def pcgnn_case(name, seed, n, f, rel_deg, pos_rate, emb, batch, rhos, nonneg=False, lr=0.01, wd=0.001, alpha=2.0):
    X, labels, rels, homo = synth_graph(seed, n, f, rel_deg, pos_rate, nonneg)
    R = len(rels)
    rs = np.random.RandomState(seed + 1)
    idx_train = sorted(rs.choice(n, size=int(0.4 * n), replace=False).tolist())
    y_train = labels[np.array(idx_train)]
    train_pos = [v for v in idx_train if labels[v] == 1]

    # P1 pick: the reference's pick_step under a seeded `random`
    random.seed(seed)
    picked = pick_step(idx_train, y_train, homo, size=2 * len(train_pos))
    random.seed(seed)
    uniforms = np.array([random.random() for _ in range(2 * len(train_pos))], dtype=np.float64)

    # batch: picked nodes (duplicates included) + the hub + the tiny-degree rows
    nodes = (picked[:batch - 8] + [7, 11, 12, 13, 14, 15] + picked[:2])[:batch]
    blab = labels[np.array(nodes)]

    out = {
        "n": n, "f": f, "R": R, "emb": emb, "alpha": alpha, "lr": lr, "wd": wd, "rhos": np.array(rhos),
        "X": X, "labels": labels, "idx_train": np.array(idx_train), "train_pos": np.array(train_pos),
        "pick_uniforms": uniforms, "pick_out": np.array(picked),
        "nodes": np.array(nodes), "batch_labels": blab,
    }
    for r, adj in enumerate(rels):
        out[f"indptr{r}"], out[f"indices{r}"] = to_csr(adj, n)
    out["homo_indptr"], out["homo_indices"] = to_csr(homo, n)

    min_gap = math.inf
    for rho in rhos:
        tag = f"rho{rho}"
        model = build_model(X, rels, train_pos, emb, rho, alpha, seed)
        sd = model.state_dict()
        if "w_weight" not in out:
            for k in PARAM_KEYS(R):
                out["w_" + k] = sd[k].numpy().copy()
            # full-table label-aware scores from the reference's own label_clf
            out["table_scores"] = model.inter1.label_clf(model.inter1.features.weight).detach().numpy()

        for mode, flag in (("train", True), ("test", False)):
            if mode == "test" and rho != rhos[0]:
                continue  # test mode does not depend on rho
            rec = Recorder()
            rec.install()
            try:
                logits, cscores = model.forward(nodes, torch.LongTensor(blab), flag)
            finally:
                rec.remove()
            min_gap = min(min_gap, rec.min_gap)
            key = f"{tag}_{mode}" if mode == "train" else "test"
            out[f"{key}_logits"] = logits.detach().numpy()
            out[f"{key}_center_scores"] = cscores.detach().numpy()
            for r in range(R):
                out[f"{key}_sel_off{r}"], out[f"{key}_sel_idx{r}"] = sets_to_csr(rec.sets[r])
                if rho == rhos[0] or mode == "test":
                    out[f"{key}_feats{r}"] = rec.feats[r]
            combined, _ = model.inter1(nodes, torch.LongTensor(blab), flag)
            if rho == rhos[0] or mode == "test":
                out[f"{key}_combined"] = combined.detach().numpy()
            gp, lp = model.to_prob(nodes, torch.LongTensor(blab), flag)
            out[f"{key}_gnn_prob"] = gp.detach().numpy()

        # loss / backward / one Adam step (model_handler.py:124,149-153)
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()), lr=lr, weight_decay=wd)
        opt.zero_grad()
        loss = model.loss(nodes, torch.LongTensor(blab))
        loss.backward()
        out[f"{tag}_loss"] = np.float32(loss.item())
        if rho == rhos[0]:
            named = dict(model.named_parameters())
            for k in PARAM_KEYS(R):
                out[f"{tag}_grad_" + k] = named[k].grad.numpy().copy()
            opt.step()
            sd2 = model.state_dict()
            for k in PARAM_KEYS(R):
                out[f"{tag}_step_" + k] = sd2[k].numpy().copy()

    out["min_cut_gap"] = np.float64(min_gap)
    assert min_gap > 0, f"{name}: a selection cut falls on a tie - pick another seed"

    # S1: GraphSAGE / GCN aggregators + encoders on the homo graph
    features = torch.nn.Embedding(n, f)
    features.weight = torch.nn.Parameter(torch.FloatTensor(X), requires_grad=False)
    sub = nodes[:64]
    neighs = [homo[int(v)] for v in sub]
    out["s1_nodes"] = np.array(sub)
    out["s1_mean"] = MeanAggregator(features, cuda=False).forward(sub, neighs).detach().numpy()
    out["s1_mean_gcn"] = MeanAggregator(features, cuda=False, gcn=True).forward(sub, neighs).detach().numpy()
    out["s1_gcn"] = GCNAggregator(features, cuda=False).forward(sub, neighs).detach().numpy()
    torch.manual_seed(seed + 5)
    enc = Encoder(features, f, emb, homo, MeanAggregator(features, cuda=False), gcn=True, cuda=False)
    out["s1_sage_enc_w"] = enc.weight.detach().numpy().copy()
    out["s1_sage_enc"] = enc(sub).detach().numpy()
    genc = GCNEncoder(features, f, emb, homo, GCNAggregator(features, cuda=False), cuda=False)
    out["s1_gcn_enc_w"] = genc.weight.detach().numpy().copy()
    out["s1_gcn_enc"] = genc(sub).detach().numpy()
    # random fan-out (graphsage.py:70-74): random.sample over each neighbour SET under a seeded `random`.  The sets are
    # built as set(sorted(...)) so that a test can rebuild objects with the same CPython iteration order.
    fan_k, fan_seed = 5, seed + 17
    out["s1_fanout_k"], out["s1_fanout_seed"] = fan_k, fan_seed
    for gcn_flag, key in ((False, "s1_fanout_mean"), (True, "s1_fanout_mean_gcn")):
        random.seed(fan_seed)
        fsets = [set(sorted(homo[int(v)])) for v in sub]
        out[key] = MeanAggregator(features, cuda=False, gcn=gcn_flag).forward(sub, fsets, num_sample=fan_k).detach().numpy()

    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, min gap at any cut = {min_gap:.3e}, "
          f"B={len(nodes)} P={len(train_pos)} pos-in-batch={int(blab.sum())}")


def kat_case():
    """Hand-sized known answers straight from the reference's choose functions
    (SURVEY.md section 8c)."""
    out = {}
    # (1) centre 0.0, five neighbours, threshold .5 -> k=3 -> {11,12,14}
    cs = torch.tensor([[0.0, 0.0]])
    ns = [torch.tensor([[.3, 0], [.1, 0], [.2, 0], [.9, 0], [.05, 0]])]
    sets, _ = RL.choose_step_test(cs, ns, [[10, 11, 12, 13, 14]], [3])
    out["kat1_center"], out["kat1_ids"], out["kat1_s0"], out["kat1_k"] = 0.0, [10, 11, 12, 13, 14], [.3, .1, .2, .9, .05], 3
    out["kat1_out"] = sorted(sets[0])
    # (2) positive centre 1.0, one neighbour, minority over-sampling with several rho
    cs = torch.tensor([[1.0, 0.0]])
    pos = torch.tensor([[1.5, 0], [.9, 0], [1.0, 0], [3.0, 0]])
    out["kat2_center"], out["kat2_ids"], out["kat2_s0"], out["kat2_k"] = 1.0, [20], [0.4], 1
    out["kat2_pos_ids"], out["kat2_pos_s0"] = [20, 31, 32, 33], [1.5, .9, 1.0, 3.0]
    for rho in (0.2, 0.5, 0.8, 2.0, 3.5):
        sets, _ = RL.choose_step_neighs(cs, torch.tensor([1]), [torch.tensor([[0.4, 0.0]])], [[20]], pos,
                                        [20, 31, 32, 33], [1], rho)
        out[f"kat2_out_rho{rho}"] = sorted(sets[0])
    # (3) keep-all table for threshold 0.5: deg -> number kept
    kept = []
    for deg in range(1, 13):
        k = math.ceil(deg * 0.5)
        ns = [torch.stack([torch.linspace(0.1, 1.0, deg), torch.zeros(deg)], 1)]
        sets, _ = RL.choose_step_test(torch.tensor([[0.0, 0.0]]), ns, [list(range(100, 100 + deg))], [k])
        kept.append(len(sets[0]))
    out["kat3_deg"], out["kat3_kept"] = list(range(1, 13)), kept
    np.savez_compressed(os.path.join(OUT, "kat.npz"), **{k: np.asarray(v) for k, v in out.items()})
    print("kat:", {k: (v if not hasattr(v, "shape") else v.tolist()) for k, v in out.items() if "out" in k or "kept" in k})


def split_case():
    """The reference's train / valid / test split (src/model_handler.py:36-48): two stratified
    ``sklearn.model_selection.train_test_split`` calls, run here with scikit-learn itself."""
    from sklearn.model_selection import train_test_split
    out = {}
    rs = np.random.RandomState(4)
    for tag, n, rate, first, train_ratio, seed in (("yelp", 4000, 0.1453, 0, 0.4, 2), ("amazon", 1500, 0.0687, 300, 0.1, 7),
                                                   ("tiny", 90, 0.3, 0, 0.05, 11)):
        labels = (rs.rand(n) < rate).astype(np.int64)
        index = list(range(first, n))
        a = train_test_split(index, labels[first:], stratify=labels[first:], train_size=train_ratio, random_state=seed, shuffle=True)
        b = train_test_split(a[1], a[3], stratify=a[3], test_size=0.67, random_state=seed, shuffle=True)
        out[f"{tag}_labels"], out[f"{tag}_first"], out[f"{tag}_train_ratio"], out[f"{tag}_seed"] = labels, first, train_ratio, seed
        out[f"{tag}_idx_train"], out[f"{tag}_idx_valid"], out[f"{tag}_idx_test"] = np.array(a[0]), np.array(b[0]), np.array(b[1])
        out[f"{tag}_y_train"], out[f"{tag}_y_valid"], out[f"{tag}_y_test"] = a[2], b[2], b[3]
    np.savez_compressed(os.path.join(OUT, "split.npz"), **out)
    print("split:", {k: len(v) for k, v in out.items() if k.endswith("idx_train")})


if __name__ == "__main__":
    kat_case()
    split_case()
    pcgnn_case("yelp_small", seed=3, n=1500, f=32, rel_deg=(2.5, 9, 28), pos_rate=0.145, emb=64, batch=256,
               rhos=(0.5, 0.2, 0.8, 2.0))
    pcgnn_case("amazon_small", seed=5, n=900, f=25, rel_deg=(8, 40, 20), pos_rate=0.09, emb=64, batch=128,
               rhos=(0.5, 0.8, 0.2), nonneg=True, lr=0.005, wd=0.0005)
    pcgnn_case("single_rel", seed=9, n=800, f=32, rel_deg=(12,), pos_rate=0.12, emb=32, batch=100,
               rhos=(0.5,))
    # BASELINE configs[2] shape: emb 128 (the dense tail's weights no longer fit the LDS: the L2-streamed kernel variant)
    pcgnn_case("yelp_emb128", seed=13, n=1500, f=32, rel_deg=(2.5, 9, 28), pos_rate=0.145, emb=128, batch=256,
               rhos=(0.5,))
    # wide features (F=100: 400-B rows, 32 lanes per row; again the streamed-weights variant, for the K dimension this time)
    pcgnn_case("feat100", seed=17, n=700, f=100, rel_deg=(4, 14, 9), pos_rate=0.15, emb=64, batch=96,
               rhos=(0.5, 0.8))
    # five relations (InterAgg5, layers.py:16-158), emb 48 (E/4 = 12 does not divide the 1024-thread staging pattern)
    pcgnn_case("five_rel", seed=37, n=700, f=16, rel_deg=(3, 6, 10, 5, 16), pos_rate=0.14, emb=48, batch=90,
               rhos=(0.5, 2.0))
