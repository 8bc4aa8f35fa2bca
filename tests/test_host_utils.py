"""CPU tests of the host-side mirrors in pc-gnn_amd/utils.py (metrics vs sklearn, ingestion helpers)."""
import numpy as np
import pytest

from pcgnn_amd import utils as U


def test_metrics_match_sklearn():
    sk = pytest.importorskip("sklearn.metrics")
    rs = np.random.RandomState(0)
    for n, rate in ((500, 0.15), (64, 0.5), (2000, 0.03)):
        y = (rs.rand(n) < rate).astype(int)
        y[0], y[1] = 0, 1
        score = np.round(rs.rand(n) * 0.6 + y * 0.25, 2)          # ties included
        pred = (score > 0.5).astype(int)
        m = U.binary_metrics(y, pred, score)
        assert abs(m["auc"] - sk.roc_auc_score(y, score)) < 1e-12
        assert abs(m["accuracy"] - sk.accuracy_score(y, pred)) < 1e-12
        assert abs(m["f1"] - sk.f1_score(y, pred)) < 1e-12
        assert abs(m["f1_macro"] - sk.f1_score(y, pred, average="macro")) < 1e-12
        assert abs(m["precision"] - sk.precision_score(y, pred, zero_division=0)) < 1e-12
        assert abs(m["precision_macro"] - sk.precision_score(y, pred, zero_division=0, average="macro")) < 1e-12
        assert abs(m["recall"] - sk.recall_score(y, pred)) < 1e-12
        assert abs(m["recall_macro"] - sk.recall_score(y, pred, average="macro")) < 1e-12


def test_pos_neg_split_and_normalize_and_csr():
    nodes, labels = [5, 9, 2, 7, 11], np.array([1, 0, 1, 0, 0])
    assert U.pos_neg_split(nodes, labels) == ([5, 2], [9, 7, 11])
    mx = np.abs(np.random.RandomState(1).randn(6, 4))
    out = np.asarray(U.normalize(mx))
    np.testing.assert_allclose(out, mx / (mx.sum(1, keepdims=True) + 0.01), rtol=1e-12)
    import scipy.sparse as sp
    a = sp.csc_matrix(np.array([[0, 1, 0], [0, 0, 0], [1, 0, 0]]))
    indptr, idx = U.sparse_to_csr(a)
    assert indptr.tolist() == [0, 3, 5, 7] and idx.tolist() == [0, 1, 2, 0, 1, 0, 2]


def test_get_best_f1_matches_the_reference_loop():
    """get_best_f1 (src/utils(f1).py:334-350) restated with one sort: same best F1 and threshold as the per-threshold
    sklearn loop the reference runs."""
    sk = pytest.importorskip("sklearn.metrics")
    rs = np.random.RandomState(5)
    for n, rate in ((1, 1.0), (50, 0.3), (4000, 0.1), (4000, 0.0)):
        y = (rs.rand(n) < rate).astype(np.int64)
        p = np.clip(0.35 * y + 0.5 * rs.rand(n), 0, 1)
        p = np.round(p, 2) if n == 50 else p            # exact ties with thresholds
        best_f1, best_t = 0, 0
        for t in np.linspace(0.01, 0.99, 100):
            pred = np.zeros_like(y)
            pred[p > t] = 1
            f = sk.f1_score(y, pred, zero_division=0)
            if f > best_f1:
                best_f1, best_t = f, t
        got_f1, got_t = U.get_best_f1(y, p)
        assert abs(got_f1 - best_f1) < 1e-12 and abs(got_t - best_t) < 1e-12, (n, rate, got_f1, best_f1, got_t, best_t)


def test_state_dict_keys_and_shapes_match_the_reference():
    """Checkpoint compatibility (model_handler.py:169,176: torch.save / load_state_dict of the model): the mirror
    classes expose exactly the reference's state-dict keys, in its order, with its shapes - including the feature
    table it registers four times.  Expected list captured from /root/reference's PCALayer(2, InterAgg3(...), alpha)."""
    import torch.nn as nn
    import pcgnn_amd as P
    n, f, e = 10, 4, 8
    feats = nn.Embedding(n, f)
    adj = [{i: {i} for i in range(n)}] * 3
    intra = [P.IntraAgg(feats, f, e, [1, 2], 0.5, cuda=False) for _ in range(3)]
    model = P.PCALayer(2, P.InterAgg3(feats, f, e, [1, 2], adj, intra, cuda=False), 2.0)
    want = [("weight", (2, e)), ("inter1.weight", (f + 3 * e, e)), ("inter1.features.weight", (n, f)),
            ("inter1.intra_agg1.weight", (2 * f, e)), ("inter1.intra_agg1.features.weight", (n, f)),
            ("inter1.intra_agg2.weight", (2 * f, e)), ("inter1.intra_agg2.features.weight", (n, f)),
            ("inter1.intra_agg3.weight", (2 * f, e)), ("inter1.intra_agg3.features.weight", (n, f)),
            ("inter1.label_clf.weight", (2, f)), ("inter1.label_clf.bias", (2,))]
    assert [(k, tuple(v.shape)) for k, v in model.state_dict().items()] == want


def test_stratified_split_matches_sklearn_and_fixture():
    """utils.train_test_split / split_dataset restate the two stratified sklearn.model_selection.train_test_split calls of
    src/model_handler.py:36-48: identical index lists and labels, against scikit-learn itself (when importable) and against
    the fixture scikit-learn wrote in the build container (tests/golden/split.npz)."""
    import os
    from tests.util import GOLDEN
    z = np.load(os.path.join(GOLDEN, "split.npz"))
    for tag in ("yelp", "amazon", "tiny"):
        got = U.split_dataset(z[f"{tag}_labels"], float(z[f"{tag}_train_ratio"]), 0.67, int(z[f"{tag}_seed"]), int(z[f"{tag}_first"]))
        for name, g in zip(("idx_train", "y_train", "idx_valid", "y_valid", "idx_test", "y_test"), got):
            assert np.array_equal(np.asarray(g), z[f"{tag}_{name}"]), (tag, name)
    sk = pytest.importorskip("sklearn.model_selection")
    rs = np.random.RandomState(1)
    for n, rate, tr, seed in ((1000, 0.15, 0.4, 3), (8639, 0.0687, 0.01, 7), (51, 0.3, 0.05, 11), (2000, 0.5, 0.1, 0)):
        y = (rs.rand(n) < rate).astype(int)
        idx = list(range(100, 100 + n))
        a = sk.train_test_split(idx, y, stratify=y, train_size=tr, random_state=seed, shuffle=True)
        b = U.train_test_split(idx, y, stratify=y, train_size=tr, random_state=seed)
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        a2 = sk.train_test_split(a[1], a[3], stratify=a[3], test_size=0.67, random_state=seed, shuffle=True)
        b2 = U.train_test_split(b[1], b[3], stratify=b[3], test_size=0.67, random_state=seed)
        assert a2[0] == b2[0] and a2[1] == b2[1] and np.array_equal(a2[2], b2[2]) and np.array_equal(a2[3], b2[3])


def test_result_manager_writes_the_reference_formats(tmp_path):
    """ResultManager (src/result_manager.py:18-157): directory layout, exp_id, log lines, DataFrame columns, and the test
    frame rebuilt from the logs of earlier runs of the same (model, data) pair."""
    import pandas as pd
    from pcgnn_amd.result_manager import METRICS, ResultManager
    cfg = dict(data_name="yelp", model="PCGNN", seed=2, train_ratio=0.4, test_ratio=0.67, emb_size=64, lr=0.01, weight_decay=0.001,
               alpha=2, rho=0.5, epochs=1000, valid_epochs=10, batch_size=1024, patience=100, exp_num="0003")
    root = str(tmp_path / "experimental_results")
    line = ("- F1: 0.5000\t- Recall: 0.6000\t- Precision: 0.4286\t- Accuracy: 0.8000\t- AUC-ROC: 0.8500\t- F1-macro: 0.7000\t"
            "- Recall-macro: 0.7200\t- AP: 0.6900\t\n")          # utils.py:325
    r1 = ResultManager(cfg, root=root)
    assert r1.exp_id.startswith("PCGNN-yelp-") and r1.model_path.endswith(f"saved_models/{r1.exp_id}.pickle")
    for sub in ("saved_models", "predictions", "validation_df", "test_df", "validation_log", "test_log"):
        assert (tmp_path / "experimental_results" / sub).is_dir()
    r1.write_val_log(9, 0, 0.8, 0.5, 0.7, 0.4286, 0.69, 0.6, 0.72, 0.85, line, print_line=False)
    r1.write_val_log(19, 9, 0.81, 0.5, 0.7, 0.4286, 0.69, 0.6, 0.72, 0.86, line, print_line=False)
    r1.write_test_log(19, 0.8, 0.5, 0.7, 0.4286, 0.69, 0.6, 0.72, 0.85, line, print_line=False)
    val = open(r1.log_val_path).read().splitlines()
    head = [f"{k}: {cfg[k]}" for k in sorted(cfg)]
    assert val[:len(head)] == head                                  # the configuration, one "key: value" per line, keys sorted
    assert val[len(head)] == "[Epoch-009] Validation performance" and val[len(head) + 1].startswith("- F1: 0.5000\t- Recall: 0.6000")
    tst = open(r1.log_test_path).read().splitlines()
    assert tst[len(head)].startswith("Test performance: - Epoch_Best: 19\t- F1: 0.5000")
    dv = pd.read_pickle(r1.df_val_path)
    assert list(dv.columns) == ["epoch", "epoch_best"] + list(METRICS) and dv["epoch"].tolist() == [9.0, 19.0]
    assert abs(dv["auc"].iloc[1] - 0.86) < 1e-12
    # a second run of the same pair: its test frame starts from the first run's test LOG (result_manager.py:47-75)
    r2 = ResultManager(cfg, root=root)
    r2.write_test_log(29, 0.9, 0.6, 0.8, 0.5, 0.7, 0.7, 0.8, 0.9, line, print_line=False)
    dt = pd.read_pickle(r2.df_test_path)
    assert len(dt) == 2 and set(dt["exp_id"]) == {r1.exp_id, r2.exp_id}
    first = dt[dt["exp_id"] == r1.exp_id].iloc[0]
    assert first["epoch_best"] == 19.0 and abs(first["auc"] - 0.85) < 1e-9 and abs(first["precision_macro"] - 0.69) < 1e-9
    assert first["data_name"] == "yelp" and str(first["exp_num"]) == "0003"
    assert r2.get_best_model_exp_id("auc") == r2.exp_id and r2.get_best_model_path("f1").endswith(".pickle")
    r2.save_predictions(np.arange(4), "test")
    assert (tmp_path / "experimental_results" / "predictions" / f"{r2.exp_id}-test.npy").exists()


def test_interagg_arities_and_state_dict_of_the_five_relation_model():
    """InterAgg1 / InterAgg5 (src/layers.py:417-535, 16-158): same constructor contract as InterAgg3; state-dict keys of R = 5."""
    import torch.nn as nn
    import pcgnn_amd as P
    n, f, e = 6, 4, 16
    feats = nn.Embedding(n, f)
    adj = [{i: {i} for i in range(n)}] * 5
    intra = [P.IntraAgg(feats, f, e, [1], 0.5, cuda=False) for _ in range(5)]
    m5 = P.PCALayer(2, P.InterAgg5(feats, f, e, [1], adj, intra, cuda=False), 2.0)
    keys = list(m5.state_dict().keys())
    assert keys[:2] == ["weight", "inter1.weight"] and tuple(m5.state_dict()["inter1.weight"].shape) == (f + 5 * e, e)
    assert [k for k in keys if k.endswith(".weight") and "intra_agg" in k and "features" not in k] == [f"inter1.intra_agg{r}.weight" for r in range(1, 6)]
    with pytest.raises(ValueError):
        P.InterAgg5(feats, f, e, [1], adj[:3], intra[:3], cuda=False)
    with pytest.raises(ValueError):
        P.InterAgg1(feats, f, e, [1], adj[:3], intra[:3], cuda=False)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` outside a launcher starts N ranks itself (torch.distributed.run, before any GPU call) and
    hands back their exit status; PCG_BENCH_DRY=1 stops every rank after the process group is up (no GPU here).  A launcher
    whose process count disagrees with --gpus is an error; and algorithmic_bytes follows SURVEY 8(d)'s terms."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["PCG_BENCH_DRY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0]) == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2}
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env2,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
    sys.path.insert(0, root)
    import bench

    class G:
        R, feat_dim = 2, 8
        deg_host = [np.array([3, 5, 0]), np.array([1, 1, 4])]
    ids = np.array([0, 2])
    counts = [np.array([2, 0]), np.array([1, 2])]
    # per relation: 4 * (2B + D) CSR + 4 * D scores + 4 * F * S rows + 4 * B * F output
    want = (4 * (4 + 3) + 4 * 3 + 4 * 8 * 2 + 4 * 2 * 8) + (4 * (4 + 5) + 4 * 5 + 4 * 8 * 3 + 4 * 2 * 8)
    assert bench.algorithmic_bytes(G, ids, counts) == want


def test_bench_algorithmic_bytes_counts_the_riders_once():
    """bench.py's byte count of the select + gather call: the SURVEY 8(d) terms of the call itself, plus - for the training
    call, whose two launches also carry the next step's score pass, the label classifier's step and the deferred Adam update -
    the table stream, the train positives' rows, the centres' rows and the optimizer state; never the per-tile slabs."""
    import importlib.util
    import os
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    deg = [np.array([3, 0, 5, 2]), np.array([1, 1, 1, 1])]
    g = types.SimpleNamespace(R=2, feat_dim=8, deg_host=deg)
    ids = np.array([0, 2, 2])
    counts = [np.array([2, 3, 3]), np.array([1, 1, 1])]
    base = bench.algorithmic_bytes(g, ids, counts)
    B, F = 3, 8
    want = 0
    for r in range(2):
        D, S = int(deg[r][ids].sum()), int(counts[r].sum())
        want += 4 * (2 * B + D) + 4 * D + 4 * F * S + 4 * B * F
    assert base == want
    riders = dict(table_rows=100, n_pos=7, n_params=50)
    assert bench.algorithmic_bytes(g, ids, counts, riders) == want + 100 * (4 * F + 4) + 7 * (4 * F + 8) + B * (4 * F + 8) + 28 * 50
    assert bench.algorithmic_bytes(g, ids, counts, dict(table_rows=0, n_pos=7, n_params=50)) < bench.algorithmic_bytes(g, ids, counts, riders)
    # U counted from the CSR: the batch's unique nodes and all their neighbours (layers.py:226-227)
    csr = [(np.array([0, 2, 2, 4, 5]), np.array([1, 3, 0, 3, 2])), (np.array([0, 1, 2, 3, 4]), np.array([0, 1, 2, 3]))]
    assert bench.unique_rows(csr, 4, ids) == 4                                     # {0, 2} + {1, 3} + {0, 3} = every node
    assert bench.unique_rows(csr, 4, np.array([1])) == 1
    by_csr = bench.algorithmic_bytes(g, ids, counts, dict(csr=csr, n_nodes=4, n_pos=7, n_params=50))
    assert by_csr == bench.algorithmic_bytes(g, ids, counts, dict(table_rows=4, n_pos=7, n_params=50))


def test_relabel_by_degree_is_the_same_graph():
    """synth.relabel_by_degree (bench.py --relabel-by-degree, a diagnostic): the graph renumbered by descending total degree is
    isomorphic to the original - same edges under the permutation, rows ascending in the new ids, features / labels / training
    split carried along, the train positives in their original ORDER (it is the minority picks' tie-break)."""
    from pcgnn_amd import synth
    w = synth.power_law(3000, 30000, 1)
    r = synth.relabel_by_degree(w)
    deg_w = sum(np.diff(ip) for ip, _ in w.csr)
    deg_r = sum(np.diff(ip) for ip, _ in r.csr)
    assert np.all(np.diff(deg_r) <= 0) and sorted(deg_w.tolist()) == sorted(deg_r.tolist())
    # recover the permutation from the features (distinct rows): new -> old
    key = {tuple(np.round(row, 6)): i for i, row in enumerate(w.X)}
    perm = np.array([key[tuple(np.round(row, 6))] for row in r.X])
    assert sorted(perm.tolist()) == list(range(w.n))
    inv = np.empty(w.n, dtype=np.int64)
    inv[perm] = np.arange(w.n)
    assert np.array_equal(r.labels, w.labels[perm]) and np.array_equal(r.homo_deg, w.homo_deg[perm])
    assert np.array_equal(r.idx_train, np.sort(inv[w.idx_train]))
    assert r.train_pos == [int(inv[v]) for v in w.train_pos]
    for (ip_w, idx_w), (ip_r, idx_r) in zip(w.csr, r.csr):
        for new in (0, 1, 17, w.n - 1):
            row = idx_r[ip_r[new]:ip_r[new + 1]]
            assert np.all(np.diff(row) > 0)
            old = perm[new]
            assert sorted(inv[idx_w[ip_w[old]:ip_w[old + 1]]].tolist()) == row.tolist()
