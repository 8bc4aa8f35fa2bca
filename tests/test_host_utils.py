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
