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
