"""Host-side helpers around the hot path, mirroring the reference's src/utils.py where the
training / evaluation loop touches them: ``test`` (:280-333), ``pos_neg_split`` (:256-271),
``normalize`` (:212-222), ``sparse_to_adjlist_for_train`` / graph ingestion (:226-254), ``set_seeds``.

``test`` keeps the reference's signature and return value; predictions stay on the device for the
whole pass and come back in ONE copy (the reference copies every batch, :305), and the
degenerate empty trailing batch its ``int(len/B)+1`` produces is not run.  Metrics are computed
with numpy restatements of the sklearn functions the reference calls (checked against sklearn in
the tests), so evaluation does not depend on sklearn being installed.
"""
import random
from typing import Optional, Tuple

import numpy as np
import torch


# ---- metrics (sklearn.metrics.{accuracy,f1,precision,recall,roc_auc}_score restated) --------------
def _prf(y, p, cls):
    tp = float(np.sum((p == cls) & (y == cls)))
    fp = float(np.sum((p == cls) & (y != cls)))
    fn = float(np.sum((p != cls) & (y == cls)))
    prec = tp / (tp + fp) if tp + fp > 0 else 0.0         # zero_division=0 (utils.py:319)
    rec = tp / (tp + fn) if tp + fn > 0 else 0.0
    f1 = 2 * prec * rec / (prec + rec) if prec + rec > 0 else 0.0
    return prec, rec, f1


def binary_metrics(y_true, y_pred, y_score) -> dict:
    y = np.asarray(y_true).astype(np.int64)
    p = np.asarray(y_pred).astype(np.int64)
    p1, r1, f1 = _prf(y, p, 1)
    p0, r0, f0 = _prf(y, p, 0)
    return {"accuracy": float(np.mean(y == p)), "f1": f1, "f1_macro": (f1 + f0) / 2, "precision": p1,
            "precision_macro": (p1 + p0) / 2, "recall": r1, "recall_macro": (r1 + r0) / 2, "auc": roc_auc(y, y_score)}


def roc_auc(y_true, score) -> float:
    """Area under the ROC curve = Mann-Whitney U with average ranks for ties."""
    y = np.asarray(y_true).astype(bool)
    s = np.asarray(score, dtype=np.float64)
    n1, n0 = int(y.sum()), int((~y).sum())
    if n1 == 0 or n0 == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), dtype=np.float64)
    sorted_s = s[order]
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and sorted_s[j + 1] == sorted_s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def test(test_nodes, labels, model, batch_size: int, result=None, epoch: Optional[int] = None,
         epoch_best: Optional[int] = None, flag: Optional[str] = None,
         print_line: Optional[bool] = True) -> Tuple[float, float, float, float]:
    """Evaluate ``model`` (PCALayer / GCN / GraphSage mirror, or a FusedPCGNN) on ``test_nodes``:
    batched ``to_prob(..., train_flag=False)`` -> argmax / positive-class confidence -> metrics.
    Returns (auc, recall, f1_macro, precision) like the reference (utils.py:333)."""
    nodes = np.asarray(test_nodes)
    labels = np.asarray(labels)
    outs = []
    fused = hasattr(model, "predict")                                       # FusedPCGNN: ids go to the device ONCE, batches are views
    ids_dev = torch.as_tensor(nodes, dtype=torch.int32, device=model.dev) if fused else None
    with torch.no_grad():
        for start in range(0, len(nodes), batch_size):                      # :298-303 (no empty trailing batch)
            batch = nodes[start:start + batch_size]
            blab = labels[start:start + batch_size]
            if fused:
                outs.append(torch.sigmoid(model.predict(ids_dev[start:start + batch_size], None, False)[0]))
            else:
                outs.append(model.to_prob(batch.tolist(), blab, train_flag=False)[0])   # :305
    prob = torch.cat(outs).float().cpu().numpy() if outs else np.zeros((0, 2), np.float32)
    if hasattr(model, "check"):
        model.check()          # a batch that overflowed its selection list must not pass for a prediction
    pred = prob.argmax(axis=1)                                               # :306
    m = binary_metrics(labels, pred, prob[:, 1])                             # :308, :316-323
    line = (f"- F1: {m['f1']:.4f}\t- Recall: {m['recall']:.4f}\t- Precision: {m['precision']:.4f}\t"
            f"- Accuracy: {m['accuracy']:.4f}\t- AUC-ROC: {m['auc']:.4f}\t- F1-macro: {m['f1_macro']:.4f}\t"
            f"- Recall-macro: {m['recall_macro']:.4f}\t- AP: {m['precision_macro']:.4f}\t\n")   # :325
    if result is not None:
        args = (m["accuracy"], m["f1"], m["f1_macro"], m["precision"], m["precision_macro"], m["recall"],
                m["recall_macro"], m["auc"], line, print_line)
        if flag == "val":
            result.write_val_log(epoch, epoch_best, *args)
        elif flag == "test":
            result.write_test_log(epoch_best, *args)
    elif print_line:
        print(line, end="")
    return m["auc"], m["recall"], m["f1_macro"], m["precision"]


def get_best_f1(labels, probs, thresholds=None) -> Tuple[float, float]:
    """Best positive-class F1 over the reference's 100 thresholds linspace(0.01, 0.99) and the threshold reaching it
    (the "(f1)" variant, src/utils(f1).py:334-350: one sklearn f1_score call per threshold; the first threshold wins
    ties).  Here: one sort of the confidences, then every threshold's TP / FP by binary search."""
    y = np.asarray(labels).astype(np.int64)
    p = np.asarray(probs, dtype=np.float64)
    th = np.linspace(0.01, 0.99, 100) if thresholds is None else np.asarray(thresholds, dtype=np.float64)
    order = np.argsort(p, kind="stable")
    ps, ys = p[order], y[order]
    pos_from = np.concatenate([np.cumsum(ys[::-1])[::-1], [0]])       # positives among ps[i:]
    first = np.searchsorted(ps, th, side="right")                     # predictions 1: probs > thresh  (:345)
    tp = pos_from[first].astype(np.float64)
    n_pred = (len(ps) - first).astype(np.float64)
    n_pos = float(ys.sum())
    denom = n_pred + n_pos                                             # 2TP + FP + FN
    f1 = np.where(denom > 0, 2.0 * tp / np.maximum(denom, 1.0), 0.0)
    best_f1, best_t = 0.0, 0.0                                         # :342 (a threshold must beat 0 to be taken)
    for f, t in zip(f1, th):
        if f > best_f1:
            best_f1, best_t = float(f), float(t)
    return best_f1, best_t


def test_f1(test_nodes, labels, model, batch_size: int, flag: str = "valid", valid_thresh: Optional[float] = None):
    """The "(f1)" evaluation (src/utils(f1).py:280-332): F1-macro at the best validation threshold (flag "valid":
    searched here and returned; otherwise ``valid_thresh`` is applied), the other metrics from the argmax prediction.
    Returns (auc, recall, f1_macro, precision, threshold) like the reference."""
    nodes = np.asarray(test_nodes)
    y = np.asarray(labels)
    outs = []
    with torch.no_grad():
        for start in range(0, len(nodes), batch_size):
            batch = nodes[start:start + batch_size]
            if hasattr(model, "predict"):
                ids = torch.as_tensor(batch, dtype=torch.int32, device=model.dev)
                outs.append(torch.sigmoid(model.predict(ids, None, False)[0]))
            else:
                outs.append(model.to_prob(batch.tolist(), y[start:start + batch_size], train_flag=False)[0])
    prob = torch.cat(outs).float().cpu().numpy() if outs else np.zeros((0, 2), np.float32)
    if hasattr(model, "check"):
        model.check()
    threshold = None
    if flag == "valid":
        _, threshold = get_best_f1(y, prob[:, 1])                     # :316-317
        cut = threshold
    else:
        cut = valid_thresh                                            # :320
    preds = (prob[:, 1] > cut).astype(np.int64)
    f1_macro = 0.5 * (_prf(y, preds, 1)[2] + _prf(y, preds, 0)[2])    # :318 / :322
    m = binary_metrics(y, prob.argmax(axis=1), prob[:, 1])            # :324-330
    return m["auc"], m["recall"], f1_macro, m["precision"], threshold


# ---- data helpers ------------------------------------------------------------------------------------
def pos_neg_split(nodes, labels):
    """positive / negative node ids in input order (utils.py:256-271, which is O(n^2) there)."""
    nodes = list(nodes)
    lab = np.asarray(labels)
    pos = [n for n, l in zip(nodes, lab) if l == 1]
    neg = [n for n, l in zip(nodes, lab) if l != 1]
    return pos, neg


def normalize(mx):
    """Row-normalise a dense or scipy-sparse feature matrix: x / (rowsum + 0.01) (utils.py:212-222)."""
    import scipy.sparse as sp
    rowsum = np.array(mx.sum(1)) + 0.01
    r_inv = np.power(rowsum, -1).flatten()
    r_inv[np.isinf(r_inv)] = 0.
    return sp.diags(r_inv).dot(mx)


def sparse_to_csr(sp_matrix):
    """What sparse_to_adjlist_for_train (utils.py:243-254) builds - self-loops added, symmetrised - but
    straight to the device layout: (indptr int64, indices int32 ascending), no dict-of-sets in between."""
    import scipy.sparse as sp
    a = sp.csr_matrix(sp_matrix)
    a = a + a.T + sp.eye(a.shape[0], format="csr")
    a.sum_duplicates()
    a.sort_indices()
    return a.indptr.astype(np.int64), a.indices.astype(np.int32)


def set_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ---- stratified train / valid / test split (src/model_handler.py:36-48) ------------------------------------------------------
def _approximate_mode(class_counts: np.ndarray, n_draws: int, rng: np.random.RandomState) -> np.ndarray:
    """How many of n_draws go to every class: proportional, the remainders settled largest first with random tie-breaks
    (scikit-learn 1.7 ``utils/_approximate_mode``; consumes the generator exactly as it does)."""
    continuous = class_counts / class_counts.sum() * n_draws
    floored = np.floor(continuous)
    need = int(n_draws - floored.sum())
    if need > 0:
        remainder = continuous - floored
        for value in np.sort(np.unique(remainder))[::-1]:
            (inds,) = np.where(remainder == value)
            add_now = min(len(inds), need)
            inds = rng.choice(inds, size=add_now, replace=False)
            floored[inds] += 1
            need -= add_now
            if need == 0:
                break
    return floored.astype(int)


def train_test_split(index, labels, stratify, train_size: Optional[float] = None, test_size: Optional[float] = None,
                     random_state: int = 0, shuffle: bool = True):
    """``sklearn.model_selection.train_test_split(index, labels, stratify=labels, train_size=... | test_size=...,
    random_state=seed, shuffle=True)`` restated (scikit-learn 1.7: one StratifiedShuffleSplit draw) so that the splits
    the reference makes (model_handler.py:42-43, 47-48) can be reproduced without scikit-learn: same generator
    (``np.random.RandomState(seed)``), same order of draws, same outputs - checked against scikit-learn in the tests.
    Returns (idx_train, idx_test, y_train, y_test) as lists / arrays like the reference consumes them."""
    assert shuffle and stratify is not None
    index, labels, y = list(index), np.asarray(labels), np.asarray(stratify)
    n = len(index)
    if test_size is None and train_size is None:
        test_size = 0.25
    # scikit-learn's _validate_shuffle_split for float sizes
    n_test = int(np.ceil(test_size * n)) if test_size is not None else None
    n_train = int(np.floor(train_size * n)) if train_size is not None else None
    if n_train is None:
        n_train = n - n_test
    elif n_test is None:
        n_test = n - n_train
    if n_train + n_test > n or n_train <= 0:
        raise ValueError("train_size / test_size do not fit the number of samples")
    rng = np.random.RandomState(random_state)
    classes, y_indices = np.unique(y, return_inverse=True)
    class_counts = np.bincount(y_indices)
    if class_counts.min() < 2:
        raise ValueError("The least populated class in y has only 1 member, which is too few.")
    if n_train < len(classes) or n_test < len(classes):
        raise ValueError("fewer samples than classes in one side of the split")
    class_indices = np.split(np.argsort(y_indices, kind="mergesort"), np.cumsum(class_counts)[:-1])
    n_i = _approximate_mode(class_counts, n_train, rng)
    t_i = _approximate_mode(class_counts - n_i, n_test, rng)
    train, test = [], []
    for i in range(len(classes)):
        perm = class_indices[i].take(rng.permutation(class_counts[i]), mode="clip")
        train.extend(perm[:n_i[i]])
        test.extend(perm[n_i[i]:n_i[i] + t_i[i]])
    train, test = rng.permutation(train), rng.permutation(test)
    pick = lambda seq, ids: [seq[i] for i in ids]
    return pick(index, train), pick(index, test), labels[train], labels[test]


def split_dataset(labels, train_ratio: float, test_ratio: float, seed: int, first_labeled: int = 0):
    """The reference's train / valid / test split (model_handler.py:36-48): nodes [first_labeled, N) (Amazon: the first 3305 -
    amazon_new: 2013 - are unlabeled), ``train_ratio`` of them for training, ``test_ratio`` of the rest for testing, both
    stratified, both seeded with ``seed``.  Returns idx_train, y_train, idx_valid, y_valid, idx_test, y_test."""
    labels = np.asarray(labels)
    index = list(range(first_labeled, len(labels)))
    lab = labels[first_labeled:]
    idx_train, idx_rest, y_train, y_rest = train_test_split(index, lab, stratify=lab, train_size=train_ratio, random_state=seed)
    idx_valid, idx_test, y_valid, y_test = train_test_split(idx_rest, y_rest, stratify=y_rest, test_size=test_ratio, random_state=seed)
    return idx_train, y_train, idx_valid, y_valid, idx_test, y_test
