"""torch-tensor level wrappers over the C ABI (include/pcgnn.h).

Every function enqueues HIP kernels on torch's current stream and returns
torch tensors that live on the graph's device; nothing here computes on the CPU
and nothing synchronises (except ``chosen_sets``, a test/debug helper that
copies index sets back to the host).
"""
import ctypes as C
from typing import List, Optional, Sequence, Set

import numpy as np
import torch

from . import _lib
from .graph import DeviceGraph


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _i32(x, device) -> torch.Tensor:
    """ids / labels in whatever the reference passes (python list, numpy, LongTensor) -> int32 on device."""
    if torch.is_tensor(x):
        return x.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.int32)).to(device, non_blocking=True)


def score_table(g: DeviceGraph, W: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None,
                row_begin: int = 0, row_end: Optional[int] = None) -> torch.Tensor:
    """Class-0 label-aware logit of every node (layers.py:230-237)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(g.n_nodes, dtype=torch.float32, device=g.device)
    W = W.detach().contiguous()
    b = b.detach().contiguous()
    _lib.check(lib.pcg_score_table(g.desc_ref(), _p(W), _p(b), row_begin, g.n_nodes if row_end is None else row_end,
                                   _p(out), _stream(g.device)), "pcg_score_table")
    return out


def score_rows(g: DeviceGraph, W: torch.Tensor, b: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """Both label-aware logits of the given rows -> [n, 2] (layers.py:243)."""
    lib = _lib.load()
    out = torch.empty(ids.numel(), 2, dtype=torch.float32, device=g.device)
    W = W.detach().contiguous()
    b = b.detach().contiguous()
    _lib.check(lib.pcg_score_rows(g.desc_ref(), _p(W), _p(b), _p(ids), ids.numel(), _p(out), _stream(g.device)),
               "pcg_score_rows")
    return out


def pos_sort(g: DeviceGraph, s0: torch.Tensor, keys: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sorted (score, position) keys of the training positives (uint64 bit patterns in an int64 tensor)."""
    lib = _lib.load()
    cap = lib.pcg_pos_sort_capacity(g.n_pos)
    if keys is None:
        keys = torch.empty(cap, dtype=torch.int64, device=g.device)
    _lib.check(lib.pcg_pos_sort(g.desc_ref(), _p(s0), _p(keys), _stream(g.device)), "pcg_pos_sort")
    return keys


def gather_rows(g: DeviceGraph, ids: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    lib = _lib.load()
    if out is None:
        out = torch.empty(ids.numel(), g.feat_dim, dtype=torch.float32, device=g.device)
    _lib.check(lib.pcg_gather_rows(g.desc_ref(), _p(ids), ids.numel(), _p(out), out.stride(0), _stream(g.device)),
               "pcg_gather_rows")
    return out


class ChooseWorkspace:
    """Per-(graph, batch size) scratch so the step itself never allocates: the tier queues,
    the selection list every row's chosen ids are written to, the gather's chunk table and
    partial sums.  ``list_capacity`` (entries) must cover sum over rows of the per-row bound
    ``(deg > k+1 ? k : deg) + m (+1)``; the default is the worst case for this graph and batch
    size (every centre being the largest hub), clipped to ``max_list_bytes``."""

    def __init__(self, g: DeviceGraph, B: int, list_capacity: Optional[int] = None, max_list_bytes: int = 8 << 30,
                 status: Optional[torch.Tensor] = None):
        lib = _lib.load()
        self.B = B
        self.clipped = False       # the default capacity was cut below the worst case: overflow is possible, check()!
        if list_capacity is None:
            # kept <= deg, minority m = int(ceil(deg/2) * rho) <= 2 * deg for rho <= 4 (and <= n_pos), +1 self
            per_row = g.max_degree + min(2 * max(g.max_degree, 1), g.n_pos) + 1
            worst = per_row * g.R * max(B, 1)
            list_capacity = min(worst, max_list_bytes // 4, (1 << 31) - 1)
            self.clipped = list_capacity < worst
        self.list_capacity = int(max(list_capacity, 1))
        nbytes = lib.pcg_choose_workspace_bytes(g.desc_ref(), B, self.list_capacity)
        if nbytes < 0:
            raise _lib.PcgnnLibraryError("pcg_choose_workspace_bytes rejected the arguments")
        self.buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=g.device)
        self.status = status if status is not None else torch.zeros(1, dtype=torch.int32, device=g.device)
        self._g = g

    def view(self, which: int, dtype, count: int) -> torch.Tensor:
        off = _lib.load().pcg_choose_workspace_offset(self._g.desc_ref(), self.B, self.list_capacity, which)
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return self.buf[off:off + nbytes].view(dtype)

    def check(self):
        """Raise if a batch did not fit the selection list (reads the status word: synchronises)."""
        if int(self.status.item()) & _lib.PCG_ST_SEL_OVERFLOW:
            self.status.zero_()
            raise _lib.PcgnnLibraryError("selection list overflow: raise ChooseWorkspace(list_capacity=...)")


def sel_capacity(g: DeviceGraph, nodes_host: np.ndarray, labels_host: Optional[np.ndarray],
                 thresholds: Sequence[float], rho, train_flag: bool, add_self: bool = False) -> np.ndarray:
    """Host-side upper bound of every row's chosen-set size, [R, B] (pcg_sel_capacity_row, vectorised)."""
    caps = []
    for r in range(g.R):
        deg = g.deg_host[r][nodes_host].astype(np.int64)
        k = np.ceil(deg * float(thresholds[r])).astype(np.int64)
        cap = np.where(deg > k + 1, k, deg)
        if train_flag:
            rr = float(rho) if np.isscalar(rho) else float(rho[r])
            m = np.minimum((k * rr).astype(np.int64), g.n_pos)
            cap = cap + np.where(np.asarray(labels_host) == 1, np.maximum(m, 0), 0)
        caps.append(cap + (1 if add_self else 0))
    return np.stack(caps)


def _host_arrays(g, thresholds, rho):
    thr = (C.c_double * g.R)(*[float(t) for t in thresholds])
    rhos = (C.c_double * g.R)(*([float(rho)] * g.R if np.isscalar(rho) else [float(x) for x in rho]))
    return thr, rhos


def choose_select(g: DeviceGraph, nodes, labels, s0, pos_keys, thresholds, rho, train_flag: bool, ws: ChooseWorkspace,
                  cnt: torch.Tensor, add_self: bool = False, center_s0=None, planned: bool = False):
    """plan + select only: every row's chosen ids into ws's selection list, |set| into cnt [R*B].
    planned: ws already holds this batch's plan (step_front / step_front_a + _b)."""
    lib = _lib.load()
    thr, rhos = _host_arrays(g, thresholds, rho)
    args = (g.desc_ref(), _p(nodes), _p(labels), nodes.numel(), _p(s0), _p(center_s0), _p(pos_keys), thr, rhos,
            1 if train_flag else 0, 1 if add_self else 0, _p(cnt), _p(ws.buf))
    rest = (ws.list_capacity, _p(ws.status), _stream(g.device))
    if planned:
        _lib.check(lib.pcg_choose_select_planned(*args, None, ws.list_capacity, _p(ws.status), None, 0, _stream(g.device)),
                   "pcg_choose_select_planned")
    else:
        _lib.check(lib.pcg_choose_select(*args, *rest), "pcg_choose_select")


def aggregate_lists(g: DeviceGraph, X: torch.Tensor, B: int, ws: ChooseWorkspace, cnt: torch.Tensor, agg: torch.Tensor,
                    norm: int = _lib.PCG_NORM_COUNT):
    """gather + mean of the lists ws holds, from feature table X [*, feat_stride] (g.X or an extended table)."""
    lib = _lib.load()
    _lib.check(lib.pcg_aggregate_lists(_p(X), g.feat_dim, X.stride(0), X.shape[0], g.R * B, _p(cnt), g.desc_ref(), B, _p(ws.buf),
                                       ws.list_capacity, norm, _p(agg), agg.stride(-2), _p(ws.status), _stream(g.device)),
               "pcg_aggregate_lists")


def choose_aggregate(g: DeviceGraph, nodes: torch.Tensor, labels: Optional[torch.Tensor], s0: torch.Tensor,
                     pos_keys: Optional[torch.Tensor], thresholds: Sequence[float], rho, train_flag: bool,
                     norm: int = _lib.PCG_NORM_COUNT, add_self: bool = False,
                     center_s0: Optional[torch.Tensor] = None, ws: Optional[ChooseWorkspace] = None,
                     agg: Optional[torch.Tensor] = None, cnt: Optional[torch.Tensor] = None, planned: bool = False):
    """choose + mean for all relations of a batch -> agg [R, B, F] (and |set| [R, B]).
    planned: ws already holds this batch's plan (step_front)."""
    lib = _lib.load()
    B = nodes.numel()
    if ws is None or ws.B != B:
        ws = ChooseWorkspace(g, B)
    if agg is None:
        agg = torch.empty(g.R, B, g.feat_dim, dtype=torch.float32, device=g.device)
    if cnt is None:
        cnt = torch.empty(g.R, B, dtype=torch.int32, device=g.device)
    thr, rhos = _host_arrays(g, thresholds, rho)
    fn = lib.pcg_choose_aggregate_planned if planned else lib.pcg_choose_aggregate
    _lib.check(fn(
        g.desc_ref(), _p(nodes), _p(labels), B, _p(s0), _p(center_s0), _p(pos_keys), thr, rhos,
        1 if train_flag else 0, norm, 1 if add_self else 0, _p(agg), agg.stride(-2), _p(cnt),
        _p(ws.buf), ws.list_capacity, _p(ws.status), _stream(g.device)), "pcg_choose_aggregate")
    return agg, cnt


def step_front(g: DeviceGraph, W: torch.Tensor, b: torch.Tensor, s0: torch.Tensor, pos_keys: Optional[torch.Tensor],
               nodes: torch.Tensor, labels: Optional[torch.Tensor], thresholds: Sequence[float], rho, train_flag: bool,
               ws: ChooseWorkspace, add_self: bool = False):
    """score table + train-pos sort + the batch's plan in two launches (pcg_step_front); follow with
    choose_aggregate(..., planned=True) on the same arguments.  Returns the sorted keys (or None)."""
    lib = _lib.load()
    B = nodes.numel()
    assert ws.B == B
    thr, rhos = _host_arrays(g, thresholds, rho)
    sort = bool(train_flag) and g.n_pos > 0
    _lib.check(lib.pcg_step_front(
        g.desc_ref(), _p(W), _p(b), _p(s0), _p(pos_keys) if sort else None, _p(nodes), _p(labels), B, thr, rhos,
        1 if train_flag else 0, 1 if add_self else 0, _p(ws.buf), ws.list_capacity, _p(ws.status),
        _stream(g.device)), "pcg_step_front")
    return pos_keys if sort else None


def read_sets(g: DeviceGraph, B: int, ws: ChooseWorkspace):
    """Copy the selection list of the last call to the host -> sets[r][b] (synchronises)."""
    rows = g.R * B
    torch.cuda.synchronize(g.device)
    ws.check()
    begin = ws.view(0, torch.int64, rows + 1).cpu().numpy()
    length = ws.view(1, torch.int32, rows).cpu().numpy()
    total = int(begin[-1])
    lst = ws.view(2, torch.int32, max(total, 1)).cpu().numpy()
    sets: List[List[Set[int]]] = []
    for r in range(g.R):
        row_sets = []
        for b in range(B):
            row = r * B + b
            seg = lst[begin[row]:begin[row] + length[row]]
            row_sets.append(set(seg[seg >= 0].tolist()))
        sets.append(row_sets)
    return sets


def chosen_sets(g: DeviceGraph, nodes, labels, s0, pos_keys, thresholds, rho, train_flag,
                norm=_lib.PCG_NORM_COUNT, add_self=False, center_s0=None, list_capacity=None):
    """Debug / parity helper: run the hot path and copy the chosen index sets to the host.
    Returns (sets[r][b], agg, cnt)."""
    B = nodes.numel()
    if list_capacity is None:     # exact requirement of this batch (host arithmetic)
        labels_h = None if labels is None else labels.cpu().numpy()
        list_capacity = int(sel_capacity(g, nodes.cpu().numpy(), labels_h, thresholds, rho, train_flag, add_self).sum())
    ws = ChooseWorkspace(g, B, list_capacity=max(list_capacity, 1))
    agg, cnt = choose_aggregate(g, nodes, labels, s0, pos_keys, thresholds, rho, train_flag, norm, add_self,
                                center_s0, ws)
    sets = read_sets(g, B, ws)
    return sets, agg, cnt


def segment_mean(g: DeviceGraph, begin: torch.Tensor, count: torch.Tensor, idx: torch.Tensor,
                 norm: int = _lib.PCG_NORM_COUNT, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Mean of explicit index lists (mask.div(n).mm(X[unique]), layers.py:599-624)."""
    lib = _lib.load()
    n = begin.numel()
    if out is None:
        out = torch.empty(n, g.feat_dim, dtype=torch.float32, device=g.device)
    _lib.check(lib.pcg_segment_mean(g.desc_ref(), _p(begin), _p(count), _p(idx), n, norm, _p(out), out.stride(0),
                                    _stream(g.device)), "pcg_segment_mean")
    return out


def pick(cum: torch.Tensor, idx_train: torch.Tensor, k: int, uniforms: Optional[torch.Tensor] = None,
         seed: int = 0, epoch: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Label-balanced pick (random.choices with cumulative weights, utils.py:274-278)."""
    lib = _lib.load()
    if out is None:
        out = torch.empty(k, dtype=torch.int32, device=cum.device)
    _lib.check(lib.pcg_pick(_p(cum), _p(idx_train), idx_train.numel(), _p(uniforms), seed & (2 ** 64 - 1),
                            epoch & (2 ** 64 - 1), k, _p(out), _stream(cum.device)), "pcg_pick")
    return out


def pick_shuffled(cum: torch.Tensor, idx_train: torch.Tensor, k: int, seed: int, epoch_base: int,
                  out_ids: torch.Tensor, labels_all: Optional[torch.Tensor] = None,
                  out_labels: Optional[torch.Tensor] = None, epoch_counter: Optional[torch.Tensor] = None,
                  bump: bool = False, n_epochs: int = 1):
    """An epoch's picks, shuffled, with their labels, in one launch (pcg_pick_shuffled): the same draws as
    ``pick(seed, epoch)``, in a uniformly random order.  epoch = epoch_base + epoch_counter[0] (a device int64 the
    call increments afterwards when ``bump``: a captured graph then replays a new epoch every time).  n_epochs > 1: that many
    consecutive epochs in one launch, epoch e's k draws at out_ids[e * k:] (the counter moves on by n_epochs)."""
    lib = _lib.load()
    _lib.check(lib.pcg_pick_shuffled_epochs(_p(cum), _p(idx_train), idx_train.numel(), seed & (2 ** 64 - 1),
                                            epoch_base & (2 ** 64 - 1), _p(epoch_counter), 1 if bump else 0, n_epochs, k,
                                            _p(labels_all), _p(out_ids), _p(out_labels), _stream(cum.device)),
               "pcg_pick_shuffled_epochs")
    return out_ids


def step_front_a(g: DeviceGraph, W: torch.Tensor, b: torch.Tensor, s0_out: torch.Tensor, row_begin: int, row_end: int,
                 nodes: torch.Tensor, labels: Optional[torch.Tensor], thresholds: Sequence[float], rho, train_flag: bool,
                 ws: ChooseWorkspace, add_self: bool = False, row_ids: Optional[torch.Tensor] = None):
    """first half of step_front: scores of rows [row_begin, row_end) (as score_table; with row_ids: row r's score goes to
    s0_out[row_ids[r]], negative = skipped) || plan pass 1."""
    lib = _lib.load()
    thr, rhos = _host_arrays(g, thresholds, rho)
    _lib.check(lib.pcg_step_front_a(g.desc_ref(), _p(W), _p(b), row_begin, row_end, _p(s0_out), _p(row_ids), None, _p(nodes), _p(labels),
                                    nodes.numel(), thr, rhos, 1 if train_flag else 0, 1 if add_self else 0, _p(ws.buf),
                                    ws.list_capacity, _p(ws.status), _stream(g.device)), "pcg_step_front_a")


def step_front_b(g: DeviceGraph, s0: torch.Tensor, pos_keys: Optional[torch.Tensor], nodes: torch.Tensor,
                 labels: Optional[torch.Tensor], thresholds: Sequence[float], rho, train_flag: bool, ws: ChooseWorkspace,
                 add_self: bool = False, center_out: Optional[torch.Tensor] = None, center_id_offset: int = 0):
    """second half of step_front: train-pos sort by s0 || plan pass 2 (|| center_out[i] = s0[nodes[i] + center_id_offset]).
    Returns the sorted keys (or None)."""
    lib = _lib.load()
    thr, rhos = _host_arrays(g, thresholds, rho)
    sort = bool(train_flag) and g.n_pos > 0
    _lib.check(lib.pcg_step_front_b(g.desc_ref(), _p(s0), _p(pos_keys) if sort else None, 0, _p(nodes), _p(labels),
                                    nodes.numel(), thr, rhos, 1 if train_flag else 0, 1 if add_self else 0, _p(ws.buf),
                                    ws.list_capacity, _p(ws.status), _p(center_out), center_id_offset, _stream(g.device)),
               "pcg_step_front_b")
    return pos_keys if sort else None
