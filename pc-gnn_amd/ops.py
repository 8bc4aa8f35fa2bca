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
    """Per-(graph, batch size) scratch so the step itself never allocates."""

    def __init__(self, g: DeviceGraph, B: int):
        lib = _lib.load()
        self.B = B
        nbytes = lib.pcg_choose_workspace_bytes(g.desc_ref(), B)
        self.buf = torch.zeros(max(int(nbytes), 256), dtype=torch.uint8, device=g.device)   # counters start at zero
        self.status = torch.zeros(1, dtype=torch.int32, device=g.device)


def sel_capacity(g: DeviceGraph, nodes_host: np.ndarray, labels_host: Optional[np.ndarray],
                 thresholds: Sequence[float], rho: float, train_flag: bool, add_self: bool = False) -> np.ndarray:
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


def choose_aggregate(g: DeviceGraph, nodes: torch.Tensor, labels: Optional[torch.Tensor], s0: torch.Tensor,
                     pos_keys: Optional[torch.Tensor], thresholds: Sequence[float], rho: float, train_flag: bool,
                     norm: int = _lib.PCG_NORM_COUNT, add_self: bool = False,
                     center_s0: Optional[torch.Tensor] = None, ws: Optional[ChooseWorkspace] = None,
                     agg: Optional[torch.Tensor] = None, cnt: Optional[torch.Tensor] = None,
                     sel_begin: Optional[torch.Tensor] = None, sel_indices: Optional[torch.Tensor] = None):
    """Fused choose + mean for all relations of a batch -> agg [R, B, F] (and |set| [R, B])."""
    lib = _lib.load()
    B = nodes.numel()
    if ws is None or ws.B < B:
        ws = ChooseWorkspace(g, B)
    if agg is None:
        agg = torch.empty(g.R, B, g.feat_dim, dtype=torch.float32, device=g.device)
    if cnt is None:
        cnt = torch.empty(g.R, B, dtype=torch.int32, device=g.device)
    thr = (C.c_double * g.R)(*[float(t) for t in thresholds])
    rhos = (C.c_double * g.R)(*([float(rho)] * g.R if np.isscalar(rho) else [float(x) for x in rho]))
    cap = 0 if sel_indices is None else sel_indices.numel()
    _lib.check(lib.pcg_choose_aggregate(
        g.desc_ref(), _p(nodes), _p(labels), B, _p(s0), _p(center_s0), _p(pos_keys), thr, rhos,
        1 if train_flag else 0, norm, 1 if add_self else 0, _p(agg), agg.stride(1), _p(cnt),
        _p(sel_begin), _p(sel_indices), cap, _p(ws.buf), _p(ws.status), _stream(g.device)), "pcg_choose_aggregate")
    return agg, cnt


def chosen_sets(g: DeviceGraph, nodes, labels, s0, pos_keys, thresholds, rho, train_flag,
                norm=_lib.PCG_NORM_COUNT, add_self=False, center_s0=None):
    """Debug / parity helper: run the hot kernel with materialisation on and copy the
    chosen index sets to the host.  Returns (sets[r][b], agg, cnt)."""
    nodes_h = nodes.cpu().numpy()
    labels_h = None if labels is None else labels.cpu().numpy()
    caps = sel_capacity(g, nodes_h, labels_h, thresholds, rho, train_flag, add_self)
    begin = np.zeros(caps.size, dtype=np.int64)
    np.cumsum(caps.reshape(-1)[:-1], out=begin[1:])
    total = int(caps.sum())
    sel_begin = torch.from_numpy(begin).to(g.device)
    sel_idx = torch.full((max(total, 1),), -1, dtype=torch.int32, device=g.device)
    ws = ChooseWorkspace(g, nodes.numel())
    agg, cnt = choose_aggregate(g, nodes, labels, s0, pos_keys, thresholds, rho, train_flag, norm, add_self,
                                center_s0, ws, sel_begin=sel_begin, sel_indices=sel_idx)
    torch.cuda.synchronize(g.device)
    if int(ws.status.item()) & _lib.PCG_ST_SEL_OVERFLOW:
        raise _lib.PcgnnLibraryError("selection buffer overflow (capacity bound is wrong)")
    idx_h, cnt_h = sel_idx.cpu().numpy(), cnt.cpu().numpy()
    B = nodes.numel()
    sets: List[List[Set[int]]] = []
    for r in range(g.R):
        row_sets = []
        for b in range(B):
            s = begin[r * B + b]
            row_sets.append(set(idx_h[s:s + cnt_h[r, b]].tolist()))
        sets.append(row_sets)
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
