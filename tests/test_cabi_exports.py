"""CPU checks of the drop-in boundary: the shared library builds/loads and exports
every symbol include/pcgnn.h declares; host-side helpers agree with the oracle.
No compute calls (there is no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pcgnn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pcg_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import pcgnn_amd
    from pcgnn_amd import _lib
    pcgnn_amd.build_library()
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"libpcgnn_hip.so does not export {s}"
        assert s in _lib.PROTOTYPES, f"ctypes prototype missing for {s}"
    assert lib.pcg_abi_version() == _lib.ABI_VERSION
    assert lib.pcg_version().startswith(b"pcgnn_hip gfx950")


def test_host_helpers_and_argument_checks():
    from pcgnn_amd import _lib
    lib = _lib.load()
    # pcg_sel_capacity_row == the oracle's counting rule (layers.py:260-262, 662, 681)
    import math
    for deg in range(0, 40):
        for thr in (0.2, 0.5, 0.8, 1.0):
            for rho in (0.0, 0.5, 2.0):
                k = math.ceil(deg * thr)
                want = (k if deg > k + 1 else deg) + min(int(k * rho), 7)
                assert lib.pcg_sel_capacity_row(deg, thr, rho, 1, 7, 0) == want
                assert lib.pcg_sel_capacity_row(deg, thr, rho, 0, 7, 1) == (k if deg > k + 1 else deg) + 1
    assert lib.pcg_pos_sort_capacity(0) == 8192 and lib.pcg_pos_sort_capacity(4097) == 16384 and lib.pcg_pos_sort_capacity(20000) == 65536
    # null / inconsistent arguments are rejected before any launch
    assert lib.pcg_score_table(None, None, None, 0, 0, None, None) == _lib.PCG_E_ARG
    assert lib.pcg_pick(None, None, 0, None, 0, 0, 1, None, None) == _lib.PCG_E_ARG
    assert lib.pcg_choose_workspace_bytes(None, 4, 10) == _lib.PCG_E_ARG


def test_product_path_refuses_cpu():
    """No CPU fallback: building a graph on a CPU device must raise."""
    import torch
    import pcgnn_amd
    with pytest.raises(pcgnn_amd.PcgnnLibraryError):
        pcgnn_amd.DeviceGraph(np.zeros((4, 8), np.float32), [(np.zeros(5, np.int64), np.zeros(0, np.int32))], [],
                              torch.device("cpu"))


def test_adj_to_csr_matches_oracle():
    from oracle import pcgnn_oracle as O
    from pcgnn_amd.graph import adj_to_csr
    rs = np.random.RandomState(0)
    n = 200
    adj = {v: {v} | set(rs.randint(0, n, size=rs.randint(0, 9)).tolist()) for v in range(n)}
    a, b = adj_to_csr(adj, n)
    c, d = O.adj_to_csr(adj, n)
    assert np.array_equal(a, c) and np.array_equal(b, d)
