"""pc-gnn_amd: the PC-GNN pick -> choose -> aggregate hot path for AMD MI355X (gfx950).

HIP kernels behind a C ABI (``include/pcgnn.h``, built into ``lib/libpcgnn_hip.so``)
plus a Python host side that mirrors the reference's ``PCALayer`` / ``InterAgg*`` /
``IntraAgg`` / ``pick_step`` interface.  There is no CPU fallback: using the layers
without the built library, or without a GPU, raises.

The directory name has a hyphen; ``import pcgnn_amd`` (alias module at the repo
root) or ``importlib.import_module("pc-gnn_amd")`` both give this package.
"""
from . import _lib  # noqa: F401
from ._lib import PcgnnLibraryError  # noqa: F401
from .build import build_library, lib_path  # noqa: F401


def __getattr__(name):
    # torch-dependent pieces are imported on first use so that `build()` works in
    # an environment where only the toolchain is wanted
    import importlib
    table = {
        "DeviceGraph": ".graph", "adj_to_csr": ".graph",
        "IntraAgg": ".layers", "InterAgg": ".layers", "InterAgg1": ".layers", "InterAgg3": ".layers",
        "InterAgg5": ".layers", "PCALayer": ".model", "ModelHandler": ".handler", "PCGNNTrainer": ".handler",
        "ResultManager": ".result_manager", "result_manager": None, "utils": None, "synth": None, "fused": None, "ops": None, "layers": None, "model": None, "graph": None,
        "sampler": None, "handler": None, "graphsage": None, "dist": None,
    }
    if name in table:
        if table[name] is None:
            return importlib.import_module("." + name, __name__)
        return getattr(importlib.import_module(table[name], __name__), name)
    raise AttributeError(name)
