"""Importable alias of the ``pc-gnn_amd`` package (its directory name has a hyphen,
which the ``import`` statement cannot spell):  ``import pcgnn_amd`` == the package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pc-gnn_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("pcgnn_amd", _pkg)
