"""Importable alias of the ``pc-gnn_amd`` package (its directory name has a hyphen,
which the ``import`` statement cannot spell):  ``import pcgnn_amd`` == the package, and
``pcgnn_amd.sub`` IS ``pc-gnn_amd.sub`` - one module object under both names (two copies of a module
would mean two copies of every class, and ``isinstance`` checks across them would fail)."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_REAL, _ALIAS = "pc-gnn_amd", "pcgnn_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """``pcgnn_amd.x.y`` -> the module object of ``pc-gnn_amd.x.y``"""

    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[__name__] = _pkg
sys.modules.setdefault(_ALIAS, _pkg)
