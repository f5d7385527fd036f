"""Import shim: ``import recbole_gnn_amd`` -> the package in ``recbole-gnn_amd/`` (a hyphen is not a
valid identifier, so the real package is loaded through importlib and aliased)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("recbole-gnn_amd")
sys.modules[__name__] = _pkg
# ... and its submodules under the alias too: `from recbole_gnn_amd._lib import X` must find the module that is already loaded,
# not execute a second copy of it (a second CDLL handle, a second RbgError class that `except rbg.RbgError` does not catch)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("recbole-gnn_amd."):
        sys.modules.setdefault(__name__ + _name[len("recbole-gnn_amd"):], _mod)
