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
