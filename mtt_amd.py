"""Importable alias of the `multi-task-transformer_amd/` package (its directory name has a hyphen)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("multi-task-transformer_amd")
sys.modules[__name__] = _pkg
