"""Importable alias for the package directory ``audio-key-estimation_amd/`` (a hyphen cannot
appear in an ``import`` statement): ``import ake_amd`` == that package, and
``ake_amd.<submodule>`` == the same module objects (no second copy is ever imported)."""
import importlib
import os
import sys

_REAL = "audio-key-estimation_amd"
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_REAL + "."):
        sys.modules["ake_amd" + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
