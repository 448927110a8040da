"""Importable alias for the package directory ``audio-key-estimation_amd/`` (a hyphen cannot
appear in an ``import`` statement): ``import ake_amd`` == that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("audio-key-estimation_amd")
sys.modules[__name__] = _pkg
