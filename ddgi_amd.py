"""Importable alias of the `dynamic-diffuse-global-illumination-minecraft_amd` package (whose
mandated directory name is not a valid Python identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("dynamic-diffuse-global-illumination-minecraft_amd")
sys.modules[__name__] = _pkg
