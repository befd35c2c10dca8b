"""Import alias: ``import rain_rendering_amd`` -> the package directory ``rain-rendering_amd/``."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("rain-rendering_amd")
sys.modules[__name__] = _pkg
