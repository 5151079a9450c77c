"""Alias: `import fsr1_amd` == the package in ./fidelityfx-fsr_amd (whose name is not an identifier)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("fidelityfx-fsr_amd")
sys.modules[__name__] = _pkg
