"""`import macx` == the package in ./mac-network_amd (whose directory name is not a Python identifier)."""
import importlib
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module("mac-network_amd")
sys.modules[__name__] = _pkg
