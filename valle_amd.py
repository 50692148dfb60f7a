"""Import shim: the package directory is ``vall-e_amd/`` (not a valid Python identifier);
``import valle_amd`` resolves to it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("vall-e_amd")
sys.modules[__name__] = _pkg
