"""Importable alias for the package directory `mc-alf_amd/` (a hyphen cannot appear in an
import statement).  `import mcalf_amd` returns that package; submodules are aliased too."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("mc-alf_amd")
for _k in list(sys.modules):
    if _k.startswith("mc-alf_amd."):
        sys.modules["mcalf_amd" + _k[len("mc-alf_amd"):]] = sys.modules[_k]
sys.modules[__name__] = _pkg
