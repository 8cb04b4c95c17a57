"""Import alias: `import cnerf_amd` gives the package in ./conditioned-nerf-gan_amd (a name Python's import
statement cannot spell)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("conditioned-nerf-gan_amd")
sys.modules[__name__] = _pkg
