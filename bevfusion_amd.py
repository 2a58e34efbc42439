"""Importable alias of the hyphen-named package directory `bevfusion-3d_object_detection_amd/`.

    import bevfusion_amd
    from bevfusion_amd.ops import bev_pool, Voxelization
"""
import importlib
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("bevfusion-3d_object_detection_amd")
sys.modules[__name__] = _pkg
sys.modules.setdefault("bevfusion_amd", _pkg)
