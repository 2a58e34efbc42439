"""MI355X-native BEVFusion hot path (gfx950 HIP kernels behind a C ABI).

Host-side mirror of the reference's operator interface
(lhn0323/BEVFUSION-3D_object_detection, projects/BEVFusion/bevfusion/ops):

    from <this package>.ops import bev_pool, Voxelization, DynamicScatter

The directory name contains a hyphen, so import it through `bevfusion_amd` (repo root), which
aliases this package, or with importlib.import_module("bevfusion-3d_object_detection_amd").
"""
from . import _lib  # noqa: F401  (ctypes binding; loads the .so lazily, fails loudly if absent)

__version__ = "0.1.0"


def build(force: bool = False):
    """Compile csrc/*.hip for gfx950 into csrc/libbevfusion_hip.so (in-tree)."""
    import importlib
    return importlib.import_module(__name__ + "._build").build(force=force)
