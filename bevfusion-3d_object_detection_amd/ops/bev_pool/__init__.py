"""bev_pool operator (same public name as the reference's bevfusion/ops/bev_pool package) on csrc/bev_pool.hip."""
from . import bev_pool_ext  # C-ABI backed replacement of the reference's pybind module of the same name
from .bev_pool import bev_pool as bev_pool

__all__ = ["bev_pool", "bev_pool_ext"]
