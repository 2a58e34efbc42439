from . import bev_pool_ext
from .bev_pool import bev_pool

__all__ = ["bev_pool", "bev_pool_ext"]
