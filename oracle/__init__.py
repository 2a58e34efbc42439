"""CPU oracle for the BEVFusion hot path.

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg.  The product package never imports this module
(``tests/test_no_oracle_in_product.py`` enforces it).
"""
from .oracle import *  # noqa: F401,F403
