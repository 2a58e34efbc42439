import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_vox():
    return np.load(os.path.join(GOLDEN, "voxelization_ref.npz"))


@pytest.fixture(scope="session")
def golden_lss():
    return np.load(os.path.join(GOLDEN, "lss_ref.npz"))


@pytest.fixture(scope="session")
def golden_bev():
    """Outputs of the reference's own QuickCumsum (fp64) + interval tables (tests/golden/make_golden.py section D)."""
    return np.load(os.path.join(GOLDEN, "bev_pool_ref.npz"))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
