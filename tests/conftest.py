import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import pm_oracle
    pm_oracle.build()
    return pm_oracle


@pytest.fixture(scope="session")
def ctx():
    """One pm_ctx on cuda:0 for the GPU parity tests.  No GPU => the gpu tests fail loudly
    (they are only collected with -m gpu on the GPU box)."""
    import points_matching_amd as pm
    c = pm.Context(0)
    yield c
    c.close()
