import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's torch-CPU leg must not oversubscribe the box's CPU share (16 cores per GPU)
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def synthetic_weights():
    from coupe.dvsg_amd.weights import make_synthetic_weights
    return make_synthetic_weights(seed=0)
