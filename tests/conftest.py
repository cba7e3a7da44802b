import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference mounted (build container only)")


@pytest.fixture(autouse=True)
def _seed():
    # the reference seeds torch/random/numpy with 787 in tests/conftest.py:16-18
    import random
    random.seed(787)
    np.random.seed(787)
    try:
        import torch
        torch.manual_seed(787)
    except Exception:
        pass
    yield


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))
