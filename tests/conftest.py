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


def pytest_collection_modifyitems(config, items):
    """The multi-process tests whose ranks all share ONE GPU (parametrised by `world`) run last, smallest world first: they depend on
    how the GPU scheduler time-slices 2..8 (+1) processes, which the product -- one process per GPU -- never asks of it; under `-x`
    nothing else of the suite should be hidden behind them."""
    def world_of(item):
        cs = getattr(item, "callspec", None)
        if cs is None or "world" not in cs.params or item.get_closest_marker("gpu") is None:
            return 0
        return int(cs.params["world"])
    items.sort(key=world_of)      # stable: everything else keeps its order


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


def golden_planar(g):
    """True for a FORMAT.GPTQ_P fixture (3-bit split-plane words), None otherwise (the bit width's own layout: 5 / 6 / 7 always planar)."""
    return True if "planar" in g.files and int(g["planar"]) else None
